#!/bin/bash
# does the host launch rate (and with it evals/s at the headline size) depend on where the launching thread runs?
fmt='import sys,json; d=json.loads(sys.stdin.read()); print("evals/s", d["value"], "| ms/step", d["ms_per_step"], "| bulk us", d["roofline"]["avg_launch_us"])'
python3 -c "import os; print('allowed cpus', sorted(os.sched_getaffinity(0)))"
lscpu | grep -i "numa\|model name\|^CPU(s)" | head -8
cpus=$(python3 -c "import os; a=sorted(os.sched_getaffinity(0)); print(' '.join(str(c) for c in (a[0], a[len(a)//2], a[-1])))")
for rep in 1 2; do
  echo -n "[no pin] "; timeout -k 10 300 python bench.py --no-cpu 2>/dev/null | python3 -c "$fmt"
  for c in $cpus; do
    echo -n "[taskset -c $c] "; timeout -k 10 300 taskset -c $c python bench.py --no-cpu 2>/dev/null | python3 -c "$fmt"
  done
done
