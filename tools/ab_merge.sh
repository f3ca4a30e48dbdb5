#!/bin/bash
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(d["config"]["workload"][:60], "| evals/s", d["value"], "| W", d["config"]["waves_per_tile"], "| launches", d["config"]["launches_per_eval"], "| bulk us", d["roofline"]["avg_launch_us"], "| GB/s", d["roofline"]["achieved"])'
for cfg in "--problem delta_iii --sections 3125 --order 5 --steps 300" "--problem delta_iii --sections 12500 --order 5 --steps 100" "--problem delta_iii --sections 300 --order 5 --steps 1000"; do
for m in 0 1; do
  echo -n "[MERGE=$m] "
  PYCOLLO_AMD_MERGE=$m timeout -k 10 300 python bench.py --no-cpu $cfg 2>/dev/null | python3 -c "$fmt" || echo failed
done; done
