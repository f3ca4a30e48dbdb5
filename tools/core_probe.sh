#!/bin/bash
# does a slow launching core announce itself?  per process: the core picked, a fixed interpreter loop timed on it
# before HIP initialises, and the evaluation rate that process then reaches
fmt='import sys,json; d=json.loads(sys.stdin.read()); print("cpu", d["config"]["launch_thread_cpu"], "moved", d["config"].get("runtime_threads_colocated"), "| evals/s", d["value"], "| ms/step", d["ms_per_step"], "| device", d["device_ms_per_step"])'
lscpu | grep -i "model name\|^CPU(s)\|thread(s) per core\|numa node" | head -12
python3 -c "import os; a=sorted(os.sched_getaffinity(0)); print('allowed', len(a), a[:4], '...', a[-4:])"
for rep in 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16; do
  PYCOLLO_AMD_SPIN_PROBE=1 timeout -k 10 200 python bench.py --no-cpu 2>gpurun_out/spin.err | python3 -c "$fmt"
  grep "spin probe" gpurun_out/spin.err
done
