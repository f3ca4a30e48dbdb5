#!/bin/bash
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"]["launch_thread_cpu"])'
for rep in 1 2 3 4 5 6; do
for w in 1 2 4; do
  echo -n "[width $w] "; PYCOLLO_AMD_PIN_WIDTH=$w timeout -k 10 300 python bench.py --no-cpu 2>/dev/null | python3 -c "$fmt"
done; done
