#!/bin/bash
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"])'
for rep in 1 2 3 4 5 6; do
for v in 0 1; do
  echo -n "[HIP_FORCE_DEV_KERNARG=$v] "; HIP_FORCE_DEV_KERNARG=$v timeout -k 10 300 python bench.py --no-cpu 2>/dev/null | python3 -c "$fmt"
done; done
