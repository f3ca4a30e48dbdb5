#!/bin/bash
fmt='import sys,json; d=json.loads(sys.stdin.read()); print("evals/s", d["value"], "| ms/step", d["ms_per_step"], "| bulk us", d["roofline"]["avg_launch_us"], "| W", d["config"]["waves_per_tile"])'
for rep in 1 2 3; do
for w in 1 2; do
  echo -n "[WPT=$w] "
  PYCOLLO_AMD_WPT=$w timeout -k 10 300 python bench.py --no-cpu 2>/dev/null | python3 -c "$fmt" || echo failed
done; done
