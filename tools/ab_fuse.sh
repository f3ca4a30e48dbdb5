#!/bin/bash
# A/B on one box: two launches per evaluation against the fused tail (PYCOLLO_AMD_FUSE=1), long runs, interleaved
fmt='import sys,json; d=json.loads(sys.stdin.read()); print("evals/s", d["value"], "| ms/step", d["ms_per_step"], "| bulk us", d["roofline"]["avg_launch_us"], "| launches", d["config"].get("launches_per_eval"))'
for rep in 1 2 3 4; do
for f in 0 1; do
  echo -n "[FUSE=$f] "
  PYCOLLO_AMD_FUSE=$f timeout -k 10 300 python bench.py --no-cpu 2>/dev/null | python3 -c "$fmt" || echo failed
done; done
