#!/bin/bash
# usage: DEFS="A=1|-|B=2" CFG="--sections ..." tools/ab_defs.sh  -- interleaved A/B of -D sets on one workload, 3 rounds
fmt='import sys,json; d=json.loads(sys.stdin.read()); print("evals/s", d["value"], "| ms/step", d["ms_per_step"], "| bulk us", d["roofline"]["avg_launch_us"], "| GB/s", d["roofline"]["achieved"])'
IFS='|' read -ra SETS <<< "${DEFS:--}"
for rep in 1 2 3; do
  for d in "${SETS[@]}"; do
    [ "$d" = "-" ] && d=""
    echo -n "[${d:-default}] "
    PYCOLLO_AMD_DEFINES="$d" timeout -k 10 300 python bench.py --no-cpu $CFG 2>/dev/null | python3 -c "$fmt" || echo failed
  done
done
