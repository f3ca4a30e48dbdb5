#!/bin/bash
# usage: tools/wpt_sweep.sh  -- waves-per-tile sweep (PYCOLLO_AMD_WPT) over the multi-state configs, 64-node tiles
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(d["config"]["workload"][:64], "| evals/s", d["value"], "| TB", d["config"]["threads_per_block"], "| bulk us", d["roofline"]["avg_launch_us"], "| GB/s", d["roofline"]["achieved"])'
for cfg in "--problem cart_pole --sections 5000 --order 4 --steps 1000" "--problem cart_pole --sections 20000 --order 4 --steps 1000" "--problem shuttle --sections 20000 --order 4 --steps 300" "--problem shuttle --sections 8000 --order 4 --steps 300" "--problem shuttle --sections 60000 --order 4 --steps 300" "--problem delta_iii --sections 3125 --order 5 --steps 300" "--problem delta_iii --sections 12500 --order 5 --steps 100"; do
  for w in 1 2 4; do
    echo -n "WPT=$w  "
    PYCOLLO_AMD_WPT=$w timeout -k 10 200 python bench.py --no-cpu $cfg --tpb 64 2>/dev/null | python3 -c "$fmt" || echo "failed: $cfg wpt $w"
  done
done
