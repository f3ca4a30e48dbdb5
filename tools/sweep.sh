#!/bin/bash
# usage: tools/sweep.sh  -- TB sweep over the mid-size configs
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(d["config"]["workload"][:64], "| evals/s", d["value"], "| TB", d["config"]["threads_per_block"], "| bulk us", d["roofline"]["avg_launch_us"], "| GB/s", d["roofline"]["achieved"])'
for cfg in "--problem cart_pole --sections 5000 --order 4 --steps 1000" "--problem shuttle --sections 20000 --order 4 --steps 300" "--problem hypersensitive --sections 20000 --order 6 --steps 1000" "--problem delta_iii --sections 3125 --order 5 --steps 300"; do
  for t in 64 128 256; do
    timeout -k 10 200 python bench.py --no-cpu $cfg --tpb $t 2>/dev/null | python3 -c "$fmt" || echo "failed: $cfg tpb $t"
  done
done
