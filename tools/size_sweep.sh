#!/bin/bash
# usage: tools/size_sweep.sh  -- default-configuration bench lines over problem families and sizes
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(d["config"]["workload"][:72], "| evals/s", d["value"], "| TB", d["config"]["threads_per_block"], "x", d["config"].get("waves_per_tile"), "| bulk us", d["roofline"]["avg_launch_us"], "| GB/s", d["roofline"]["achieved"])'
for cfg in "--steps 2000" "--sections 200000 --order 6 --steps 300" "--problem cart_pole --sections 5000 --order 4 --steps 1000" "--problem cart_pole --sections 100000 --order 4 --steps 200" "--problem shuttle --sections 2000 --order 4 --steps 1000" "--problem shuttle --sections 20000 --order 4 --steps 300" "--problem shuttle --sections 200000 --order 4 --steps 100" "--problem delta_iii --sections 3125 --order 5 --steps 300" "--problem hypersensitive --sections 20000 --ragged --steps 300" "--problem shuttle --sections 20000 --ragged --steps 100"; do
  timeout -k 10 200 python bench.py --no-cpu $cfg 2>/dev/null | python3 -c "$fmt" || echo "failed: $cfg"
done
