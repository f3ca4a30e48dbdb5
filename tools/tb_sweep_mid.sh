#!/bin/bash
# TB sweep around the auto-selection thresholds (W forced to 1 so that only the tile size varies)
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(d["config"]["workload"][:58], "| TB", d["config"]["threads_per_block"], "| bulk us", d["roofline"]["avg_launch_us"], "| GB/s", d["roofline"]["achieved"])'
for cfg in "--sections 10000 --order 6 --steps 2000" "--sections 20000 --order 6 --steps 2000" "--sections 40000 --order 6 --steps 1000" "--sections 80000 --order 6 --steps 500" "--problem cart_pole --sections 25000 --order 4 --steps 1000" "--problem cart_pole --sections 50000 --order 4 --steps 500" "--problem shuttle --sections 30000 --order 4 --steps 300"; do
for t in 64 128 256; do
  PYCOLLO_AMD_WPT=1 timeout -k 10 200 python bench.py --no-cpu $cfg --tpb $t 2>/dev/null | python3 -c "$fmt" || echo "failed: $cfg"
done; done
