#!/bin/bash
# five default-build bench lines (two repetitions): the quick look after a kernel change
fmt='import sys,json; d=json.loads(sys.stdin.read()); print("evals/s", d["value"], "| device us", round(d["device_ms_per_step"]*1e3,3), "| bulk us", d["roofline"]["avg_launch_us"], "| GB/s", d["roofline"]["achieved"])'
for cfg in "--steps 20000" "--problem cart_pole --sections 5000 --order 4 --steps 5000" "--problem shuttle --sections 2000 --order 4 --steps 3000" "--problem shuttle --sections 20000 --order 4 --steps 1000" "--sections 200000 --order 6 --steps 300" "--problem delta_iii --sections 3125 --order 5 --steps 300" "--problem hypersensitive --sections 20000 --ragged --steps 300"; do
  echo "== $cfg"
  for rep in 1 2; do
    timeout -k 10 400 python bench.py --no-cpu $cfg 2>/dev/null | python3 -c "$fmt" || echo failed
  done
done
