#!/bin/bash
# what fused multiply-add contraction would buy (the default build keeps it off for reference-exact rounding)
fmt='import sys,json; d=json.loads(sys.stdin.read()); print("evals/s", d["value"], "| device us", round(d["device_ms_per_step"]*1e3,3), "| bulk us", d["roofline"]["avg_launch_us"], "| GB/s", d["roofline"]["achieved"])'
for cfg in "--steps 20000" "--problem cart_pole --sections 5000 --order 4 --steps 5000" "--problem shuttle --sections 20000 --order 4 --steps 1000" "--problem shuttle --sections 200000 --order 4 --steps 100" "--problem delta_iii --sections 3125 --order 5 --steps 500" "--problem space_station --sections 2000 --order 4 --steps 2000"; do
  echo "== $cfg"
  for fc in off fast; do
    echo -n "[contract $fc] "; PYCOLLO_AMD_FP_CONTRACT=$fc timeout -k 10 400 python bench.py --no-cpu $cfg 2>/dev/null | python3 -c "$fmt" || echo failed
  done
done
