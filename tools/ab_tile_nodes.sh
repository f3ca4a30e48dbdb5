#!/bin/bash
# tile capacity below the workgroup size: bench lines per PYCOLLO_AMD_TILE_NODES (TB stays 64)
fmt='import sys,json; d=json.loads(sys.stdin.read()); print("evals/s", d["value"], "| device us", round(d["device_ms_per_step"]*1e3,3), "| tiles", d["config"]["tiles"], "x W", d["config"]["waves_per_tile"], "| bulk us", d["roofline"]["avg_launch_us"])'
for cfg in "--steps 20000" "--problem cart_pole --sections 5000 --order 4 --steps 5000" "--problem shuttle --sections 2000 --order 4 --steps 3000"; do
  echo "== $cfg"
  for tc in 64 48 32 24 16; do
    echo -n "[tile nodes $tc] "; PYCOLLO_AMD_TB=64 PYCOLLO_AMD_TILE_NODES=$tc timeout -k 10 200 python bench.py --no-cpu $cfg 2>/dev/null | python3 -c "$fmt" || echo failed
  done
done
