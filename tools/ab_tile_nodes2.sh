#!/bin/bash
# Delta III 4 x 12.5 k, two waves per tile (static + split kernel): tile capacity sweep
fmt='import sys,json; d=json.loads(sys.stdin.read()); print("tiles", d["config"].get("tiles"), "| W", d["config"]["waves_per_tile"], "| lds", d["config"].get("lds_bytes_per_workgroup"), "|", d["roofline"]["kernel"], "| device us", round(d["device_ms_per_step"]*1e3,2), "| frac", round(d["roofline"]["frac"],3))'
for tn in 64 61 57 53 49 45; do
  echo -n "[tn $tn] "
  PYCOLLO_AMD_TILE_NODES=$tn PYCOLLO_AMD_WPT=2 timeout -k 10 300 python bench.py --no-cpu --no-host --problem delta_iii --sections 3125 --order 5 --steps 300 --warmup 50 2>/dev/null | python3 -c "$fmt" || echo failed
done
echo -n "[auto] "; timeout -k 10 300 python bench.py --no-cpu --no-host --problem delta_iii --sections 3125 --order 5 --steps 300 --warmup 50 2>/dev/null | python3 -c "$fmt"
