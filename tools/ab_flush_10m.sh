#!/bin/bash
# A/B on one box at 1e7 nodes (1.6 GB per evaluation): flush forms
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["device_ms_per_step"]*1e3,2), d["roofline"]["frac"])'
for rep in 1 2; do for d in "" "PC_FLUSH_PEEL" "PC_FLUSH_PRED_ALL"; do
  echo -n "[10M nodes] [${d:-default}] "
  PYCOLLO_AMD_DEFINES="$d" timeout -k 10 300 python bench.py --no-cpu --no-host --sections 2000000 --order 6 --steps 60 --warmup 6 2>/dev/null | python3 -c "$fmt" || echo failed
done; done
