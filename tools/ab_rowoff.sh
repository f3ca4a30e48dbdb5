#!/bin/bash
# A/B on one box: first row offset of a column pass computed once (default) against per row inside the loop (PC_ROWOFF_IN_LOOP)
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["device_ms_per_step"]*1e3,2))'
for spec in "--problem delta_iii --sections 4167 --order 4 --steps 300 --warmup 50" "--problem delta_iii --sections 2605 --order 6 --steps 300 --warmup 50" "--problem delta_iii --sections 2500 --ragged --steps 200 --warmup 30"; do
  for d in "" "PC_ROWOFF_IN_LOOP"; do for r in 1 2; do
    echo -n "[$spec] [${d:-default}] "
    PYCOLLO_AMD_DEFINES="$d" timeout -k 10 300 python bench.py --no-cpu --no-host $spec 2>/dev/null | python3 -c "$fmt" || echo failed
  done; done
done
