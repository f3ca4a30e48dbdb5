#!/bin/bash
# A/B on one box: the staging flush with the next batch's LDS reads issued ahead of this batch's stores (PC_FLUSH_PIPELINED)
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["device_ms_per_step"]*1e3,2))'
for spec in "--steps 5000 --warmup 500" "--problem cart_pole --sections 5000 --order 4 --steps 2000 --warmup 200" "--problem shuttle --sections 20000 --order 4 --steps 500 --warmup 50" "--problem shuttle --sections 200000 --order 4 --steps 100 --warmup 10" "--sections 200000 --order 6 --steps 300 --warmup 30" "--problem delta_iii --sections 3125 --order 5 --steps 300 --warmup 50" "--problem delta_iii --sections 12500 --order 5 --steps 100 --warmup 20"; do
  for d in "" "PC_FLUSH_PIPELINED"; do for r in 1 2; do
    echo -n "[$spec] [${d:-default}] "
    PYCOLLO_AMD_DEFINES="$d" timeout -k 10 300 python bench.py --no-cpu --no-host $spec 2>/dev/null | python3 -c "$fmt" || echo failed
  done; done
done
