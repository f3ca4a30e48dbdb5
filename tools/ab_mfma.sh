#!/bin/bash
# A/B on one box: the defect contraction on the matrix cores (-DPC_MFMA_DEFECT) against the default vector form
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["device_ms_per_step"]*1e3,2))'
for spec in "--steps 5000 --warmup 500" "--problem cart_pole --sections 5000 --order 4 --steps 2000 --warmup 200" "--problem shuttle --sections 2000 --order 4 --steps 1000 --warmup 100" "--problem shuttle --sections 20000 --order 4 --steps 500 --warmup 50"; do
  for d in "" "PC_MFMA_DEFECT"; do for r in 1 2; do
    echo -n "[$spec] [${d:-vector form}] "
    PYCOLLO_AMD_DEFINES="$d" timeout -k 10 300 python bench.py --no-cpu --no-host $spec 2>/dev/null | python3 -c "$fmt" || echo failed
  done; done
done
