#!/bin/bash
# A/B: replica index as a template argument (PC_STATIC_W=n: one instantiation of the bulk body per replica) against the
# run-time replica index, on workloads whose tiles are shared by n waves.  usage: tools/ab_static_w.sh
run() { python bench.py "$@" --steps 1000 --warmup 100 --no-cpu --no-host 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['device_ms_per_step'], d['config']['waves_per_tile'])"; }
for spec in "4|--problem cart_pole --sections 5000 --order 4" "2|--problem shuttle --sections 20000 --order 4" "4|--problem shuttle --sections 2000 --order 4"; do
  n=${spec%%|*}; w=${spec#*|}
  echo "== $w  dynamic"; run $w
  echo "== $w  PC_STATIC_W=$n"; PYCOLLO_AMD_DEFINES="PC_STATIC_W=$n" run $w
done
