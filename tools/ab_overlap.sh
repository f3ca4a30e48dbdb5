#!/bin/bash
# does the tile kernel overlap its arithmetic with its stores?  no-store build (arithmetic + LDS only), split build (first
# partials -> Jacobian stores -> second partials -> Hessian stores), split + staggered waves, non-temporal stores
fmt='import sys,json; d=json.loads(sys.stdin.read()); print("tiles", d["config"].get("tiles"), "| W", d["config"]["waves_per_tile"], "| device us", round(d["device_ms_per_step"]*1e3,2))'
run() { local label=$1 envs=$2; shift 2; for r in 1 2; do echo -n "[$label] "; env $envs timeout -k 10 300 python bench.py --no-cpu --no-host "$@" 2>/dev/null | python3 -c "$fmt" || echo failed; done; }
for d in "" "PC_EXP_NOSTORE" "PC_SPLIT_MIN=40" "PC_SPLIT_MIN=40 PC_STAGGER=64" "PC_NT_STORES"; do
  run "shuttle 60k [$d]" "PYCOLLO_AMD_DEFINES=$d" --problem shuttle --sections 20000 --order 4 --steps 500 --warmup 50
done
for d in "" "PC_EXP_NOSTORE" "PC_NT_STORES"; do
  run "d3 12.5k [$d]" "PYCOLLO_AMD_DEFINES=$d" --problem delta_iii --sections 3125 --order 5 --steps 300 --warmup 50
done
