#!/bin/bash
# experiment: the wave in the odd hardware slot of every SIMD starts PC_STAGGER x 64 clocks late, so that one wave of a
# SIMD evaluates while the other stores (in lockstep the CU's store path idles during the evaluation phase)
fmt='import sys,json; d=json.loads(sys.stdin.read()); print("tiles", d["config"].get("tiles"), "| W", d["config"]["waves_per_tile"], "| device us", round(d["device_ms_per_step"]*1e3,2), "| frac", round(d["roofline"]["frac"],3))'
run() { local label=$1 envs=$2; shift 2; for r in 1 2; do echo -n "[$label] "; env $envs timeout -k 10 300 python bench.py --no-cpu --no-host "$@" 2>/dev/null | python3 -c "$fmt" || echo failed; done; }
for st in 0 16 32 64 96; do
  e=""; [ $st != 0 ] && e="PYCOLLO_AMD_DEFINES=PC_STAGGER=$st"
  run "shuttle 60k stagger $st" "$e" --problem shuttle --sections 20000 --order 4 --steps 500 --warmup 50
  run "d3 12.5k   stagger $st" "$e" --problem delta_iii --sections 3125 --order 5 --steps 300 --warmup 50
done
