#!/bin/bash
# the multi-phase (Delta III) lines only
fmt='import sys,json; d=json.loads(sys.stdin.read()); print("tiles", d["config"].get("tiles"), "| TB", d["config"]["threads_per_block"], "x", d["config"]["waves_per_tile"], "|", d["roofline"]["kernel"], "| device us", round(d["device_ms_per_step"]*1e3,2), "| frac", round(d["roofline"]["frac"],3))'
run() { local label=$1; shift; for rep in 1 2; do echo -n "[$label] "; timeout -k 10 300 python bench.py --no-cpu --no-host "$@" 2>/dev/null | python3 -c "$fmt" || echo failed; done; }
run "d3 4x12.5k n5" --problem delta_iii --sections 3125 --order 5 --steps 300 --warmup 50
run "d3 4x12.5k n4" --problem delta_iii --sections 4167 --order 4 --steps 300 --warmup 50
run "d3 4x12.5k n6" --problem delta_iii --sections 2605 --order 6 --steps 300 --warmup 50
run "d3 ragged 50k" --problem delta_iii --sections 2500 --ragged --steps 200 --warmup 30
run "d3 4x50k n5  " --problem delta_iii --sections 12500 --order 5 --steps 100 --warmup 20
