#!/bin/bash
# A/B on one box: odd tiles walk the defect-Jacobian states last to first (PC_REVERSE_ODD)
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(d["roofline"]["kernel"], round(d["device_ms_per_step"]*1e3,2))'
run() { local label=$1 defs=$2; shift 2; for r in 1 2; do echo -n "[$label] [${defs:-default}] "; PYCOLLO_AMD_DEFINES="$defs" timeout -k 10 300 python bench.py --no-cpu --no-host "$@" 2>/dev/null | python3 -c "$fmt" || echo failed; done; }
D="--problem delta_iii --sections 3125 --order 5 --steps 300 --warmup 50"
S="--problem shuttle --sections 20000 --order 4 --steps 500 --warmup 50"
run "d3 4x12.5k n5" "" $D; run "d3 4x12.5k n5" "PC_REVERSE_ODD" $D; run "d3 4x12.5k n5" "" $D
run "shuttle 60k" "" $S; run "shuttle 60k" "PC_REVERSE_ODD" $S; run "shuttle 60k" "" $S
