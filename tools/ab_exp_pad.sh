#!/bin/bash
# timing-only experiment: padded LDS staging image of the defect Jacobian runs (PC_EXP_PAD: results are wrong on purpose)
fmt='import sys,json; d=json.loads(sys.stdin.read()); print("tiles", d["config"].get("tiles"), "| W", d["config"]["waves_per_tile"], "| lds", d["config"].get("lds_bytes_per_workgroup"), "| device us", round(d["device_ms_per_step"]*1e3,2))'
run() { local label=$1 envs=$2; shift 2; for r in 1 2; do echo -n "[$label] "; env $envs timeout -k 10 300 python bench.py --no-cpu --no-host "$@" 2>/dev/null | python3 -c "$fmt" || echo failed; done; }
D="--problem delta_iii --sections 3125 --order 5 --steps 300 --warmup 50"
run "d3 tn53 base" "PYCOLLO_AMD_TILE_NODES=53 PYCOLLO_AMD_WPT=2 PYCOLLO_AMD_LDS_ROWS_EXTRA=4" $D
run "d3 tn53 pad " "PYCOLLO_AMD_TILE_NODES=53 PYCOLLO_AMD_WPT=2 PYCOLLO_AMD_LDS_ROWS_EXTRA=4 PYCOLLO_AMD_DEFINES=PC_EXP_PAD" $D
S="--problem shuttle --sections 20000 --order 4 --steps 500 --warmup 50"
run "sh tn58 base" "PYCOLLO_AMD_TILE_NODES=58 PYCOLLO_AMD_WPT=2 PYCOLLO_AMD_LDS_ROWS_EXTRA=6" $S
run "sh tn58 pad " "PYCOLLO_AMD_TILE_NODES=58 PYCOLLO_AMD_WPT=2 PYCOLLO_AMD_LDS_ROWS_EXTRA=6 PYCOLLO_AMD_DEFINES=PC_EXP_PAD" $S
run "sh default  " "" $S
