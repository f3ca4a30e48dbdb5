#!/bin/bash
# heavy model on a small mesh (space station, 96 tiles): run-time replica index against per-replica kernels for W = 4 / 2
fmt='import sys,json; d=json.loads(sys.stdin.read()); print("tiles", d["config"].get("tiles"), "| W", d["config"]["waves_per_tile"], "| lds", d["config"].get("lds_bytes_per_workgroup"), "|", d["roofline"]["kernel"], "| device us", round(d["device_ms_per_step"]*1e3,2), "| frac", round(d["roofline"]["frac"],3))'
run() { local label=$1 envs=$2; shift 2; echo -n "[$label] "; env $envs timeout -k 10 300 python bench.py --no-cpu --no-host "$@" 2>/dev/null | python3 -c "$fmt" || echo failed; }
A="--problem space_station --sections 2000 --order 4 --steps 300 --warmup 30"
for rep in 1 2; do
run "station default (W4 run-time)" "" $A
run "station W4 per-replica       " "PYCOLLO_AMD_HEAVY_W4=1" $A
run "station W2 per-replica       " "PYCOLLO_AMD_WPT=2" $A
run "station W1                   " "PYCOLLO_AMD_WPT=1" $A
done
B="--problem shuttle --sections 2000 --order 4 --steps 300 --warmup 30"
run "shuttle 6k default" "" $B
