#!/bin/bash
# heavy model (Delta III), split + register-capped kernels: one wave per tile against the two-wave build, by mesh size
fmt='import sys,json; d=json.loads(sys.stdin.read()); print("tiles", d["config"].get("tiles"), "| W", d["config"]["waves_per_tile"], "| lds", d["config"].get("lds_bytes_per_workgroup"), "|", d["roofline"]["kernel"], "| device us", round(d["device_ms_per_step"]*1e3,2), "| frac", round(d["roofline"]["frac"],3))'
run() { local label=$1 envs=$2; shift 2; echo -n "[$label] "; env $envs timeout -k 10 300 python bench.py --no-cpu --no-host "$@" 2>/dev/null | python3 -c "$fmt" || echo failed; }
for spec in "3125 5 300" "6250 5 200" "12500 5 100" "4167 4 300" "2605 6 300"; do
  set -- $spec
  A="--problem delta_iii --sections $1 --order $2 --steps $3 --warmup 30"
  run "K=$1 n=$2 auto     " "" $A
  run "K=$1 n=$2 W1       " "PYCOLLO_AMD_TWO_WAVE=0 PYCOLLO_AMD_WPT=1" $A
  run "K=$1 n=$2 W2 forced" "PYCOLLO_AMD_TWO_WAVE_MAX_TILES=100000" $A
done
R="--problem delta_iii --sections 2500 --ragged --steps 200 --warmup 30"
run "ragged auto        " "" $R
run "ragged W1          " "PYCOLLO_AMD_TWO_WAVE=0 PYCOLLO_AMD_WPT=1" $R
run "ragged W2 tn64     " "PYCOLLO_AMD_TILE_NODES=64 PYCOLLO_AMD_WPT=2" $R
run "ragged W2 tn48     " "PYCOLLO_AMD_TILE_NODES=48 PYCOLLO_AMD_WPT=2" $R
