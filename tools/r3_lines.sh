#!/bin/bash
# round-3 quick look: the workloads VERDICT r2 names, two repetitions each; optional env prefix per line
fmt='import sys,json; d=json.loads(sys.stdin.read()); print("tiles", d["config"].get("tiles"), "| TB", d["config"]["threads_per_block"], "x", d["config"]["waves_per_tile"], "|", d["roofline"]["kernel"], "| device us", round(d["device_ms_per_step"]*1e3,2), "| frac", round(d["roofline"]["frac"],3))'
run() {  # label, env assignments (may be empty), bench args...
  local label=$1 envs=$2; shift 2
  for rep in 1 2; do
    echo -n "[$label] "
    env $envs timeout -k 10 300 python bench.py --no-cpu --no-host "$@" 2>/dev/null | python3 -c "$fmt" || echo failed
  done
}
run "config2 hs 10k        " "" --steps 5000 --warmup 500
run "cart-pole 15k         " "" --problem cart_pole --sections 5000 --order 4 --steps 2000 --warmup 200
run "shuttle 6k            " "" --problem shuttle --sections 2000 --order 4 --steps 1000 --warmup 100
run "shuttle 60k           " "" --problem shuttle --sections 20000 --order 4 --steps 500 --warmup 50
run "shuttle 600k          " "" --problem shuttle --sections 200000 --order 4 --steps 100 --warmup 10
run "hs 1M                 " "" --sections 200000 --order 6 --steps 300 --warmup 30
run "d3 4x12.5k n5         " "" --problem delta_iii --sections 3125 --order 5 --steps 300 --warmup 50
run "d3 4x12.5k n4         " "" --problem delta_iii --sections 4167 --order 4 --steps 300 --warmup 50
run "d3 4x12.5k n6         " "" --problem delta_iii --sections 2605 --order 6 --steps 300 --warmup 50
run "d3 ragged 50k         " "" --problem delta_iii --sections 2500 --ragged --steps 200 --warmup 30
run "d3 4x50k n5           " "" --problem delta_iii --sections 12500 --order 5 --steps 100 --warmup 20
run "space station 6k      " "" --problem space_station --sections 2000 --order 4 --steps 500 --warmup 50
