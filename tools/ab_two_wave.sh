#!/bin/bash
# A/B for the heavy multi-state kernels: two waves per SIMD resident (<= 256 VGPRs, <= 40 KB LDS per 2-wave tile)
# against the default one-wave-per-SIMD launch.  usage: tools/ab_two_wave.sh [reps]
# Variants are (TILE_NODES, WPT, DEFINES); code objects must have been built beforehand (bench.py --build-only with
# the same environment) so that the box does not spend its time in hipcc.
fmt='import sys,json; d=json.loads(sys.stdin.read()); print("tiles", d["config"].get("tiles"), "| TB", d["config"]["threads_per_block"], "x", d["config"]["waves_per_tile"], "| device us", round(d["device_ms_per_step"]*1e3,2), "| bulk us", d["roofline"]["avg_launch_us"], "| frac", round(d["roofline"]["frac"],3))'
REPS=${1:-2}
run() {  # label, tile_nodes, wpt, defines, bench args...
  local label=$1 tn=$2 wpt=$3 defs=$4; shift 4
  for rep in $(seq $REPS); do
    echo -n "[$label] "
    PYCOLLO_AMD_TILE_NODES=$tn PYCOLLO_AMD_WPT=$wpt PYCOLLO_AMD_DEFINES="$defs" timeout -k 10 300 python bench.py --no-cpu --no-host "$@" 2>/dev/null | python3 -c "$fmt" || echo failed
  done
}
D3="--problem delta_iii --sections 3125 --order 5 --steps 300 --warmup 50"
run "d3 default            " "" "" "" $D3
run "d3 split W1           " "" 1 "PC_SPLIT_MIN=40" $D3
run "d3 split W2 tn57      " 57 2 "PC_SPLIT_MIN=40" $D3
run "d3 split W2s tn57     " 57 2 "PC_STATIC_W=2 PC_SPLIT_MIN=40" $D3
run "d3 split W2s tn53     " 53 2 "PC_STATIC_W=2 PC_SPLIT_MIN=40" $D3
run "d3 split W2s tn64     " "" 2 "PC_STATIC_W=2 PC_SPLIT_MIN=40" $D3
run "d3 nosplit W2s tn57   " 57 2 "PC_STATIC_W=2" $D3
