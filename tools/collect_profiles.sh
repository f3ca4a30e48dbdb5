#!/bin/bash
# usage: tools/collect_profiles.sh <tag>      (on the GPU box; writes gpurun_out/profiles_<tag>/)
# For each workload: rocprofv3 --kernel-trace --stats of the bench command, then FETCH_SIZE and WRITE_SIZE in two
# separate --pmc passes (kernel-trace only, as the MI355X guide prescribes).  Summaries are collected by
# tools/summarise_profiles.py into profiles/.
export TMPDIR=/tmp
R=$PWD; TAG=${1:-r01}; OUT=$R/gpurun_out/profiles_$TAG; mkdir -p $OUT; cd /tmp
run() {  # name, bench args...
  local name=$1; shift
  python3 $R/bench.py --build-only "$@" > /dev/null || { echo "$name build failed"; return 1; }   # never under the profiler
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name/stats -- python3 $R/bench.py --no-cpu --no-host "$@" > $OUT/$name.stats.log 2>&1 || { echo "$name stats failed"; return 1; }
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$name/$c -- python3 $R/bench.py --no-cpu --no-host "$@" > $OUT/$name.$c.log 2>&1 || { echo "$name $c failed"; return 1; }
  done
  echo "$name done: $(tail -1 $OUT/$name.stats.log | cut -c1-160)"
}
# usage: tools/collect_profiles.sh <tag> [part]   part 1: configs 2-4 and the size extremes; part 2: the Delta III family
PART=${2:-all}
if [ "$PART" = all ] || [ "$PART" = 1 ]; then
run hypersensitive10k --steps 3000 --warmup 300 &&
run hypersensitive1M --sections 200000 --steps 50 --warmup 10 &&
run cart_pole15k --problem cart_pole --sections 5000 --order 4 --steps 1000 --warmup 100 &&
run shuttle60k --problem shuttle --sections 20000 --order 4 --steps 300 --warmup 30 &&
run shuttle600k --problem shuttle --sections 200000 --order 4 --steps 50 --warmup 10 &&
run shuttle6k --problem shuttle --sections 2000 --order 4 --steps 1000 --warmup 100 &&
run space_station6k --problem space_station --sections 2000 --order 4 --steps 300 --warmup 30 ;
fi
if [ "$PART" = all ] || [ "$PART" = 2 ]; then
run delta_iii12k --problem delta_iii --sections 3125 --order 5 --steps 100 --warmup 10 &&
run delta_iii12k_n4 --problem delta_iii --sections 4167 --order 4 --steps 100 --warmup 10 &&
run delta_iii50k --problem delta_iii --sections 12500 --order 5 --steps 50 --warmup 10 &&
run delta_iii_ragged50k --problem delta_iii --sections 2500 --ragged --steps 50 --warmup 10 &&
run delta_iii_refined50k --problem delta_iii --refined 12500 --steps 50 --warmup 10 &&
run delta_iii12k_n8 --problem delta_iii --sections 1785 --order 8 --steps 100 --warmup 10 ;
fi
if [ "$PART" = 3 ]; then
run space_station6k --problem space_station --sections 2000 --order 4 --steps 300 --warmup 30 ;
fi
# the raw traces are tens of MB per workload: condense on the box, ship only the summaries (copy them to profiles/)
cd $R && PROFILES_DST=$R/gpurun_out/profiles_out_$PART python3 tools/summarise_profiles.py $TAG > /dev/null && rm -rf $OUT && ls $R/gpurun_out/profiles_out_$PART
