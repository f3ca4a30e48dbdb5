#!/bin/bash
# rocprofv3 kernel stats of the default bench (bulk and tail durations)
export TMPDIR=/tmp
R=$PWD; OUT=$R/${1:-gpurun_out/tail_time}; mkdir -p $OUT; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --no-cpu --steps 3000 --warmup 300 > $OUT/bench.log 2>&1 || { tail -5 $OUT/bench.log; exit 1; }
tail -1 $OUT/bench.log | cut -c1-200
for f in $OUT/*/*_kernel_stats.csv; do head -6 $f | cut -d, -f1-8; done
