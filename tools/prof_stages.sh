#!/bin/bash
# usage: tools/prof_stages.sh <outdir> <bench args...> ; runs rocprofv3 kernel-trace for each diagnostic stage
export TMPDIR=/tmp
R=$PWD
OUT=$R/$1; shift
mkdir -p $OUT
cd /tmp
for st in ${STAGES:-1 2 3 4 5 6 0}; do
  PYCOLLO_AMD_DBG_STAGE=$st timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st$st -- python3 $R/bench.py --no-cpu "$@" > $OUT/st$st.log 2>&1 || exit 1
  echo "stage $st: $(grep pc_bulk $OUT/st$st/*/*_kernel_stats.csv | cut -d, -f1-4,6-7)  tail: $(grep pc_tail $OUT/st$st/*/*_kernel_stats.csv | cut -d, -f4)"
done
