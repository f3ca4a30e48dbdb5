#!/bin/bash
# usage: tools/pmc_write.sh <outdir> <bench args...>   -- WRITE_SIZE per dispatch of the bulk kernels
export TMPDIR=/tmp
R=$PWD; OUT=$R/$1; shift; mkdir -p $OUT; cd /tmp
# the code object is built in a plain process first: no compiler may be spawned under the profiler preload
python3 $R/bench.py --build-only "$@" > /dev/null || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT -- python3 $R/bench.py --no-cpu --no-host "$@" > $OUT/log.txt 2>&1 || { tail -3 $OUT/log.txt; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("pc_bulk"):
            a = acc[r["Kernel_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
print({k: round(v[0] / v[1] / 1024, 3) for k, v in acc.items()}, "MB written per launch")
PY
grep -o '"nnz_jac": [0-9]*, "nnz_hess": [0-9]*' $OUT/log.txt | tail -1
