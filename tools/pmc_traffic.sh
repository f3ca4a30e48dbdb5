#!/bin/bash
# usage: tools/pmc_traffic.sh <outdir> <bench args...>  -- FETCH_SIZE and WRITE_SIZE per bulk launch (two passes)
export TMPDIR=/tmp
R=$PWD; OUT=$R/$1; shift; mkdir -p $OUT; cd /tmp
# the code object is built in a plain process first: no compiler may be spawned under the profiler preload
python3 $R/bench.py --build-only "$@" > /dev/null || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -- python3 $R/bench.py --no-cpu --no-host "$@" > $OUT/$c.log 2>&1 || { tail -3 $OUT/$c.log; exit 1; }
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("pc_bulk"):
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
fs, ws = (acc[k][0] / max(1, acc[k][1]) for k in ("FETCH_SIZE", "WRITE_SIZE"))
print(f"FETCH_SIZE {fs:.1f} KB  WRITE_SIZE {ws:.1f} KB  hbm bytes {(2 * fs + ws) * 1024:.0f}")
PY
