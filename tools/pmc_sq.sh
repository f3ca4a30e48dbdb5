#!/bin/bash
# usage: tools/pmc_sq.sh <outdir> <bench args...>
# SQ counter passes (no trace domains besides --kernel-trace) for the bulk kernel; prints per-dispatch averages.
export TMPDIR=/tmp
R=$PWD
OUT=$R/$1; shift
mkdir -p $OUT
cd /tmp
# the code object is built in a plain process first: no compiler may be spawned under the profiler preload
python3 $R/bench.py --build-only "$@" > /dev/null || exit 1
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SMEM"
P3="SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT"
# instruction fetch: straight-line kernels of tens of KB per wave may wait on the instruction cache (names differ by
# rocprofv3 build; a pass with an unknown counter fails by itself and the others still count)
P4="SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_IFETCH SQ_INSTS_VALU"
P5="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE"
P6="SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES SQ_WAVE_CYCLES"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5" "$P6"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/p$i -- python3 $R/bench.py --no-cpu --no-host "$@" > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; continue; }
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if not row["Kernel_Name"].startswith("pc_bulk"):
            continue
        a = acc[row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    print(f"{k:28s} {acc[k][0] / acc[k][1]:16.1f}  (avg of {acc[k][1]} dispatches)")
PY
# the raw per-dispatch CSVs are tens of MB: gpurun_out/ only travels back below 64 MiB
rm -rf $OUT/p[0-9] $OUT/p[0-9].log
