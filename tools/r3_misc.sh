#!/bin/bash
# round-3 evidence that is not a bench line: space station stamps + SQ counters, KKT per-kernel table, IPM solve times
export TMPDIR=/tmp PYTHONPATH=$PWD
R=$PWD; O=$R/gpurun_out; mkdir -p $O
# 1. KKT kernels under the profiler (the same script ran unprofiled first so nothing is compiled under the preload)
python3 tools/kkt_time.py > $O/r03_kkt_time.txt 2>&1
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kkt_prof -- python3 $R/tools/kkt_time.py > $O/kkt_prof.log 2>&1)
f=$(find $O/kkt_prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/r03_kkt_hypersensitive10k_kernel_stats.csv
rm -rf $O/kkt_prof
# 2. solve-time tables (GPU KKT vs host SuperLU), config 2 and config 3
python3 tools/solve_time_table.py hypersensitive 2000 6 > $O/r03_solve_time_table.jsonl 2> $O/solve_time.err
python3 tools/solve_time_table.py cart_pole 5000 4 >> $O/r03_solve_time_table.jsonl 2>> $O/solve_time.err
python3 tools/ipm_iter_time.py > $O/r03_ipm_iter_time.txt 2>&1
python3 tools/ipm_iter_time.py cart_pole 5000 4 >> $O/r03_ipm_iter_time.txt 2>&1
# 3. space station: SQ counters, then the clock stamps of a PC_STAMPS build
if [ -z "$SKIP_STATION" ]; then
tools/pmc_sq.sh gpurun_out/sq_station --problem space_station --sections 2000 --order 4 --steps 300 --warmup 30 > $O/r03_sq_space_station6k.txt 2>&1
PYCOLLO_AMD_DEFINES=PC_STAMPS python3 bench.py --build-only --problem space_station --sections 2000 --order 4 > /dev/null 2>&1
PYCOLLO_AMD_DEFINES=PC_STAMPS python3 tools/stamps.py --problem space_station --sections 2000 --order 4 > $O/r03_stamps_space_station.txt 2>&1
fi
echo done
