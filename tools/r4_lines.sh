#!/bin/bash
# round-4 bench lines on one box: the refined-mesh (mixed build) workloads beside their any-order and uniform baselines
cd "$(dirname "$0")/.."
out=gpurun_out/r4_lines.txt
: > $out
run() { label="$1"; shift; echo "== $label: $*" >> $out; env $ENVV python bench.py --no-cpu --no-host --no-pin "$@" 2>>gpurun_out/r4_lines.err | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if not l.startswith('{'): continue
    d = json.loads(l); r = d['roofline']; c = d['config']
    print(json.dumps({'us': r['avg_launch_us'], 'frac': r['frac'], 'GBs': r['achieved'], 'bytes': r['algorithmic_bytes_per_launch'], 'kernel': r['kernel'], 'tiles': c['tiles'], 'wpt': c['waves_per_tile'], 'lds': c['lds_bytes_per_workgroup'], 'mixed': c.get('mixed_build'), 'batches': r['launch_us_batches'], 'workload': c['workload']}))
" >> $out; }
ENVV=""
run "d3 refined 4x12.5k mixed" --problem delta_iii --refined 12500 --steps 200 --warmup 30
run "d3 refined 4x12.5k any-order" --problem delta_iii --refined 12500 --generic --steps 200 --warmup 30
run "d3 ragged 50k" --problem delta_iii --sections 2500 --ragged --steps 200 --warmup 30
run "d3 uniform n5 (config 5)" --problem delta_iii --sections 3125 --order 5 --steps 300 --warmup 30
run "hypersensitive refined 30k mixed" --problem hypersensitive --refined 30000 --steps 2000 --warmup 100
run "hypersensitive refined 30k any-order" --problem hypersensitive --refined 30000 --generic --steps 2000 --warmup 100
run "headline" --steps 20000
cat $out
