#!/bin/bash
# round-4 bench lines on one box (profiles/r04_bench_lines.txt): every workload of DESIGN.md section 3's table
cd "$(dirname "$0")/.."
out=gpurun_out/r4_lines.txt
: > $out
run() { label="$1"; shift; env $ENVV python bench.py --no-cpu --no-host --no-pin "$@" 2>>gpurun_out/r4_lines.err | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if not l.startswith('{'): continue
    d = json.loads(l); r = d['roofline']; c = d['config']
    m = c.get('mixed_build')
    print('[%-34s] tiles %5d | TB %d x %d | %-18s | device us %8.3f | frac %.4f | %6.1f MB | LDS %6d B%s' % ('$label', c['tiles'], c['threads_per_block'], c['waves_per_tile'], r['kernel'], r['avg_launch_us'], r['frac'], r['algorithmic_bytes_per_launch'] / 1e6, c['lds_bytes_per_workgroup'], (' | order-pure tiles %d, any-order %d, bodies %s' % (m['order_pure_tiles'], m['any_order_tiles'], m['specialised_orders'])) if m else ''))
" >> $out; }
ENVV=""
run "config 2 hypersensitive 10k" --steps 20000
run "hypersensitive 1M" --sections 200000 --steps 300 --warmup 30
run "config 3 cart-pole 15k" --problem cart_pole --sections 5000 --order 4 --steps 5000 --warmup 300
run "shuttle 6k" --problem shuttle --sections 2000 --order 4 --steps 2000 --warmup 100
run "config 4 shuttle 60k" --problem shuttle --sections 20000 --order 4 --steps 500 --warmup 50
run "shuttle 600k" --problem shuttle --sections 200000 --order 4 --steps 60 --warmup 10
run "space station 6k" --problem space_station --sections 2000 --order 4 --steps 300 --warmup 30
for n in 4 5 6 7 8 9; do
  K=$(( 12500 / (n - 1) ))
  run "d3 4x12.5k order $n" --problem delta_iii --sections $K --order $n --steps 200 --warmup 30
done
run "d3 4x50k order 5" --problem delta_iii --sections 12500 --order 5 --steps 100 --warmup 20
run "d3 ph-refined 50k, mixed build" --problem delta_iii --refined 12500 --steps 200 --warmup 30
run "d3 ph-refined 50k, any-order" --problem delta_iii --refined 12500 --generic --steps 200 --warmup 30
run "d3 random orders 50k (any-order)" --problem delta_iii --sections 2500 --ragged --steps 200 --warmup 30
run "hs ph-refined 30k, mixed build" --problem hypersensitive --refined 30000 --steps 2000 --warmup 100
run "hs ph-refined 30k, any-order" --problem hypersensitive --refined 30000 --generic --steps 2000 --warmup 100
cat $out
