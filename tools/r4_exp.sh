#!/bin/bash
# experiment lines: ENVV="VAR=.. VAR=.." label args...
cd "$(dirname "$0")/.."
out=${OUT:-gpurun_out/r4_exp.txt}
run() { envv="$1"; label="$2"; shift 2; echo "== $label [$envv]: $*" >> $out; env $envv python bench.py --no-cpu --no-host --no-pin "$@" 2>>gpurun_out/r4_exp.err | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if not l.startswith('{'): continue
    d = json.loads(l); r = d['roofline']; c = d['config']
    print(json.dumps({'us': r['avg_launch_us'], 'frac': r['frac'], 'kernel': r['kernel'], 'tiles': c['tiles'], 'wpt': c['waves_per_tile'], 'lds': c['lds_bytes_per_workgroup'], 'mixed': c.get('mixed_build')}))
" >> $out; }
