#!/bin/bash
# usage: DEFS="A=1|A=2 B=3|..." tools/define_sweep.sh   -- bench lines per -D set (sets separated by '|'; '-' = none)
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(d["config"]["workload"][:52], "| evals/s", d["value"], "| TB", d["config"]["threads_per_block"], "x", d["config"].get("waves_per_tile"), "| bulk us", d["roofline"]["avg_launch_us"], "| GB/s", d["roofline"]["achieved"])'
IFS='|' read -ra SETS <<< "${DEFS:--}"
for cfg in "--steps 2000" "--sections 200000 --order 6 --steps 300" "--problem cart_pole --sections 5000 --order 4 --steps 1000" "--problem shuttle --sections 2000 --order 4 --steps 1000" "--problem shuttle --sections 20000 --order 4 --steps 300" "--problem shuttle --sections 200000 --order 4 --steps 100" "--problem delta_iii --sections 3125 --order 5 --steps 300"; do
  for d in "${SETS[@]}"; do
    [ "$d" = "-" ] && d=""
    echo -n "[${d:-default}] "
    PYCOLLO_AMD_DEFINES="$d" timeout -k 10 300 python bench.py --no-cpu $cfg 2>/dev/null | python3 -c "$fmt" || echo "failed: $cfg"
  done
done
