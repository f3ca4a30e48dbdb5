#!/bin/bash
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(d["config"]["workload"][:60], "| evals/s", d["value"], "| TB", d["config"]["threads_per_block"], "x", d["config"]["waves_per_tile"], "| bulk us", d["roofline"]["avg_launch_us"], "| GB/s", d["roofline"]["achieved"])'
for cfg in "--problem shuttle --sections 200000 --order 4 --steps 100" "--problem shuttle --sections 20000 --order 4 --steps 300" "--problem shuttle --sections 2000 --order 4 --steps 1000" "--problem delta_iii --sections 3125 --order 5 --steps 300" "--problem delta_iii --sections 12500 --order 5 --steps 100" "--problem space_station --sections 20000 --order 4 --steps 100"; do
for d in "PC_SPLIT_MIN=100000" ""; do
  echo -n "[${d:-split}] "
  PYCOLLO_AMD_DEFINES="$d" timeout -k 10 300 python bench.py --no-cpu $cfg 2>/dev/null | python3 -c "$fmt" || echo failed
done; done
echo -n "[split, waves_per_eu=3] "; PYCOLLO_AMD_WAVES_PER_EU=3 timeout -k 10 300 python bench.py --no-cpu --problem shuttle --sections 200000 --order 4 --steps 100 2>/dev/null | python3 -c "$fmt"
echo -n "[split, waves_per_eu=3] "; PYCOLLO_AMD_WAVES_PER_EU=3 timeout -k 10 300 python bench.py --no-cpu --problem shuttle --sections 20000 --order 4 --steps 300 2>/dev/null | python3 -c "$fmt"
