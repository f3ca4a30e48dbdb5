#!/bin/bash
# the headline model over N = 1e4 .. 1e7 collocation nodes (uniform mesh, order 6)
fmt='import sys,json; d=json.loads(sys.stdin.read()); c=d["config"]; r=d["roofline"]; print(c["workload"].split("=")[1].strip(), "| tiles", c["tiles"], "x", c["threads_per_block"], "| evals/s", d["value"], "| eval us", round(d["device_ms_per_step"]*1e3,2), "| bulk us", r["avg_launch_us"], "| GB/s", r["achieved"], "| frac", r["frac"])'
for cfg in "2000 20000" "20000 5000" "200000 500" "2000000 60"; do
  set -- $cfg
  timeout -k 10 400 python bench.py --no-cpu --no-host --sections $1 --order 6 --steps $2 --warmup $(( $2 / 10 )) 2>/dev/null | python3 -c "$fmt" || echo "failed: $cfg"
done
