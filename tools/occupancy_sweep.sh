fmt='import sys,json; d=json.loads(sys.stdin.read()); print(d["config"]["workload"][:60], "| evals/s", d["value"], "| TB", d["config"]["threads_per_block"], "x", d["config"].get("waves_per_tile"), "| bulk us", d["roofline"]["avg_launch_us"], "| GB/s", d["roofline"]["achieved"])'
for cfg in "--problem shuttle --sections 200000 --order 4 --steps 100 --tpb 64" "--problem shuttle --sections 20000 --order 4 --steps 300 --tpb 64" "--problem cart_pole --sections 100000 --order 4 --steps 200 --tpb 128" "--problem delta_iii --sections 31250 --order 5 --steps 50 --tpb 64"; do
for w in 0 3 4; do
  echo -n "waves_per_eu=$w "
  PYCOLLO_AMD_WAVES_PER_EU=$w timeout -k 10 300 python bench.py --no-cpu $cfg 2>/dev/null | python3 -c "$fmt" || echo "failed: $cfg"
done; done
