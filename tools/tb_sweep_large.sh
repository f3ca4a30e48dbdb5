fmt='import sys,json; d=json.loads(sys.stdin.read()); print(d["config"]["workload"][:60], "| evals/s", d["value"], "| TB", d["config"]["threads_per_block"], "x", d["config"].get("waves_per_tile"), "| bulk us", d["roofline"]["avg_launch_us"], "| GB/s", d["roofline"]["achieved"])'
for cfg in "--problem shuttle --sections 200000 --order 4 --steps 100" "--problem cart_pole --sections 100000 --order 4 --steps 200" "--problem shuttle --sections 60000 --order 4 --steps 100"; do
for t in 64 128 256; do
  PYCOLLO_AMD_WPT=1 timeout -k 10 200 python bench.py --no-cpu $cfg --tpb $t 2>/dev/null | python3 -c "$fmt" || echo "failed: $cfg"
done; done
