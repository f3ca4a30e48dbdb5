fmt='import sys,json; d=json.loads(sys.stdin.read()); print(d["config"]["workload"][:40], "| evals/s", d["value"], "| bulk us", d["roofline"]["avg_launch_us"])'
for rep in 1 2 3; do
for d in "" "PC_HOIST_MAX=0" "PC_PIN_BUDGET=0" "PC_PIN_BUDGET=400"; do
  echo -n "[${d:-default}] "
  PYCOLLO_AMD_DEFINES="$d" timeout -k 10 300 python bench.py --no-cpu --steps 3000 2>/dev/null | python3 -c "$fmt" || echo failed
done; done
