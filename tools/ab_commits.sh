fmt='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["device_ms_per_step"], d["roofline"]["avg_launch_us"])'
for rep in 1 2 3 4; do
for c in ab/62f0351 ab/559d67e .; do
  echo -n "[$c] "; (cd $c && timeout -k 10 300 python bench.py --no-cpu 2>/dev/null | python3 -c "$fmt")
done; done
