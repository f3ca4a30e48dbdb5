#!/bin/bash
# the measured choice of the launching core: processes started on given cores (PYCOLLO_AMD_PIN_CPU) and wherever the scheduler put them
fmt='import sys,json; d=json.loads(sys.stdin.read()); print("cpu", d["config"]["launch_thread_cpu"], "| pin survived", d["config"].get("pin_survived_hip_init"), "| probe", d["config"].get("launch_core_probe"), "| evals/s", d["value"], "| device", d["device_ms_per_step"])'
for c in 121 "" "" "" "" "" "" "" "" "" "" "" "" "" "" ""; do
  echo -n "[start ${c:-any}] "; PYCOLLO_AMD_PIN_CPU=$c timeout -k 10 200 python bench.py --no-cpu 2>/dev/null | python3 -c "$fmt"
done
