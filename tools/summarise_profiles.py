"""Condense gpurun_out/profiles_<tag>/ (tools/collect_profiles.sh) into profiles/<tag>_*: per-workload kernel
stats CSVs (our kernels only) and one JSON with the HBM traffic per launch.

hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: rocprofv3 reports KB; on gfx950 FETCH_SIZE tallies 128-B read
requests at 64 B (MI355X_MICROARCH.md, section HBM), WRITE_SIZE is exact for 16-B-per-lane streaming stores."""
import csv, glob, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"profiles_{tag}")
dst = os.environ.get("PROFILES_DST") or os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
summary = {"_about": __doc__.split("\n\n")[1].replace("\n", " ")}
for d in sorted(glob.glob(os.path.join(src, "*/"))):
    name = os.path.basename(d.rstrip("/"))
    stats = glob.glob(os.path.join(d, "stats", "**", "*_kernel_stats.csv"), recursive=True)
    if not stats:
        continue
    rows = [r for r in csv.DictReader(open(stats[0])) if r["Name"].startswith("pc_")]
    with open(os.path.join(dst, f"{tag}_{name}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader(); w.writerows(rows)
    entry = {"command": open(os.path.join(src, f"{name}.stats.log")).read().strip().splitlines()[-1][:0] or None}
    log = open(os.path.join(src, f"{name}.stats.log")).read().strip().splitlines()
    bench = next((json.loads(l) for l in reversed(log) if l.startswith("{")), None)
    entry = {"workload": bench["config"]["workload"] if bench else name,
             "algorithmic_bytes": bench["roofline"]["algorithmic_bytes_per_launch"] if bench else None,
             "kernels": {r["Name"]: {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": int(r["MinNs"])} for r in rows}}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        acc = {}
        for f in glob.glob(os.path.join(d, c, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Kernel_Name"].startswith("pc_") and r["Counter_Name"] == c:
                    a = acc.setdefault(r["Kernel_Name"], [0.0, 0])
                    a[0] += float(r["Counter_Value"]); a[1] += 1
        for k, (tot, n) in acc.items():
            entry["kernels"].setdefault(k, {})[c + "_KB"] = round(tot / n, 4)
    for k, v in entry["kernels"].items():
        if "FETCH_SIZE_KB" in v and "WRITE_SIZE_KB" in v:
            v["hbm_bytes_per_launch"] = int(round((2 * v["FETCH_SIZE_KB"] + v["WRITE_SIZE_KB"]) * 1024))
    # the same run's own bench line (HIP events inside the profiled process) and rocprof's durations split by loop:
    # dispatches of the bulk-only loop (what bench.py's roofline times) against those inside evaluations
    if bench:
        entry["bench_under_profiler"] = {"avg_launch_us": bench["roofline"]["avg_launch_us"], "evals_per_s": bench["value"]}
    tr = glob.glob(os.path.join(d, "stats", "**", "*_kernel_trace.csv"), recursive=True)
    if tr:
        kr = [r for r in csv.DictReader(open(tr[0])) if r["Kernel_Name"].startswith("pc_")]
        kr.sort(key=lambda r: int(r["Start_Timestamp"]))
        solo, pair = [], []
        for i, r in enumerate(kr):
            if r["Kernel_Name"].startswith("pc_bulk"):
                nxt = kr[i + 1]["Kernel_Name"] if i + 1 < len(kr) else ""
                (pair if nxt.startswith("pc_tail") else solo).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        if solo and pair:
            entry["rocprof_bulk_split"] = {"bulk_only_loop": {"calls": len(solo), "avg_ns": round(sum(solo) / len(solo), 1)},
                                           "inside_evaluations": {"calls": len(pair), "avg_ns": round(sum(pair) / len(pair), 1)}}
    summary[name] = entry
with open(os.path.join(dst, f"{tag}_pmc_hbm_traffic.json"), "w") as f:
    json.dump(summary, f, indent=1)
print(json.dumps(summary, indent=1)[:3000])
