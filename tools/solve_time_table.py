"""Solve-time table for SURVEY.md section 8f row N4: one mesh iteration of an NLP at a given size, solved twice with
the same interior-point algorithm -- KKT systems on the GPU (pc_kkt_*, G~ / H~ device-resident) and on the host
(SuperLU on arrays the callbacks copied down) -- with the time split into callbacks and linear algebra.

    python tools/solve_time_table.py [problem] [K] [order]      (GPU box; prints one JSON line per run)
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from pycollo_amd import problems  # noqa: E402
from pycollo_amd.iteration import MeshIteration  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "hypersensitive"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
order = int(sys.argv[3]) if len(sys.argv) > 3 else 6
max_iter = int(os.environ.get("MAX_ITER", "200"))
for ls in os.environ.get("LINEAR_SOLVERS", "resident,gpu,host").split(","):
    it = MeshIteration(problems.REGISTRY[name](K=K, order=order), device=0)
    t0 = time.perf_counter()
    res = it.solve_with_ipm(max_iter=max_iter, tol=1e-8, linear_solver=ls)
    wall = time.perf_counter() - t0
    ev = res.evaluations
    row = {"problem": name, "K": K, "order": order, "nodes": int(sum(pl.N for pl in it.layout.phases)),
           "num_x": it.engine.num_x, "num_c": it.engine.num_c, "linear_solver": ls, "status": res.status,
           "objective": float(it.objective), "nlp_iterations": int(res.iterations), "factorisations": ev["factorisations"],
           "wall_s": round(wall, 3), "ms_per_iteration": round(1e3 * wall / max(1, res.iterations), 2),
           "kkt_s": round(ev["kkt_seconds"], 3), "kkt_ms_per_iteration": round(1e3 * ev["kkt_seconds"] / max(1, res.iterations), 2)}
    for key in ("kkt_solves", "refined_solves", "jacobian", "hessian", "constraints"):
        if key in ev:
            row["n_" + key] = ev[key]
    if "phase_seconds" in ev:
        # the part before the first iteration (starting point, scaling, least-squares multipliers and, on the GPU path,
        # the one-off build of the KKT tables for this mesh) apart from the iterations themselves
        setup = ev["phase_seconds"]["setup"]
        row["setup_s"] = round(setup, 4)
        row["loop_ms_per_iteration"] = round(1e3 * (res.seconds - setup) / max(1, res.iterations), 2)
    if "gpu_seconds" in ev:
        row["gpu_seconds"] = {k: round(v, 4) for k, v in ev["gpu_seconds"].items()}
        row["ms_per_factorisation"] = round(1e3 * ev["gpu_seconds"]["factor"] / max(1, ev["factorisations"]), 3)
    print(json.dumps(row), flush=True)
    it.engine.close()
