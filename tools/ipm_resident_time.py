"""Wall time per interior-point iteration: the host-vector loop over the GPU factorisation ("gpu") against the
device-resident iteration ("resident"), same NLP, same iterates.

    python tools/ipm_resident_time.py [problem] [K] [order] [gpu,resident]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pycollo_amd import problems  # noqa: E402
from pycollo_amd.iteration import MeshIteration  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "hypersensitive"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
order = int(sys.argv[3]) if len(sys.argv) > 3 else 6
modes = sys.argv[4].split(",") if len(sys.argv) > 4 else ["gpu", "resident"]
for ls in modes:
    MeshIteration(problems.REGISTRY[name](K=K, order=order), device=0).solve_with_ipm(max_iter=3, tol=1e-8, linear_solver=ls)   # warm-up
    for rep in range(2):
        it = MeshIteration(problems.REGISTRY[name](K=K, order=order), device=0)
        t0 = time.perf_counter()
        res = it.solve_with_ipm(max_iter=300, tol=1e-8, linear_solver=ls)
        wall = time.perf_counter() - t0
        ph = res.evaluations.get("phase_seconds", {})
        loop = res.seconds - ph.get("setup", 0.0)
        print(f"{name} K={K} n={order} [{ls}]: {res.status}, {res.iterations} iterations, objective {res.objective:.10g}; solve() {res.seconds * 1e3:.1f} ms "
              f"(set-up {ph.get('setup', 0) * 1e3:.1f} ms), loop {loop * 1e3 / max(1, res.iterations):.3f} ms per iteration; "
              f"phases {{{', '.join(f'{k}: {v * 1e3:.1f}' for k, v in ph.items())}}} ms; "
              f"factorisations {res.evaluations.get('factorisations')}, trial points {res.evaluations.get('constraints')}", flush=True)
