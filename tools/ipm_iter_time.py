"""Where an interior-point iteration's wall time goes on the GPU path (config 2 by default): wall-clock sums per
method of GpuInteriorPointSolver, measured by wrapping the methods (no profiler in the way).

    python tools/ipm_iter_time.py [problem] [K] [order]"""
import collections
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pycollo_amd import ipm, problems  # noqa: E402
from pycollo_amd.iteration import MeshIteration  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "hypersensitive"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
order = int(sys.argv[3]) if len(sys.argv) > 3 else 6
acc = collections.defaultdict(lambda: [0.0, 0])
depth = []


def wrap(cls, meth):
    fn = getattr(cls, meth)

    def inner(self, *a, **k):
        t0 = time.perf_counter()
        depth.append(0.0)
        try:
            return fn(self, *a, **k)
        finally:
            dt = time.perf_counter() - t0
            child = depth.pop()
            if depth:
                depth[-1] += dt
            acc[meth][0] += dt - child
            acc[meth][1] += 1
    setattr(cls, meth, inner)


G = ipm.GpuInteriorPointSolver
for m in ("_solve_kkt", "_factor", "_refined_solve", "_ensure_kkt", "_JT", "_J", "_W", "_f", "_c", "_g", "_barrier",
          "_ls_multipliers", "_gn_step", "_alpha_max", "_push_interior"):
    if hasattr(G, m):
        wrap(G, m)
MeshIteration(problems.REGISTRY[name](K=K, order=order), device=0).solve_with_ipm(max_iter=3, tol=1e-8, linear_solver="gpu")
acc.clear()
it = MeshIteration(problems.REGISTRY[name](K=K, order=order), device=0)
t0 = time.perf_counter()
res = it.solve_with_ipm(max_iter=200, tol=1e-8, linear_solver="gpu")
wall = time.perf_counter() - t0
print(f"{name} K={K} n={order}: {res.status}, {res.iterations} iterations, wall {wall * 1e3:.1f} ms = {wall * 1e3 / max(1, res.iterations):.2f} ms per iteration")
tot = 0.0
for k, (s, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print(f"  {k:18s} self {s * 1e3:8.2f} ms over {n:4d} calls ({s * 1e6 / max(1, n):8.1f} us each)")
    tot += s
print(f"  {'(solver loop itself)':18s} {(wall - tot) * 1e3:8.2f} ms")
print("  counters:", {k: v for k, v in res.evaluations.items() if not isinstance(v, dict)})
print("  gpu_seconds:", res.evaluations.get("gpu_seconds"))
print("  loop phases:", res.evaluations.get("phase_seconds"), "solve() total", round(res.seconds, 4))
