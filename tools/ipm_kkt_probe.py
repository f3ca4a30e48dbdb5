"""Development aid: wall time of every GpuKkt call inside one interior-point solve at config-2 size, and the host-side
gaps between them (where a solve's time goes once the kernels take a few hundred microseconds)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pycollo_amd import problems, kkt as kk, ipm
from pycollo_amd.iteration import MeshIteration
log = []
for name in ("solve", "matvec", "factor"):
    orig = getattr(kk.GpuKkt, name)
    def wrap(self, *a, _o=orig, _n=name, **k):
        t = time.perf_counter(); r = _o(self, *a, **k); log.append((_n, t, time.perf_counter())); return r
    setattr(kk.GpuKkt, name, wrap)
from pycollo_amd import engine as eng_mod
for cls, names in ((eng_mod.NlpEngine, ("evaluate_resident", "evaluate_J", "evaluate_g", "evaluate_c", "evaluate_G_nonzeros", "evaluate_H_nonzeros")),
                   (eng_mod.PycolloGpuProblem, ("objective", "gradient", "constraints", "jacobian", "hessian"))):
    for name in names:
        if not hasattr(cls, name):
            continue
        orig = getattr(cls, name)
        def wrap(self, *a, _o=orig, _n=cls.__name__[:3] + "." + name, **k):
            t = time.perf_counter(); r = _o(self, *a, **k); log.append((_n, t, time.perf_counter())); return r
        setattr(cls, name, wrap)
orig_rs = ipm.GpuInteriorPointSolver._refined_solve
def rs(self, *a, **k):
    t = time.perf_counter(); r = orig_rs(self, *a, **k); log.append(("REFINED", t, time.perf_counter())); return r
ipm.GpuInteriorPointSolver._refined_solve = rs
it = MeshIteration(problems.hypersensitive(K=2000, order=6), device=0)
t0 = time.perf_counter()
res = it.solve_with_ipm(max_iter=200, tol=1e-8, linear_solver="gpu")
print("wall", round(time.perf_counter() - t0, 3), res.status, res.iterations)
import collections
d = collections.defaultdict(list)
for n, a, b in log: d[n].append(b - a)
for n, ts in d.items():
    if n == "REFINED":
        continue
    ts = np.array(ts) * 1e3
    print(n, len(ts), "median", round(float(np.median(ts)), 3), "mean", round(float(ts.mean()), 3), "max", round(float(ts.max()), 3), "sum", round(float(ts.sum()), 1))
outer = {"REFINED", "Pyc.objective", "Pyc.gradient", "Pyc.constraints", "Pyc.jacobian", "Pyc.hessian"}
calls = sorted([x for x in log if x[0] not in outer], key=lambda x: x[1])
gaps = [(calls[i + 1][1] - calls[i][2], calls[i][0], calls[i + 1][0]) for i in range(len(calls) - 1)]
gaps.sort(reverse=True)
print("largest host gaps between KKT calls (ms, after, before):", [(round(1e3 * g, 2), a, b) for g, a, b in gaps[:12]])
print("sum of gaps ms", round(1e3 * sum(g for g, _, _ in gaps), 1))
