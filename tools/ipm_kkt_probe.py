import sys, time, os
sys.path.insert(0, '/root/repo')
import numpy as np
from pycollo_amd import problems, kkt as kk
from pycollo_amd.iteration import MeshIteration
log = []
for name in ("solve", "matvec", "factor"):
    orig = getattr(kk.GpuKkt, name)
    def wrap(self, *a, _o=orig, _n=name, **k):
        t = time.perf_counter(); r = _o(self, *a, **k); log.append((_n, time.perf_counter() - t)); return r
    setattr(kk.GpuKkt, name, wrap)
it = MeshIteration(problems.hypersensitive(K=2000, order=6), device=0)
res = it.solve_with_ipm(max_iter=200, tol=1e-8, linear_solver="gpu")
import collections
d = collections.defaultdict(list)
for n, t in log: d[n].append(t)
for n, ts in d.items():
    ts = np.array(ts) * 1e3
    print(n, len(ts), "median", round(float(np.median(ts)), 3), "mean", round(float(ts.mean()), 3), "max", round(float(ts.max()), 3), "first5", np.round(ts[:5], 2))
print("sorted solve", np.round(np.sort(np.array(d["solve"]) * 1e3), 1))
