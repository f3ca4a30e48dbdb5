"""Development aid: does an idle gap between calls change what pc_kkt_factor / pc_kkt_solve cost?  (The interior-point loop
calls them with host work in between and sees several times the back-to-back figures of tools/kkt_time.py.)"""
import os, sys, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'tests'))
import numpy as np
from test_kkt_cpu import kkt_case
from pycollo_amd.kkt import GpuKkt
eng, ora, x, lam, ineq, fixed, sc, dvec = kkt_case("hypersensitive", dict(K=2000, order=6), device=0)
eng.evaluate_resident(x, 1.0, lam)
k = GpuKkt(eng, ineq, fixed, sc)
rhs = np.random.default_rng(0).normal(size=k.nu)
for gap in (0.0, 0.0002, 0.001, 0.005, 0.02, 0.1):
    for name, fn in (("factor", lambda: k.factor(dvec)), ("solve", lambda: k.solve(rhs)), ("matvec", lambda: k.matvec(dvec, rhs))):
        fn(); ts = []
        for _ in range(8):
            time.sleep(gap)
            t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
        print("gap ms", 1e3 * gap, name, "median ms", round(1e3 * float(np.median(ts)), 3), "min", round(1e3 * min(ts), 3), flush=True)
# host work (numpy on 30 k-vectors) instead of sleep
for name, fn in (("factor", lambda: k.factor(dvec)), ("solve", lambda: k.solve(rhs))):
    ts = []
    for _ in range(8):
        a = np.random.default_rng(1).normal(size=(200, 30000)); a = a @ a[0]     # ~10 ms of host work
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    print("host-work gap", name, "median ms", round(1e3 * float(np.median(ts)), 3), flush=True)
k.close()
