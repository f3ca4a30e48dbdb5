"""PCIe-inclusive rate of the host-pointer API (never the bench value; quoted in DESIGN.md section 4)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pycollo_amd import problems
from pycollo_amd.engine import NlpEngine
eng = NlpEngine(problems.hypersensitive(K=2000, order=6), device=0)
x = np.random.default_rng(0).uniform(-0.4, 0.4, eng.num_x); lam = np.random.default_rng(1).normal(size=eng.num_c)
for name, fn in (("evaluate_all (c+G+H, host pointers)", lambda: eng.evaluate_all(x, 1.0, lam)),
                 ("evaluate_c + G_nonzeros(new_x=False) + H", lambda: (eng.evaluate_c(x), eng.evaluate_G_nonzeros(x, new_x=False), eng.evaluate_H_nonzeros(x, 1.0, lam))),
                 ("evaluate_J", lambda: eng.evaluate_J(x)), ("evaluate_g (dense grad)", lambda: eng.evaluate_g(x))):
    for _ in range(20): fn()
    t0 = time.perf_counter(); n = 500
    for _ in range(n): fn()
    dt = (time.perf_counter() - t0) / n
    print(f"{name:45s} {dt*1e6:8.1f} us/call  {1/dt:9.0f} calls/s")
