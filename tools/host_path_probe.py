"""Where the host-pointer callback's time goes: in-place pc_eval_all (pinned blocks, no host memcpy) over a range of
mesh sizes and data-movement modes; a straight-line fit gives the latency floor (launch + completion) and the
effective PCIe rate of each mode.  Usage (GPU box): python tools/host_path_probe.py [problem]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from pycollo_amd import problems  # noqa: E402
from pycollo_amd.engine import NlpEngine  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "hypersensitive"
order = 6 if name == "hypersensitive" else 4
rows = []
for K in (20, 200, 1000, 2000, 4000, 8000):
    eng = NlpEngine(problems.REGISTRY[name](K=K, order=order), device=0)
    hx, hl, hc, hG, hH = eng.host_buffers()
    rng = np.random.default_rng(0)
    hx[:] = rng.uniform(0.05, 0.3, eng.num_x)
    hl[:] = rng.normal(size=eng.num_c)
    nbytes_out = 8 * (eng.num_c + eng.nnz_jac + eng.nnz_hess)
    nbytes_in = 8 * (eng.num_x + eng.num_c)
    line = [K, nbytes_in, nbytes_out]
    for mode in (0, 1, 2, 3):
        eng.set_host_mode(mode)
        for _ in range(50):
            eng.evaluate_all_inplace(1.0)
        ts = np.empty(1000)
        for i in range(1000):
            t0 = time.perf_counter()
            eng.evaluate_all_inplace(1.0)
            ts[i] = time.perf_counter() - t0
        line.append(float(np.median(ts) * 1e6))
    rows.append(line)
    print("K=%5d in=%8d B out=%9d B  us by mode 0..3: %7.1f %7.1f %7.1f %7.1f" % tuple(line), flush=True)
    eng.close()
a = np.array(rows)
for m in range(4):
    slope, icpt = np.polyfit(a[:, 2] / 1e6, a[:, 3 + m], 1)
    print(f"mode {m}: floor {icpt:6.1f} us + {slope:6.2f} us per MB out  (= {1e3 / slope / 1e3 * 1e3:7.1f} MB/ms = {1 / slope * 1e3 / 1e3:6.1f} GB/s)")
