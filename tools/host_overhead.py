"""Host cost of one device-resident evaluation call, by layer (kernels return at entry: PYCOLLO_AMD_DBG_STAGE=1)."""
import os, sys, time
os.environ["PYCOLLO_AMD_DBG_STAGE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, torch
from pycollo_amd import problems
from pycollo_amd.engine import NlpEngine
eng = NlpEngine(problems.hypersensitive(K=2000, order=6), device=0)
dev = torch.device("cuda", 0)
x = torch.rand(eng.num_x, dtype=torch.float64, device=dev) - 0.5
lam = torch.randn(eng.num_c, dtype=torch.float64, device=dev)
c = torch.empty(eng.num_c, dtype=torch.float64, device=dev); G = torch.empty(eng.nnz_jac, dtype=torch.float64, device=dev); H = torch.empty(eng.nnz_hess, dtype=torch.float64, device=dev)
s = torch.cuda.Stream(device=dev); st = s.cuda_stream
N = 20000
def timeit(name, fn):
    for _ in range(200): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:55s} {(t1 - t0) / N * 1e6:7.2f} us/call enqueue   ({(t2 - t0) / N * 1e6:7.2f} with drain)")
timeit("python loop + no-arg ctypes call (pc_last_error)", lambda: eng._lib.pc_last_error())
timeit("NlpEngine.evaluate_all_device", lambda: eng.evaluate_all_device(x, 1.0, lam, c, G, H, st))
fn = eng._lib.pc_eval_all_device; h = eng._h
px, pl, pc_, pG, pH = (t.data_ptr() for t in (x, lam, c, G, H))
timeit("pc_eval_all_device, pointers cached", lambda: fn(h, px, 1.0, pl, pc_, pG, pH, st))
timeit("launch_bulk_only (1 launch)", lambda: eng._lib.pc_launch_bulk_device(h, px, pl, pc_, pG, pH, st))
if hasattr(eng, "bind_device"):
    call = eng.bind_device(x, lam, c, G, H, st)
    timeit("bound call", lambda: call(1.0))
