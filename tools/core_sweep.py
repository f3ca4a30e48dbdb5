"""Within ONE process: move the launching thread from core to core and time the headline evaluation loop on each.
Separates 'which core launches' from per-process effects (tools/core_probe.sh saw 95k / 128k between processes)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
PIN_FIRST = os.environ.get("PIN_FIRST")          # pin before HIP initialises (helpers inherit) then sweep anyway
allowed = sorted(os.sched_getaffinity(0))
if PIN_FIRST:
    os.sched_setaffinity(0, {int(PIN_FIRST)})
import torch
from pycollo_amd import problems
from pycollo_amd.engine import NlpEngine
dev = torch.device("cuda", 0)
eng = NlpEngine(problems.hypersensitive(K=2000, order=6), device=0)
x = torch.from_numpy(np.random.default_rng(1234).uniform(-0.45, 0.45, eng.num_x)).to(dev)
lam = torch.from_numpy(np.random.default_rng(1235).normal(size=eng.num_c)).to(dev)
c = torch.empty(eng.num_c, dtype=torch.float64, device=dev); G = torch.empty(eng.nnz_jac, dtype=torch.float64, device=dev)
H = torch.empty(eng.nnz_hess, dtype=torch.float64, device=dev)
ts = torch.cuda.Stream(device=dev); torch.cuda.set_stream(ts)
step = eng.bind_device(x, lam, c, G, H, ts.cuda_stream)
def rate(n=4000):
    for _ in range(500): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return n / (time.perf_counter() - t0)
stride = int(os.environ.get("STRIDE", "4"))
print("started on cpu", os.sched_getcpu() if hasattr(os, "sched_getcpu") else -1, "pin_first", PIN_FIRST, flush=True)
res = []
for cpu in allowed[::stride]:
    os.sched_setaffinity(0, {cpu})
    r = rate(); res.append((cpu, r))
    print(f"cpu {cpu:3d} (ccd {cpu % 128 // 8:2d}{' smt' if cpu >= 128 else ''}): {r / 1e3:7.1f}k evals/s", flush=True)
res.sort(key=lambda t: -t[1])
print("best", res[:5]); print("worst", res[-5:])
# again on the best and the worst core: is it the core or the moment?
for cpu in (res[0][0], res[-1][0], res[0][0], res[-1][0]):
    os.sched_setaffinity(0, {cpu}); print(f"recheck cpu {cpu}: {rate() / 1e3:.1f}k", flush=True)
