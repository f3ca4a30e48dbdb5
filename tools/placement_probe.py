"""Does the device time of an evaluation depend on where its buffers live?  One process, several engines (each with
its own internal buffers) and several sets of output buffers; evaluations queued behind a blocker."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pycollo_amd.hostpin import pin_launch_thread
pin_launch_thread()
import numpy as np, torch
from pycollo_amd import problems
from pycollo_amd.engine import NlpEngine
dev = torch.device("cuda", 0)
s = torch.cuda.Stream(device=dev); torch.cuda.set_stream(s); st = s.cuda_stream
prob = problems.hypersensitive(K=2000, order=6)
blk = torch.empty((8192, 8192), dtype=torch.float64, device=dev).normal_()
def measure(step, n=2000):
    for _ in range(200): step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.mm(blk, blk); e0.record()
    for _ in range(n): step()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
keep = []
for e in range(4):
    eng = NlpEngine(prob, device=0)
    x = torch.rand(eng.num_x, dtype=torch.float64, device=dev) - 0.5
    lam = torch.randn(eng.num_c, dtype=torch.float64, device=dev)
    for b in range(3):
        c = torch.empty(eng.num_c, dtype=torch.float64, device=dev); G = torch.empty(eng.nnz_jac, dtype=torch.float64, device=dev); H = torch.empty(eng.nnz_hess, dtype=torch.float64, device=dev)
        keep += [c, G, H, torch.empty(int(np.random.default_rng(e * 3 + b).integers(1, 64)) * 4096, dtype=torch.uint8, device=dev)]
        us = measure(eng.bind_device(x, lam, c, G, H, st))
        ub = measure(lambda: eng.launch_bulk_only(x, lam, c, G, H, st))
        print(f"engine {e} buffers {b}: device us/eval {us:6.3f}  bulk only {ub:6.3f}  c@{c.data_ptr() & 0xffffff:06x} G@{G.data_ptr() & 0xffffff:06x}", flush=True)
    keep.append(eng)
