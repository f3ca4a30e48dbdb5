"""Does the device time of an evaluation depend on the HIP stream (hardware queue) it is launched on?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pycollo_amd.hostpin import pin_launch_thread
pin_launch_thread()
import numpy as np, torch
from pycollo_amd import problems
from pycollo_amd.engine import NlpEngine
dev = torch.device("cuda", 0)
prob = problems.hypersensitive(K=2000, order=6)
eng = NlpEngine(prob, device=0)
x = torch.rand(eng.num_x, dtype=torch.float64, device=dev) - 0.5
lam = torch.randn(eng.num_c, dtype=torch.float64, device=dev)
c = torch.empty(eng.num_c, dtype=torch.float64, device=dev); G = torch.empty(eng.nnz_jac, dtype=torch.float64, device=dev); H = torch.empty(eng.nnz_hess, dtype=torch.float64, device=dev)
blk = torch.empty((8192, 8192), dtype=torch.float64, device=dev).normal_()
def measure(s, n=2000):
    step = eng.bind_device(x, lam, c, G, H, s.cuda_stream)
    with torch.cuda.stream(s):
        for _ in range(200): step()
        s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.mm(blk, blk); e0.record()
        for _ in range(n): step()
        e1.record(); s.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
streams = [torch.cuda.Stream(device=dev) for _ in range(8)] + [torch.cuda.Stream(device=dev, priority=-1)]
for i, s in enumerate(streams):
    print(f"stream {i} (handle {s.cuda_stream:#x}, priority {s.priority}): device us/eval {measure(s):6.3f}", flush=True)
for i, s in enumerate(streams[:3]):
    print(f"again stream {i}: {measure(s):6.3f}", flush=True)
