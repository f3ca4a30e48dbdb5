"""Which threads does the process have while it launches, where do they run, and which of them burn CPU?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if os.environ.get("PIN_FIRST"):
    os.sched_setaffinity(0, {int(os.environ["PIN_FIRST"])})
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

def threads():
    out = {}
    for t in os.listdir("/proc/self/task"):
        try:
            st = open(f"/proc/self/task/{t}/stat").read()
            comm = st[st.index("(") + 1:st.rindex(")")]
            f = st[st.rindex(")") + 2:].split()
            allowed = [l.split(":")[1].strip() for l in open(f"/proc/self/task/{t}/status") if l.startswith("Cpus_allowed_list")][0]
            out[int(t)] = dict(comm=comm, state=f[0], utime=int(f[11]), stime=int(f[12]), cpu=int(f[36]), allowed=allowed)
        except (OSError, ValueError):
            pass
    return out

def run(n):
    for _ in range(n): step()
    torch.cuda.synchronize()

run(2000)
a = threads(); t0 = time.perf_counter(); run(60000); dt = time.perf_counter() - t0; b = threads()
print(f"main tid {os.getpid()} rate {60000 / dt / 1e3:.1f}k evals/s over {dt:.2f} s (ticks are 10 ms)")
for t, v in sorted(b.items()):
    du = v["utime"] - a.get(t, v)["utime"]; ds = v["stime"] - a.get(t, v)["stime"]
    print(f"  tid {t} {v['comm']:18s} state {v['state']} cpu {v['cpu']:3d} user +{du:3d} sys +{ds:3d} allowed {v['allowed']}")
