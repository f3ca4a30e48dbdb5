"""Diagnostic: in-kernel phase timeline of the bulk kernel (PYCOLLO_AMD_DBG_STAGE=9)."""
import os, sys, ctypes as C
os.environ["PYCOLLO_AMD_DBG_STAGE"] = "9"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pycollo_amd import problems
from pycollo_amd.engine import NlpEngine
name = sys.argv[1] if len(sys.argv) > 1 else "hypersensitive"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
order = int(sys.argv[3]) if len(sys.argv) > 3 else 6
tpb = int(sys.argv[4]) if len(sys.argv) > 4 else 64
prob = problems.REGISTRY[name](K=K, order=order)
if os.environ.get("RAGGED"):
    rr = np.random.default_rng(7)
    for ph in prob.phases:
        ph.mesh.mesh_section_sizes = rr.uniform(0.5, 1.5, K)
        ph.mesh.number_mesh_section_nodes = rr.integers(4, 9, K)
eng = NlpEngine(prob, device=0, threads_per_block=tpb, specialise=not os.environ.get("GENERIC"))
dev = torch.device("cuda", 0)
x = torch.rand(eng.num_x, dtype=torch.float64, device=dev) - 0.5
lam = torch.randn(eng.num_c, dtype=torch.float64, device=dev)
c = torch.empty(eng.num_c, dtype=torch.float64, device=dev); G = torch.empty(eng.nnz_jac, dtype=torch.float64, device=dev); H = torch.empty(eng.nnz_hess, dtype=torch.float64, device=dev)
s = torch.cuda.Stream(device=dev)
for _ in range(200):
    eng.evaluate_all_device(x, 1.0, lam, c, G, H, s.cuda_stream)
torch.cuda.synchronize()
nt = eng.info["n_tiles_total"] if len(prob.phases) == 1 else None
eng._lib.pc_debug_stamps.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
names = ["loads+staging+sync", "geometry+mu", "eval+LDS f+sync", "defect c", "path+integral", "hessian+partials", "defect G (+2nd pass)", "meet+end"]
for ip in range(len(prob.phases)):
    ntp = nt if nt is not None else 4096
    raw = np.zeros((ntp, 4, 16), dtype=np.int64)
    try:
        assert eng._lib.pc_debug_stamps(eng._h, ip, raw.ctypes.data, ntp)
    except AssertionError:
        print("no stamps for phase", ip); continue
    raw = raw[raw[:, 0, 0] > 0]
    print(f"{name} K={K} n={order} TB={tpb} W={eng.info['waves_per_tile']} phase {ip}: {len(raw)} tiles; s_memtime ticks (shader clock, ~2.1-2.4 GHz) per stage, median over tiles, per wave of the tile")
    for w in range(4):
        o = raw[:, w, :]
        o = o[o[:, 0] > 0]
        if not len(o):
            continue
        seq = np.stack([o[:, 0], o[:, 1], o[:, 2], o[:, 3], o[:, 4], o[:, 5], o[:, 6], o[:, 11], o[:, 7]], axis=1)
        d = np.diff(seq, axis=1)
        print(f"  wave {w}: " + " | ".join(f"{nm} {np.median(d[:, i]):.0f}" for i, nm in enumerate(names)) + f" | total {np.median(seq[:, -1] - seq[:, 0]):.0f} ticks")
    t0 = raw[:, :, 0][raw[:, :, 0] > 0].min()
    en = raw[:, :, 7].max() - t0
    print(f"  first start -> last end: {en} ticks")
