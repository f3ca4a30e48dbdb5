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
nt = eng.info["n_tiles_total"]
out = np.zeros((nt, 16), dtype=np.int64)
eng._lib.pc_debug_stamps.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
assert eng._lib.pc_debug_stamps(eng._h, 0, out.ctypes.data, nt)
d = np.diff(out[:, :9], axis=1)
names = ["loads+staging+sync", "geometry+mu", "eval+LDS f+sync", "defect c", "path+integral", "hessian+partials", "defect G", "end"]
print(f"{name} K={K} n={order} TB={tpb}: tiles {nt}; cycles per phase (median over tiles; each stamp costs ~250-500 cycles itself)")
for i, nm in enumerate(names):
    print(f"  {nm:22s} {np.median(d[:, i]):8.0f}   (min {d[:, i].min():6d}, max {d[:, i].max():6d})")
print(f"  total                  {np.median(out[:, 8] - out[:, 0]):8.0f} cycles = {np.median(out[:, 8] - out[:, 0]) / 100:.2f} us at 100 MHz stamp clock")
if out[:, 9].max() > 0:
    print(f"  fused: hessian+partials+drain+arrival (stamp5->9) median {np.median(out[:, 9] - out[:, 5]):.0f} max {(out[:, 9] - out[:, 5]).max()}")
    lastb = int(np.argmax(out[:, 10]))
    print(f"  fused: last workgroup = tile {lastb}: start->arrival {out[lastb, 9] - out[lastb, 0]}, arrival->own work done {out[lastb, 8] - out[lastb, 9]}, tail {out[lastb, 10] - out[lastb, 8]} cycles")
    print(f"  fused: whole kernel (first start -> tail end) {out[lastb, 10] - out[:, 0].min()} cycles; first start -> last non-tail end {out[:, 8].max() - out[:, 0].min()}")
