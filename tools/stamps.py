"""Where a tile wave's time goes: a -DPC_STAMPS build of the problem's code object writes the shader clock at nine
points of pc::bulk per wave; this script runs a few evaluations and prints, per stamp, the median / 10 % / 90 % offset
from the launch's first wave start (us), i.e. a timeline of the whole launch.

    PYCOLLO_AMD_DEFINES=PC_STAMPS python tools/stamps.py --problem shuttle --sections 20000 --order 4
(build the object first with bench.py --build-only under the same PYCOLLO_AMD_DEFINES)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NAMES = ["0 wave starts", "1 node values arrived", "2 node functions (first pass) done", "3 defect values, c stores issued",
         "4 path / integral rows (+ fused Hessian) done", "5 defect Jacobian staged + stored", "6 second partials done",
         "7 instruction stream done", "8 own stores drained"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--problem", default="shuttle")
    ap.add_argument("--sections", type=int, default=20000)
    ap.add_argument("--order", type=int, default=4)
    ap.add_argument("--ragged", action="store_true")
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    assert "PC_STAMPS" in os.environ.get("PYCOLLO_AMD_DEFINES", ""), "run with PYCOLLO_AMD_DEFINES=PC_STAMPS"
    import torch
    from pycollo_amd import problems
    from pycollo_amd.engine import NlpEngine
    prob = problems.REGISTRY[args.problem](K=args.sections, order=args.order)
    if args.ragged:
        rr = np.random.default_rng(7)
        for ph in prob.phases:
            ph.mesh.mesh_section_sizes = rr.uniform(0.5, 1.5, args.sections)
            ph.mesh.number_mesh_section_nodes = rr.integers(4, 9, args.sections)
    eng = NlpEngine(prob, device=0)
    dev = torch.device("cuda", 0)
    x = torch.from_numpy(np.random.default_rng(1234).uniform(-0.45, 0.45, eng.num_x)).to(dev)
    lam = torch.from_numpy(np.random.default_rng(1235).normal(size=eng.num_c)).to(dev)
    c = torch.empty(eng.num_c, dtype=torch.float64, device=dev)
    G = torch.empty(eng.nnz_jac, dtype=torch.float64, device=dev)
    H = torch.empty(eng.nnz_hess, dtype=torch.float64, device=dev)
    W = eng.info["waves_per_tile"]
    n_waves = min(16384, (eng.info["n_tiles_total"] + 4) * W)
    print(f"{args.problem}: {eng.info['n_tiles_total']} tiles x {W} waves, lds {eng.info['lds_bytes_max']} B")
    rows = []
    for rep in range(args.reps + 2):
        for _ in range(3):   # warm
            eng.evaluate_all_device(x, 1.0, lam, c, G, H)
        eng.synchronize()
        eng.evaluate_all_device(x, 1.0, lam, c, G, H)
        eng.synchronize()
        st = eng.read_symbol("pc_stamps", np.uint64, n_waves * 10).reshape(n_waves, 10).astype(np.int64)
        st = st[st[:, 0] > 0]
        live = st[st[:, 7] > 0]
        if rep < 2 or len(live) == 0:
            continue
        t0 = live[:, 0].min()
        # shader clock rate from the constant 100 MHz counter (stamp 9) between the first and the last wave start
        i0, i1 = np.argmin(live[:, 0]), np.argmax(live[:, 0])
        dreal = live[i1, 9] - live[i0, 9]
        ghz = (live[i1, 0] - live[i0, 0]) / max(dreal, 1) * 0.1 if dreal > 0 else 2.1
        rows.append(((live[:, :9] - t0) / (ghz * 1e3), ghz, len(live)))
    ghz = np.median([r[1] for r in rows])
    print(f"shader clock {ghz:.2f} GHz, {rows[0][2]} tile waves stamped, {len(rows)} launches; offsets from the first wave start, us")
    print(f"{'stamp':46s} {'p10':>7s} {'median':>7s} {'p90':>7s} {'max':>7s}")
    allv = np.concatenate([r[0] for r in rows], axis=0)
    for i, nm in enumerate(NAMES):
        col = allv[:, i]
        col = col[col > -1e6]
        if np.all(allv[:, i] <= 0) and i not in (0,):
            continue
        print(f"{nm:46s} {np.percentile(col, 10):7.2f} {np.median(col):7.2f} {np.percentile(col, 90):7.2f} {col.max():7.2f}")
    d = np.diff(allv[:, :9], axis=1)
    print("durations between consecutive stamps, median us:", " ".join(f"{np.median(d[:, i]):.2f}" for i in range(8)))
    eng.close()


if __name__ == "__main__":
    main()
