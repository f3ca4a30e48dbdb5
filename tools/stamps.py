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
        st = eng.read_symbol("pc_stamps", np.uint64, n_waves * 24).reshape(n_waves, 24).astype(np.int64)
        live = st[(st[:, 0] > 0) & (st[:, 7] > 0)]
        if rep >= 2 and len(live):
            rows.append(live)
    live = np.concatenate(rows, axis=0)
    # s_memtime counts per XCD (the counters of different XCDs are not aligned): durations are taken per wave; the wave
    # starts are placed with the constant 100 MHz counter (stamp 9, 10 ns a tick), which is global
    rel = live[:, :9] - live[:, :1]
    start_us = (live[:, 9] - live[:, 9].min()) * 0.01
    life = rel[:, 8]
    print(f"{len(live)} tile waves stamped over {len(rows)} launches; wave starts spread over {np.percentile(start_us % 1e6, 99):.2f} us "
          f"(launch-relative, p99); wave life: median {np.median(life):.0f} ticks, p10 {np.percentile(life, 10):.0f}, p90 {np.percentile(life, 90):.0f}")
    print(f"{'phase (between stamps)':58s} {'median':>8s} {'p10':>8s} {'p90':>8s} {'share':>6s}")
    names = ["0-1 wave start -> node values arrived", "1-2 node functions, first pass", "2-3 defect values, c~ stores",
             "3-4 path / integral rows (+ Hessian, fused build)", "4-5 Jacobian of the defect rows: staged + stored",
             "5-6 second partials (split build)", "6-7 Hessian runs from them, sums", "7-8 own stores drained"]
    prev = 0
    for i in range(1, 9):
        if np.all(rel[:, i] <= 0):      # a stamp this build does not have
            continue
        d = rel[:, i] - rel[:, prev]
        print(f"{names[i - 1]:58s} {np.median(d):8.0f} {np.percentile(d, 10):8.0f} {np.percentile(d, 90):8.0f} {np.median(d) / np.median(life):6.1%}")
        prev = i
    # per defect state (stamps 10 + 2a: block produced into LDS, 11 + 2a: read back and stores issued), over the waves
    # that own the state
    prevcol = 4
    print("Jacobian of the defect rows, state by state (waves that own the state):")
    for a in range(7):
        pa, fa = live[:, 10 + 2 * a] - live[:, 0], live[:, 11 + 2 * a] - live[:, 0]
        own = (live[:, 10 + 2 * a] > 0) & (live[:, 11 + 2 * a] > 0)
        if not own.any():
            continue
        print(f"  state {a}: {own.sum():6d} waves | produced at {np.median(pa[own]):8.0f} | produce->flushed {np.median((fa - pa)[own]):7.0f} ticks "
              f"(p10 {np.percentile((fa - pa)[own], 10):.0f}, p90 {np.percentile((fa - pa)[own], 90):.0f})")
    eng.close()


if __name__ == "__main__":
    main()
