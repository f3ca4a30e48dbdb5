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


def xcd_major(b, nb):
    q, r, x, j = nb // 8, nb % 8, b % 8, b // 8
    return x * q + np.minimum(x, r) + j


def by_order(eng, prob, live, slot, W, n_launches):
    """Mixed build: the life of the tile waves by the order of the body they ran, and which tiles end last.  Slot ->
    workgroup -> position in the XCD-major order -> the tile of that position (phase after phase, pc_engine.hip)."""
    orders = np.concatenate([eng.phase_tile_orders(p) for p in range(len(eng.mixed))])
    rows = []
    for p, ph in enumerate(prob.phases):
        k0 = eng.phase_tiles(p)[0]
        cum = np.concatenate([[0], np.cumsum(np.asarray(ph.mesh.number_mesh_section_nodes) - 1)])
        rows.append(np.diff(cum[k0]))
    rows = np.concatenate(rows)
    nb = orders.size
    blk = slot // W
    ntb = int(blk.max()) + 1 - nb          # leading tail workgroups of a resident launch (they carry no tile stamps)
    pos = xcd_major(blk - ntb, nb)
    keep = (blk >= ntb) & (pos < nb)
    live, pos, blk = live[keep], pos[keep], blk[keep]
    od, rw = orders[pos], rows[pos]
    life = (live[:, 8] - live[:, 0]).astype(float)
    start = live[:, 9] * 0.01               # us after the launch's first wave (100 MHz clock: 10 ns steps)
    print(f"tile waves by the order of their body ({nb} tiles, {ntb} tail workgroups, {n_launches} launches):")
    print(f"{'order':>6s} {'tiles':>6s} {'rows/tile':>9s} {'life median':>12s} {'p90':>8s} {'max':>8s} {'ticks/row':>9s} {'start p50 us':>12s} {'start p99 us':>12s}")
    for o in np.unique(od):
        m = od == o
        print(f"{int(o) if o else 'any':>6} {int(np.sum(orders == o)):6d} {np.mean(rw[m]):9.1f} {np.median(life[m]):12.0f} {np.percentile(life[m], 90):8.0f} "
              f"{life[m].max():8.0f} {np.median(life[m] / rw[m]):9.1f} {np.median(start[m]):12.2f} {np.percentile(start[m], 99):12.2f}")
    top = np.argsort(-life)[: max(10, len(life) // 50)]
    print("the longest 2 % of waves by order:", {int(o): int(np.sum(od[top] == o)) for o in np.unique(od[top])},
          f"| their median life {np.median(life[top]):.0f}")
    print("start (us, launch-relative) of waves: p50 %.2f p90 %.2f p99 %.2f max %.2f" % tuple(np.percentile(start, [50, 90, 99, 100])))
    late = start > 2.0
    print(f"waves that start more than 2 us after the first: {late.mean():.1%}" + (f", their start p50 {np.median(start[late]):.1f} us, "
          f"life median {np.median(life[late]):.0f}" if late.any() else ""))
    for rate in (2100.0, 2400.0):           # shader clock candidates (ticks per us)
        end = start + life / rate
        print(f"  at {rate:.0f} ticks/us: last wave ends {end.max():.1f} us after the first starts; p99 {np.percentile(end, 99):.1f}, p90 {np.percentile(end, 90):.1f}, p50 {np.median(end):.1f}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--problem", default="shuttle")
    ap.add_argument("--sections", type=int, default=20000)
    ap.add_argument("--order", type=int, default=4)
    ap.add_argument("--ragged", action="store_true")
    ap.add_argument("--refined", type=int, default=0, metavar="NODES",
                    help="ph-refined mesh of about NODES nodes per phase (mixed build): adds a per-tile-order table")
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
    if args.refined:
        prob = problems.with_refined_mesh(problems.REGISTRY[args.problem](), args.refined)
    eng = NlpEngine(prob, device=0)
    dev = torch.device("cuda", 0)
    lo, hi = (0.05, 0.3) if args.problem == "delta_iii" else (-0.45, 0.45)
    x = torch.from_numpy(np.random.default_rng(1234).uniform(lo, hi, eng.num_x)).to(dev)
    lam = torch.from_numpy(np.random.default_rng(1235).normal(size=eng.num_c)).to(dev)
    c = torch.empty(eng.num_c, dtype=torch.float64, device=dev)
    G = torch.empty(eng.nnz_jac, dtype=torch.float64, device=dev)
    H = torch.empty(eng.nnz_hess, dtype=torch.float64, device=dev)
    W = eng.info["waves_per_tile"]
    n_waves = min(16384, (eng.info["n_tiles_total"] + 4) * W)
    print(f"{args.problem}: {eng.info['n_tiles_total']} tiles x {W} waves, lds {eng.info['lds_bytes_max']} B")
    rows = []
    slots = []
    for rep in range(args.reps + 2):
        for _ in range(3):   # warm
            eng.evaluate_all_device(x, 1.0, lam, c, G, H)
        eng.synchronize()
        eng.evaluate_all_device(x, 1.0, lam, c, G, H)
        eng.synchronize()
        st = None
        for part in range(8):   # a code object in parts: the stamps are in the part whose kernel ran
            try:
                cand = eng.read_symbol(f"pc_stamps@{part}", np.uint64, n_waves * 24).reshape(n_waves, 24).astype(np.int64)
            except RuntimeError:
                break
            if st is None or np.count_nonzero(cand[:, 0]) > np.count_nonzero(st[:, 0]):
                st = cand
        ok = (st[:, 0] > 0) & (st[:, 7] > 0)
        live = st[ok]
        if rep >= 2 and len(live):
            live = live.copy()
            live[:, 9] -= live[:, 9].min()      # constant-rate clock relative to the launch's first wave
            rows.append(live)
            slots.append(np.flatnonzero(ok))
    live = np.concatenate(rows, axis=0)
    # s_memtime counts per XCD (the counters of different XCDs are not aligned): durations are taken per wave; the wave
    # starts are placed with the constant 100 MHz counter (stamp 9, 10 ns a tick), which is global
    rel = live[:, :9] - live[:, :1]
    start_us = live[:, 9] * 0.01
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
    if args.refined and any(eng.mixed):
        by_order(eng, prob, live, np.concatenate(slots), W, len(rows))
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
