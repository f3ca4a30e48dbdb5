"""Development aid: what one rank of the KKT factorisation cut across ranks (pycollo_amd/kkt_sharded.py) costs, against the
single-rank factorisation -- wall time of the C calls with their vector transfers, ranks run one after the other on the
one GPU (so a rank's figure is what it would take on a GPU of its own; the reductions between them are nb_red^2 and
nb_red doubles and are not timed).  -> profiles/r04_kkt_sharded_time.txt"""
import os, sys, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'tests'))
import numpy as np
from test_kkt_cpu import kkt_case
from pycollo_amd import kkt_sharded
from pycollo_amd.kkt import GpuKkt
from pycollo_amd.sharding import ShardPlan


def med(fn, n=15):
    fn(); ts = []
    for _ in range(n):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    return 1e3 * float(np.median(ts))


cases = [("hypersensitive", dict(K=2000, order=6)), ("cart_pole", dict(K=5000, order=4)), ("shuttle", dict(K=6000, order=4)),
         ("shuttle", dict(K=100000, order=4)), ("hypersensitive", dict(K=200000, order=6))]
only = os.environ.get("KKT_PROBLEM")
only_k = os.environ.get("KKT_K")
worlds = [int(w) for w in os.environ.get("KKT_WORLDS", "2,4,8").split(",")]
ends = os.environ.get("KKT_ENDS", "chain")
for name, kw in cases:
    if (only and name != only) or (only_k and int(only_k) != kw["K"]) or (not only_k and kw["K"] >= 100000):
        continue
    eng, ora, x, lam, ineq, fixed, sc, dvec = kkt_case(name, kw, device=0)
    eng.evaluate_resident(x, 1.0, lam)
    k0 = GpuKkt(eng, ineq, fixed, sc)
    rhs = np.random.default_rng(0).normal(size=k0.nu)
    f0, s0 = med(lambda: k0.factor(dvec)), med(lambda: k0.solve(rhs))
    x0 = k0.solve(rhs)
    inertia0 = k0.factor(dvec)
    mb0, nb0 = 8e-6 * k0.tables.total_vals, k0.tables.nb
    print(f"{name} {kw} [shared nodes: {ends}]: nu {k0.nu}  single rank: factor {f0:.3f} ms  solve {s0:.3f} ms  matrix {mb0:.1f} MB  border {nb0}", flush=True)
    k0.close()
    for world in worlds:
        t0 = time.perf_counter()
        plan = kkt_sharded.ShardedKktPlan(eng, ineq, fixed, sc, ShardPlan(eng, world), ends=ends)
        t_plan = time.perf_counter() - t0
        big = kw["K"] >= 100000
        fr, fw, bw = [], [], []
        B = np.zeros((plan.nb_red, plan.nb_red))
        cnt = np.zeros(2, np.int64)
        rb = np.zeros(plan.nb_red)
        sks = {}
        for r in range(world):                       # one rank's handle at a time (a rank's matrix is what a GPU would hold)
            sk = kkt_sharded.ShardedKkt(eng, plan, [r])
            h = sk.handles[r]
            R = plan.ranks[r]
            exp = bool(R.export_red)
            Bl, p, q = h.factor_partial(plan.local_vector(r, dvec))
            plan.add_border(r, Bl, B, h.export_panels() if exp else None); cnt += (p, q)
            rb[R.border_red] += h.forward_partial(plan.local_vector(r, rhs))
            if exp:
                np.add.at(rb, np.concatenate(R.export_red), h.export_rhs())
            if not big or r in (0, 1, world - 1):
                nexp = sum(len(a) for a in R.export_red) if exp else 0
                fr.append(med(lambda: (h.factor_partial(plan.local_vector(r, dvec)), exp and h.export_panels()), 7))
                fw.append(med(lambda: (h.forward_partial(plan.local_vector(r, rhs)), exp and h.export_rhs()), 7))
                bw.append(med(lambda: (exp and h.import_solution(np.zeros(nexp)), h.backward_partial(np.zeros(R.tables.nb))), 7))
            if big:
                sk.close()
            else:
                sks[r] = sk
        red = GpuKkt(eng, None, None, None, tables=plan.reduced)
        p, q = red.border_load_factor(B)
        assert (int(cnt[0] + p), int(cnt[1] + q)) == inertia0, ((cnt[0] + p, cnt[1] + q), inertia0)
        fred = med(lambda: red.border_load_factor(B))
        sred = med(lambda: red.solve(rb))
        diff = float("nan")
        if not big:
            xb = red.solve(rb)
            x1 = np.zeros(plan.nu)
            for r, sk in sks.items():
                R = plan.ranks[r]
                sk.handles[r].forward_partial(plan.local_vector(r, rhs))      # (the timing loops above overwrote its state)
                if R.export_red:
                    sk.handles[r].import_solution(xb[np.concatenate(R.export_red)])
                xl = sk.handles[r].backward_partial(xb[R.border_red])
                x1[R.univ[R.own]] = xl[R.own]
                sk.close()
            diff = np.max(np.abs(x1 - x0)) / np.max(np.abs(x0))
        red.close()
        print(f"  {world} ranks: slowest rank factor {max(fr):.3f} ms + reduced ({plan.nb_red} unknowns) {fred:.3f} ms = {max(fr) + fred:.3f} ms ({f0 / (max(fr) + fred):.2f}x)"
              f" | solve: forward {max(fw):.3f} + reduced {sred:.3f} + backward {max(bw):.3f} = {max(fw) + sred + max(bw):.3f} ms ({s0 / (max(fw) + sred + max(bw)):.2f}x)"
              f" | rank matrix {8e-6 * max(plan.footprint(r)['local_vals'] for r in range(world)):.1f} MB, local border {max(plan.footprint(r)['nb_local'] for r in range(world))}"
              f" | unrefined step difference {diff:.1e} | plan build {t_plan:.2f} s", flush=True)
    eng.close()
