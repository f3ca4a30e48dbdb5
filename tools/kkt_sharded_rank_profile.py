"""Development aid: one rank's part of the sharded KKT factorisation in a loop, for rocprofv3 --kernel-trace --stats.
KKT_PROBLEM / KKT_K / KKT_ORDER / KKT_WORLD / KKT_RANK select the case."""
import os, sys
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'tests'))
import numpy as np
from test_kkt_cpu import kkt_case
from pycollo_amd import kkt_sharded
from pycollo_amd.sharding import ShardPlan
name = os.environ.get("KKT_PROBLEM", "shuttle")
kw = dict(K=int(os.environ.get("KKT_K", "6000")), order=int(os.environ.get("KKT_ORDER", "4")))
world, rank = int(os.environ.get("KKT_WORLD", "8")), int(os.environ.get("KKT_RANK", "1"))
eng, ora, x, lam, ineq, fixed, sc, dvec = kkt_case(name, kw, device=0)
eng.evaluate_resident(x, 1.0, lam)
plan = kkt_sharded.ShardedKktPlan(eng, ineq, fixed, sc, ShardPlan(eng, world))
sk = kkt_sharded.ShardedKkt(eng, plan, [rank])
h = sk.handles[rank]
T = plan.ranks[rank].tables
print(name, kw, "rank", rank, "of", world, "leaves", T.n_leaf, "chain", T.n_chain, "segments", T.n_phase, "nb", T.nb, flush=True)
rhs = np.random.default_rng(0).normal(size=plan.nu)
for _ in range(10):
    h.factor_partial(plan.local_vector(rank, dvec))
    h.forward_partial(plan.local_vector(rank, rhs))
    h.backward_partial(np.zeros(T.nb))
