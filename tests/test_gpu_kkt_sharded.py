"""The KKT factorisation cut across ranks on the GPU (pycollo_amd/kkt_sharded.py; pc_kkt_factor_partial /
pc_kkt_forward_partial / pc_kkt_backward_partial / pc_kkt_border_load_factor): every rank eliminates its own leaves and
chain segments from G~ / H~ values that are NaN wherever another rank's tile kernels write, the reduced border system
is summed and factorised -- against the single-rank factorisation (inertia equal, steps to 1e-9) and SuperLU.  Ranks are
run one after the other on the one GPU here, and as two processes over gloo below."""
import os
import subprocess
import sys

import numpy as np
import pytest
import scipy.sparse.linalg as spla

from conftest import ROOT
from test_gpu_sharded_process import _free_port
from test_kkt_cpu import kkt_case, reference_matrix
from test_kkt_sharded_cpu import rank_values

pytestmark = pytest.mark.gpu

CASES = [("hypersensitive", dict(K=2000, order=6), 8), ("hypersensitive", dict(K=2000, order=6), 2),
         ("cart_pole", dict(K=500, order=4), 4), ("shuttle", dict(K=60, order=4), 3), ("sliding_mass", dict(num_phases=3, K=40, order=5), 4),
         ("time_coupled_transfer", dict(K=120, order=4), 2), ("free_flying_robot", dict(K=160, order=5), 5),
         ("sliding_mass", dict(num_phases=3, K=8, order=4), 2)]       # (the last: one tile per phase, i.e. one rank has it all)


@pytest.mark.parametrize("ends", ["chain", "border"])
@pytest.mark.parametrize("name,kw,world", CASES)
def test_ranks_one_after_the_other_match_the_single_rank_factorisation(built, name, kw, world, ends):
    """``ends``: the shared nodes as the un-eliminated ends of a rank's chain segments (default) or in its border."""
    import torch
    from pycollo_amd import kkt_sharded
    from pycollo_amd.kkt import GpuKkt
    from pycollo_amd.sharding import ShardPlan
    eng, ora, x, lam, ineq, fixed, sc, dvec = kkt_case(name, kw, device=0)
    c, G, H = eng.evaluate_all(x, 1.0, lam)
    eng.evaluate_resident(x, 1.0, lam)
    sp = ShardPlan(eng, world)
    plan = kkt_sharded.ShardedKktPlan(eng, ineq, fixed, sc, sp, ends=ends, whole_entries=True)
    dev = torch.device("cuda", 0)
    vals = [[torch.from_numpy(a).to(dev) for a in rank_values(plan, sp, r, G, H)] for r in range(world)]
    busy = [r for r in range(world) if any(te > tb for tb, te in sp.tile_ranges[r])]
    if len(busy) > 1:
        assert all(bool(torch.isnan(vals[r][0]).any()) for r in busy)   # (another rank's rows really are unreadable)
    sk = kkt_sharded.ShardedKkt(eng, plan, range(world), d_jac=[g.data_ptr() for g, _ in vals], d_hess=[h.data_ptr() for _, h in vals])
    k0 = GpuKkt(eng, ineq, fixed, sc)
    K = reference_matrix(eng, G, H, ineq, fixed, sc, dvec)
    inertia = sk.factor(dvec)
    assert inertia == k0.factor(dvec)
    rng = np.random.default_rng(1)
    rhs = rng.normal(size=plan.nu)
    rhs[np.nonzero(fixed)[0]] = 0.0
    xs = sk.solve(rhs)
    assert np.all(np.isfinite(xs))
    xs = xs + sk.solve(rhs - K @ xs)
    x0 = k0.solve(rhs)
    x0 = x0 + k0.solve(rhs - K @ x0)
    assert np.max(np.abs(xs - x0)) <= 1e-9 * np.max(np.abs(x0))
    lu = spla.splu(K)
    xr = lu.solve(rhs)
    xr = xr + lu.solve(rhs - K @ xr)
    assert np.max(np.abs(xs - xr)) <= 1e-9 * np.max(np.abs(xr))
    # the cut plan run by one rank (wide border) is the same elimination: same inertia, same step
    kw_ = GpuKkt(eng, None, None, None, tables=plan.whole)
    assert kw_.factor(dvec) == inertia
    xw = kw_.solve(rhs)
    xw = xw + kw_.solve(rhs - K @ xw)
    assert np.max(np.abs(xw - xs)) <= 1e-9 * np.max(np.abs(xs))
    # fixed order, no atomics
    sk.factor(dvec)
    np.testing.assert_array_equal(sk.solve(rhs), sk.solve(rhs))
    # a rank's matrix storage: its share of the blocks, each widened by the rank's two cut nodes per phase, which ride in
    # its local border (1.2x the share for the 3-unknown nodes of the hypersensitive problem, 2.7x for the shuttle's 14)
    #  -- as chain ends they widen nothing: the rank's border is the NLP's own
    bound = 3.0 if ends == "border" else 1.3
    assert max(plan.footprint(r)["local_vals"] for r in range(world)) <= bound * k0.tables.total_vals / len(busy) + 65536
    if ends == "chain":
        assert all(plan.ranks[r].tables.nb == k0.tables.nb for r in range(world))
    for h in (sk, k0, kw_):
        h.close()
    eng.close()


_TWO_RANKS = r'''
import sys, os, traceback
def _excepthook(t, v, tb):
    with open(os.environ["KKT_LOG"] + f".{os.environ.get('RANK', '0')}", "w") as f:
        traceback.print_exception(t, v, tb, file=f)
    traceback.print_exception(t, v, tb)
sys.excepthook = _excepthook
import numpy as np
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("gloo")
rank = dist.get_rank()
sys.path.insert(0, "tests")
from pycollo_amd import problems, kkt_sharded
from pycollo_amd.engine import NlpEngine
from pycollo_amd.kkt import GpuKkt
from pycollo_amd.sharding import ShardedNlp
from test_kkt_cpu import reference_matrix
dev = torch.device("cuda", 0)
for name, kw in (("hypersensitive", dict(K=400, order=6)), ("shuttle", dict(K=120, order=4)), ("sliding_mass", dict(num_phases=3, K=120, order=5))):
    prob = problems.REGISTRY[name](**kw)
    sh = ShardedNlp(prob, device=0)
    eng = sh.engine
    n, m = eng.num_x, eng.num_c
    rng = np.random.default_rng(3)
    xh, lamh = rng.uniform(-0.3, 0.3, n), 0.1 * rng.normal(size=m)
    x, lam = torch.from_numpy(xh).to(dev), torch.from_numpy(lamh).to(dev)
    lay = eng.layout
    ineq = []
    for pl, pm in zip(lay.phases, eng.model.phases):
        ineq += list(range(pl.c_path_off, pl.c_path_off + pm.n_p * pl.N))
    ineq = np.array(sorted(ineq + list(range(lay.c_end_off, m, 2))), dtype=np.int64)
    ns = len(ineq)
    fixed = np.zeros(n + ns, bool)
    fixed[rng.choice(n, size=max(1, n // 40), replace=False)] = True
    sc = rng.uniform(0.5, 1.0, m)
    dvec = np.concatenate([rng.uniform(0.5, 2.0, n + ns) + 50.0, -1e-8 * np.ones(m)])
    # this rank's tiles, the partial sums of the others, the tail -- into a buffer that is NaN everywhere else
    sh.buf.fill_(float("nan"))
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        c, G, H = sh.evaluate_local_device(x, 1.0, lam, st)
    torch.cuda.synchronize()
    assert bool(torch.isnan(G).any()) and bool(torch.isnan(H).any())
    plan = kkt_sharded.ShardedKktPlan(eng, ineq, fixed, sc, sh.plan)
    sk = kkt_sharded.ShardedKkt(eng, plan, [rank], d_jac=G.data_ptr(), d_hess=H.data_ptr(), distributed=True)
    inertia = sk.factor(dvec)
    # the single-rank factorisation of the whole evaluation, on an unsharded engine of this process
    ref = NlpEngine(prob, device=0)
    _, Gf, Hf = ref.evaluate_all(xh, 1.0, lamh)
    ref.evaluate_resident(xh, 1.0, lamh)
    k0 = GpuKkt(ref, ineq, fixed, sc)
    K = reference_matrix(ref, Gf, Hf, ineq, fixed, sc, dvec)
    ok = inertia == k0.factor(dvec)
    rhs = np.random.default_rng(1).normal(size=plan.nu)
    rhs[np.nonzero(fixed)[0]] = 0.0
    xs = sk.solve(rhs)
    xs = xs + sk.solve(rhs - K @ xs)
    x0 = k0.solve(rhs)
    x0 = x0 + k0.solve(rhs - K @ x0)
    err = float(np.max(np.abs(xs - x0)) / np.max(np.abs(x0)))
    ok = ok and bool(np.all(np.isfinite(xs))) and err <= 1e-9
    f = plan.footprint(rank)
    print(f"SHARDED KKT {name} rank {rank}: inertia {inertia} step error {err:.2e} local border {f['nb_local']} reduced {f['nb_reduced']} ok: {ok}", flush=True)
    sk.close(); k0.close(); ref.close()
    if not ok:
        sys.exit(1)
dist.barrier()
dist.destroy_process_group()
print(f"SHARDED KKT rank {rank} done")
'''


def test_two_processes_factorise_one_kkt_system(built, tmp_path):
    """Two ranks under torch.distributed.run sharing the one GPU (gloo): each evaluates its own tiles (only the per-tile
    partial sums are exchanged), factorises its leaves and chain segments from them, and the two reductions of the border
    system are all that crosses between the processes."""
    script = tmp_path / "kkt_two_ranks.py"
    script.write_text(_TWO_RANKS)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""),
               KKT_LOG=str(tmp_path / "trace"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)]
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    out = res.stdout + res.stderr
    traces = "".join(p.read_text() for p in tmp_path.glob("trace.*"))
    assert res.returncode == 0, traces + out[-1500:]
    assert out.count("ok: True") == 6, out[-3000:]
    from test_gpu_ipm_sharded import _keep
    _keep("\n".join(ln for ln in out.splitlines() if "SHARDED KKT" in ln), "sharded_kkt_two_ranks.txt")
    assert "SHARDED KKT rank 0 done" in out and "SHARDED KKT rank 1 done" in out
