"""Section-range sharding: the plan partitions exactly the bulk-kernel outputs, and the exchange
reassembles them over a real 2-process gloo group on the CPU (the N > 1 path of bench.py)."""
import os
import socket

import numpy as np
import pytest

from pycollo_amd import problems

CASES = [("hypersensitive", dict(K=200, order=6)), ("two_phase_transfer", dict(K=40, order=4)),
         ("time_coupled_transfer", dict(K=40, order=4)),
         ("delta_iii", dict(K=30, order=4)), ("double_pendulum", dict(K=50, order=4))]


def _expected_bulk_positions(eng):
    """Positions of the combined buffer the bulk kernels write, derived from the structure alone."""
    lay, model = eng.layout, eng.model
    gr, gc = eng.evaluate_G_structure()
    hr, hc = eng.evaluate_H_structure()
    c_pos, g_mask, h_mask = [], np.zeros(len(gr), bool), np.zeros(len(hr), bool)
    for pm, pl in zip(model.phases, lay.phases):
        c_pos.append(np.arange(pl.c_off, pl.c_int_off))                      # defect + path rows
        g_mask |= (gr >= pl.c_off) & (gr < pl.c_int_off)
        in_int = (gr >= pl.c_int_off) & (gr < pl.c_int_off + pm.n_q)
        g_mask |= in_int & (gc >= pl.x_off) & (gc < pl.q_off)                # z columns of the integral rows
        N = pl.N
        zrow = (hr >= pl.x_off) & (hr < pl.q_off)
        zcol = (hc >= pl.x_off) & (hc < pl.q_off)
        same_node = ((hr - pl.x_off) % N) == ((hc - pl.x_off) % N)
        h_mask |= zrow & zcol & same_node                                    # node bands
        trow = (hr >= pl.q_off) & (hr < pl.t_off + pm.n_t)                  # q and t rows of the phase
        srow = hr >= lay.s_off
        h_mask |= (trow | srow) & zcol                                       # strips (may include endpoint overlaps)
    return np.concatenate(c_pos), np.nonzero(g_mask)[0] + eng.num_c, np.nonzero(h_mask)[0] + eng.num_c + eng.nnz_jac


@pytest.mark.parametrize("name,kw", CASES)
@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_plan_partitions_bulk_outputs(built, name, kw, world):
    from pycollo_amd.engine import NlpEngine
    from pycollo_amd.sharding import ShardPlan
    eng = NlpEngine(problems.REGISTRY[name](**kw), device=None)
    plan = ShardPlan(eng, world)
    allidx = np.concatenate(plan.index)
    assert len(np.unique(allidx)) == len(allidx), "two ranks claim the same output position"
    c_pos, g_pos, h_pos = _expected_bulk_positions(eng)
    got = set(allidx.tolist())
    oP = eng.num_c + eng.nnz_jac + eng.nnz_hess
    # every bulk-written c / G position is shipped by exactly one rank, and nothing else of c / G is
    assert {i for i in got if i < eng.num_c} == set(c_pos.tolist())
    assert {i for i in got if eng.num_c <= i < eng.num_c + eng.nnz_jac} == set(g_pos.tolist())
    # H: whole-row runs may carry endpoint extras of the edge rows (the tail rewrites those), never fewer
    h_got = {i for i in got if eng.num_c + eng.nnz_jac <= i < oP}
    assert set(h_pos.tolist()) <= h_got
    # the split plan (c~ / G~ runs | H~ runs + partial sums) is the same partition in two parts
    cg, hp = plan.split()
    for r in range(world):
        np.testing.assert_array_equal(np.sort(np.concatenate([cg.index[r], hp.index[r]])), np.sort(plan.index[r]))
        assert np.all(cg.index[r] < eng.num_c + eng.nnz_jac) and np.all(hp.index[r] >= eng.num_c + eng.nnz_jac)
    # all partial sums travel
    assert {i for i in got if i >= oP} == set(range(oP, plan.total))
    # tile ranges tile every phase exactly
    for ip, (k0, _) in enumerate(plan.tiles):
        rs = [plan.tile_ranges[r][ip] for r in range(world)]
        assert rs[0][0] == 0 and rs[-1][1] == len(k0) - 1 and all(a[1] == b[0] for a, b in zip(rs, rs[1:]))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _ragged(prob, K, seed=7):
    rr = np.random.default_rng(seed)
    for ph in prob.phases:
        ph.mesh.number_mesh_sections = K
        ph.mesh.mesh_section_sizes = rr.uniform(0.5, 1.5, K)
        ph.mesh.number_mesh_section_nodes = rr.integers(3, 9, K)
    return prob


def _worker(rank, world, port, name, kw, q, root=None, mode="single"):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pycollo_amd.engine import NlpEngine
        from pycollo_amd.sharding import SegmentExchange, ShardPlan
        kw = dict(kw)
        ragged = kw.pop("ragged", False)
        prob = problems.REGISTRY[name](**kw)
        if ragged:
            prob = _ragged(prob, kw["K"])
        eng = NlpEngine(prob, device=None)
        plan = ShardPlan(eng, world)
        ref = np.random.default_rng(0).normal(size=plan.total)       # stands for [c | G | H | partials]
        buf = torch.full((plan.total,), float("nan"), dtype=torch.float64)
        mine = torch.from_numpy(plan.index[rank])
        buf[mine] = torch.from_numpy(ref)[mine]                      # what this rank's bulk kernels produce
        if mode == "split":      # the overlapped form's two exchanges: c~ / G~ runs, then H~ runs + partial sums
            for sub in plan.split():
                SegmentExchange(sub, rank, torch.device("cpu")).run(buf, root)
        else:
            SegmentExchange(plan, rank, torch.device("cpu")).run(buf, root, unpadded=(mode == "unpadded"))
        allidx = np.concatenate(plan.index)
        if root is None or rank == root:
            ok = bool(np.array_equal(buf.numpy()[allidx], ref[allidx]))
            untouched = np.setdiff1d(np.arange(plan.total), allidx)
            ok = ok and bool(np.all(np.isnan(buf.numpy()[untouched])))
        else:   # a sender keeps its own share and nothing else
            ok = bool(np.array_equal(buf.numpy()[plan.index[rank]], ref[plan.index[rank]]))
            others = np.setdiff1d(np.arange(plan.total), plan.index[rank])
            ok = ok and bool(np.all(np.isnan(buf.numpy()[others])))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,kw,root,mode", [("two_phase_transfer", dict(K=40, order=4), None, "single"),
                                               ("hypersensitive", dict(K=300, order=6), None, "single"),
                                               ("delta_iii", dict(K=37, order=4, ragged=True), None, "single"),      # ph-refined style mesh
                                               ("shuttle", dict(K=90, order=4, ragged=True), 0, "single"),           # gather to rank 0
                                               ("delta_iii", dict(K=37, order=4, ragged=True), None, "split"),       # overlapped form
                                               ("two_phase_transfer", dict(K=40, order=4), 1, "split"),
                                               ("shuttle", dict(K=90, order=4, ragged=True), None, "unpadded")])     # all-gatherv
def test_exchange_world2_gloo(built, name, kw, root, mode):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, name, kw, q, root, mode)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    results = dict(q.get(timeout=10) for _ in range(2))
    assert results == {0: True, 1: True}


def test_shares_are_balanced_by_nodes_on_a_refined_mesh(built):
    """Contiguous tile ranges cut by node count: on a mesh with 3..8 nodes per section the longest share stays within
    a few per cent of the mean (the all-gather is padded to the longest)."""
    from pycollo_amd.engine import NlpEngine
    from pycollo_amd.sharding import ShardPlan
    eng = NlpEngine(_ragged(problems.shuttle(K=4000, order=4), 4000), device=None)
    plan = ShardPlan(eng, 8)
    assert plan.padding_fraction < 0.03
    for ip, (k0, _) in enumerate(plan.tiles):
        rs = [plan.tile_ranges[r][ip] for r in range(8)]
        assert rs[0][0] == 0 and rs[-1][1] == len(k0) - 1 and all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
