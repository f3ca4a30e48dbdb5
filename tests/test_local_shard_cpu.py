"""CPU tests of the rank-local sharded evaluation's host side (pycollo_amd/sharding.py: LocalShard, global_tile_plan):
a rank's handle covers its section range and a one-section halo only, its segments correspond one to one to the global
plan's, its pattern IS the global pattern of those rows under its index maps, and its memory is its share.  No
evaluation happens here (no GPU); tests/test_gpu_local_shard.py runs the kernels."""
import numpy as np
import pytest

from pycollo_amd import problems
from pycollo_amd.engine import NlpEngine
from pycollo_amd.sharding import LocalShard, ShardPlan, global_tile_plan


def _ragged(prob, seed=5):
    rr = np.random.default_rng(seed)
    for ph in prob.phases:
        K = int(ph.mesh.number_mesh_sections)
        ph.mesh.mesh_section_sizes = rr.uniform(0.5, 1.5, K)
        ph.mesh.number_mesh_section_nodes = rr.integers(3, 8, K)
    return prob


CASES = [("two_phase_transfer", dict(K=40, order=4), 3, False), ("time_coupled_transfer", dict(K=60, order=4), 4, False),
         ("hypersensitive", dict(K=300, order=6), 8, False), ("delta_iii", dict(K=30, order=5), 2, False),
         ("shuttle", dict(K=90, order=4), 3, True), ("two_phase_transfer", dict(K=50, order=3), 4, True)]


@pytest.mark.parametrize("name,kw,world,ragged", CASES)
def test_local_pattern_is_the_global_pattern_of_the_owned_rows(built, name, kw, world, ragged):
    prob = problems.REGISTRY[name](**kw)
    if ragged:
        prob = _ragged(prob)
    g = NlpEngine(prob, device=None)
    plan = ShardPlan(g, world)
    gr, gc = g.evaluate_G_structure()
    hr, hc = g.evaluate_H_structure()
    oG, oH = g.num_c, g.num_c + g.nnz_jac
    covered = np.zeros(plan.total, dtype=np.int32)
    px = g.layout.point_x_index()
    point_pairs = set()
    for r_, c_, _ in g.model.point.hess:
        a_, b_ = int(px[r_]), int(px[c_])
        point_pairs.add((max(a_, b_), min(a_, b_)))
    for r in range(world):
        ls = LocalShard(prob, r, world, device=None)
        e = ls.engine
        # same tiles as the global plan's range, behind the halo tile
        for ip, (tb, te) in enumerate(ls.ranges):
            k0g = plan.tiles[ip][0]
            k0l, _ = e.phase_tiles(ip)
            if te > tb:
                assert plan.tile_ranges[r][ip] == (tb, te)
                own = np.asarray(k0l[ls.halo[ip]:len(k0l) - ls.halo_after[ip]])
                np.testing.assert_array_equal(own + ls.first_section[ip], k0g[tb:te + 1])
        # segments: one to one with the global plan's, same lengths
        assert [b - a for a, b in ls.segments] == [b - a for a, b in plan.segments[r]]
        lr, lc = e.evaluate_G_structure()
        lhr, lhc = e.evaluate_H_structure()
        lG, lH = e.num_c, e.num_c + e.nnz_jac
        for (la, lb), (ga, gb) in zip(ls.segments, plan.segments[r]):
            covered[ga:gb] += 1
            if la >= lH + e.nnz_hess:          # partial sums: positions only
                continue
            if la >= lH:                        # Hessian entries: (row, col) under the x map
                rows_g, cols_g = hr[ga - oH:gb - oH], hc[ga - oH:gb - oH]
                np.testing.assert_array_equal(ls.x_index[lhr[la - lH:lb - lH]], rows_g)
                # (an edge node's rows also hold the endpoint block's entries, which the tail of the WHOLE NLP writes:
                #  the rank's own endpoint block couples its own end nodes -- same slots, other columns)
                tile_written = np.array([(int(r_), int(c_)) not in point_pairs for r_, c_ in zip(rows_g, cols_g)])
                np.testing.assert_array_equal(ls.x_index[lhc[la - lH:lb - lH]][tile_written], cols_g[tile_written])
            elif la >= lG:                      # Jacobian entries: row under the lambda map, column under the x map
                np.testing.assert_array_equal(ls.lam_index[lr[la - lG:lb - lG]], gr[ga - oG:gb - oG])
                np.testing.assert_array_equal(ls.x_index[lc[la - lG:lb - lG]], gc[ga - oG:gb - oG])
            else:                               # constraint rows
                np.testing.assert_array_equal(ls.lam_index[la:lb], np.arange(ga, gb))
        ls.close()
    # every position a tile kernel writes is produced by exactly one rank (the rest is the tail's)
    assert covered.max() == 1
    ref = np.zeros(plan.total, dtype=np.int32)
    for r in range(world):
        ref[plan.index[r]] += 1
    np.testing.assert_array_equal(covered, ref)
    g.close()


def test_rank_local_memory_is_the_ranks_share(built):
    """Per rank: x~, lambda, outputs, send buffer and index arrays together stay within total / world plus a border (the
    halo section, the shared end node, the parameter rows and the endpoint block) -- what lets a mesh larger than one
    GPU's memory be evaluated at all."""
    world = 8
    prob = problems.shuttle(K=4000, order=4)
    g = NlpEngine(prob, device=None)
    whole = 8 * (g.num_x + g.num_c + (g.num_c + g.nnz_jac + g.nnz_hess)) + 8 * (g.nnz_jac + g.nnz_hess)
    plan = global_tile_plan(g.model, g.meshes)
    worst = 0
    for r in range(world):
        ls = LocalShard(g.model, r, world, device=None, meshes=g.meshes, plan=plan)
        worst = max(worst, ls.device_bytes())
        ls.close()
    # the send buffer repeats the owned outputs once: 2 x share + border
    assert worst <= 2.0 * whole / world + 0.02 * whole, (worst, whole)
    g.close()


def test_plan_only_handle_cuts_the_same_tiles(built):
    prob = _ragged(problems.two_phase_transfer(K=70, order=4))
    g = NlpEngine(prob, device=None)
    plan = global_tile_plan(g.model, g.meshes)
    for ip in range(len(g.model.phases)):
        np.testing.assert_array_equal(plan["tiles"][ip][0], g.phase_tiles(ip)[0])
        np.testing.assert_array_equal(plan["tiles"][ip][1], g.phase_tile_orders(ip))
    assert plan["threads_per_block"] == g.info["threads_per_block"]
    g.close()


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _local_worker(rank, world, port, name, kw, ragged, q):
    import os
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pycollo_amd.sharding import LocalShardedNlp, ShardPlan
        prob = problems.REGISTRY[name](**kw)
        if ragged:
            prob = _ragged(prob)
        sh = LocalShardedNlp(prob, device=None, root=0)
        ok = True
        # the root's x~ / lambda reach every rank as its own slices
        g = NlpEngine(prob, device=None)                  # (test side only: the reference values)
        x = torch.from_numpy(np.random.default_rng(1).normal(size=g.num_x))
        lam = torch.from_numpy(np.random.default_rng(2).normal(size=g.num_c))
        sh.distribute(x if sh.is_root else None, lam if sh.is_root else None)
        ok = ok and bool(np.array_equal(sh.x_local.numpy(), x.numpy()[sh.shard.x_index]))
        ok = ok and bool(np.array_equal(sh.lam_local.numpy(), lam.numpy()[sh.shard.lam_index]))
        # what the rank's kernels would produce, packed: here the global position of every packed entry stands for its value
        plan = ShardPlan(g, world)
        packed = torch.from_numpy(plan.index[rank].astype(np.float64))
        ok = ok and packed.numel() == sh.shard.length
        sh.collect(packed)
        if sh.is_root:
            allidx = np.concatenate(plan.index)
            ok = ok and bool(np.array_equal(sh.buf.numpy()[allidx], allidx.astype(np.float64)))
            ok = ok and sh.engine is not None
        else:
            ok = ok and sh.engine is None                 # only the root holds the whole NLP
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,kw,ragged,world", [("two_phase_transfer", dict(K=40, order=4), False, 2),
                                                  ("shuttle", dict(K=90, order=4), True, 3)])
def test_scatter_and_gather_of_the_rank_local_evaluation_gloo(built, name, kw, ragged, world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_local_worker, args=(r, world, port, name, kw, ragged, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert dict(q.get(timeout=10) for _ in range(world)) == {r: True for r in range(world)}
