"""GPU test of the rank-local sharded evaluation (pycollo_amd/sharding.py, LocalShard / LocalRoot), ranks emulated one
after the other on the one GPU: every rank evaluates its section range on a handle built for that range alone (pattern,
tables, buffers of its share; the global tiling's tiles behind a halo tile), packs its segments, the root scatters them
into the whole NLP's buffer and runs the tail -- the result must equal the unsharded evaluation BIT FOR BIT."""
import numpy as np
import pytest

from pycollo_amd import problems

pytestmark = pytest.mark.gpu


def _ragged(prob, seed=5):
    rr = np.random.default_rng(seed)
    for ph in prob.phases:
        K = int(ph.mesh.number_mesh_sections)
        ph.mesh.mesh_section_sizes = rr.uniform(0.5, 1.5, K)
        ph.mesh.number_mesh_section_nodes = rr.integers(3, 8, K)
    return prob


def _runs(prob):
    from test_gpu_mixed import run_mesh
    for i, ph in enumerate(prob.phases):
        sizes, nodes = run_mesh(3 + i)
        ph.mesh.number_mesh_sections, ph.mesh.mesh_section_sizes, ph.mesh.number_mesh_section_nodes = nodes.size, sizes, nodes
    return prob


CASES = [("hypersensitive", dict(K=2000, order=6), 8, None, None),            # config 2, uniform order
         ("two_phase_transfer", dict(K=40, order=4), 3, None, None),
         ("time_coupled_transfer", dict(K=60, order=4), 4, None, None),        # q, t0, tF inside f, p, g
         ("delta_iii", dict(K=40, order=5), 3, None, None),                    # four phases, heavy model
         ("shuttle", dict(K=90, order=4), 3, _ragged, None),                   # orders differ: any-order kernels
         ("shuttle", dict(K=4, order=4), 3, _runs, ((4, 6),)),                 # mixed build: order-pure and any-order tiles
         ("cart_pole", dict(K=5000, order=4), 8, None, None)]                  # config 3


@pytest.mark.parametrize("name,kw,world,remesh,mixed", CASES)
def test_rank_local_evaluation_reassembles_bitwise(built, name, kw, world, remesh, mixed):
    import torch
    from pycollo_amd.engine import NlpEngine
    from pycollo_amd.sharding import LocalRoot, LocalShard, global_tile_plan
    prob = problems.REGISTRY[name](**kw)
    if remesh:
        prob = remesh(prob)
    root_eng = NlpEngine(prob, device=0, **({"mixed": mixed} if mixed else {}))
    rng = np.random.default_rng(4)
    lo, hi = (0.05, 0.3) if name == "delta_iii" else (-0.45, 0.45)
    x = rng.uniform(lo, hi, root_eng.num_x)
    lam = rng.normal(size=root_eng.num_c)
    W = rng.uniform(0.5, 2.0, root_eng.layout.num_ocp_c)
    root_eng.set_scaling(root_eng.V_ocp, root_eng.r_ocp, W, 1.3)
    c, G, H = (a.copy() for a in root_eng.evaluate_all(x, 0.9, lam))
    root = LocalRoot(root_eng, world)
    plan = root.plan
    dev = torch.device("cuda", 0)
    dx, dl = torch.from_numpy(x).to(dev), torch.from_numpy(lam).to(dev)
    oG, oH = plan.num_c, plan.num_c + plan.nnz_G
    merged = torch.full((plan.total,), float("nan"), dtype=torch.float64, device=dev)
    s = torch.cuda.Stream(device=dev)
    tile_plan = global_tile_plan(root_eng.model, root_eng.meshes, device=0, mixed=root_eng.mixed, orders=root_eng.orders)
    for ip in range(len(root_eng.model.phases)):          # the plan-only handle cut what the device handle cut
        np.testing.assert_array_equal(tile_plan["tiles"][ip][0], root_eng.phase_tiles(ip)[0])
    shares = []
    with torch.cuda.stream(s):
        for r in range(world):
            ls = LocalShard(root_eng.model, r, world, device=0, meshes=root_eng.meshes, plan=tile_plan)
            ls.set_scaling(root_eng.V_ocp, root_eng.r_ocp, W, 1.3)
            packed = ls.evaluate_packed(dx, dl, s.cuda_stream)
            s.synchronize()
            assert packed.numel() == plan.lengths[r]
            merged[torch.from_numpy(plan.index[r]).to(dev)] = packed
            shares.append(ls.device_bytes())
            s.synchronize()
            ls.close()
        for ip, ((k0, nred), off) in enumerate(zip(plan.tiles, plan.part_off)):
            if nred:
                root_eng.set_partials_buffer(ip, merged[off:off + (len(k0) - 1) * nred])
        root_eng.launch_tail_only(dx, 0.9, dl, merged[:oG], merged[oG:oH], merged[oH:oH + plan.nnz_H], s.cuda_stream)
        s.synchronize()
    out = merged.cpu().numpy()
    assert np.array_equal(out[:oG], c)
    assert np.array_equal(out[oG:oH], G)
    assert np.array_equal(out[oH:oH + plan.nnz_H], H)
    for ip in range(len(plan.tiles)):
        root_eng.set_partials_buffer(ip, 0)
    if world >= 8:      # a rank's footprint is its share: outputs twice (buffer + send buffer), indices, inputs, a border
        whole = 8 * (root_eng.num_x + 2 * root_eng.num_c + 2 * (root_eng.nnz_jac + root_eng.nnz_hess))
        assert max(shares) <= 2.0 * whole / world + 0.03 * whole
    root_eng.close()
