"""GPU tests of the mixed build (run with -m gpu): a phase whose sections differ in order runs order-specialised tile
bodies on the tiles the host cuts out of runs of equal sections and the any-order body on the rest (pc::bulk_mix,
pc_pattern.hpp::build_tiles_mixed).  Everything is checked through the C ABI against the CPU oracle on the reference's
own quadrature tables, entry by entry (conftest.entry_err), and against the launch variants that must return the same
bits: waves per tile, one launch / two launches, sharded tile ranges."""
import numpy as np
import pytest

from conftest import entry_err, golden_tables
from oracle.ref_numpy import OracleNlp
from pycollo_amd import problems
from pycollo_amd.refinement import synthetic_refined_mesh

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tab():
    return golden_tables("lobatto")


def run_mesh(seed=1):
    """Section orders with runs (4, 6, 5), a stretch of random orders, odd single sections and a short run; widths random.
    With bodies for (4, 6): order-pure tiles of both, tiles whose previous section has another order, any-order tiles."""
    rr = np.random.default_rng(seed)
    n = np.concatenate([np.full(30, 4), rr.integers(4, 9, 12), np.full(3, 4), np.full(20, 6), [7, 5], np.full(10, 5),
                        np.full(40, 4), [8], np.full(9, 6)]).astype(np.int64)
    return rr.uniform(0.5, 1.5, n.size), n


def mixed_problem(name, kw=None, seed=1):
    prob = problems.REGISTRY[name](**(kw or {}))
    for i, ph in enumerate(prob.phases):
        sizes, nodes = run_mesh(seed + i)
        ph.mesh.number_mesh_sections = nodes.size
        ph.mesh.mesh_section_sizes = sizes
        ph.mesh.number_mesh_section_nodes = nodes
    return prob


def _engine(prob, spec=(4, 6), **kw):
    from pycollo_amd.engine import NlpEngine
    return NlpEngine(prob, device=0, mixed=tuple(spec for _ in prob.phases), **kw)


def _point(eng, name, seed=5):
    rng = np.random.default_rng(seed)
    lo, hi = (0.05, 0.3) if name == "delta_iii" else (-0.45, 0.45)
    return rng.uniform(lo, hi, eng.num_x), rng.normal(size=eng.num_c)


def _assert_oracle(eng, ora, x, sigma, lam, out):
    c, G, H = out
    assert entry_err(c, ora.c(x), ora.c_mag(x)) <= 1.0
    assert entry_err(G, ora.G(x), ora.G_mag(x)) <= 1.0
    assert entry_err(H, ora.H(x, sigma, lam), ora.H_mag(x, sigma, lam)) <= 1.0
    for got, ref in ((eng.evaluate_G_structure(), ora.G_structure()), (eng.evaluate_H_structure(), ora.H_structure())):
        np.testing.assert_array_equal(got[0], ref[0])
        np.testing.assert_array_equal(got[1], ref[1])


MIXED_CASES = ["hypersensitive", "cart_pole", "shuttle", "two_phase_transfer", "delta_iii", "time_coupled_transfer",
               "double_pendulum"]


@pytest.mark.parametrize("tpb", [64, 256])
@pytest.mark.parametrize("name", MIXED_CASES)
def test_mixed_build_matches_oracle(built, tab, name, tpb):
    """Every entry of c~, G~, H~ of the mixed build within 1e-10 of the oracle; index arrays equal; the tiling holds
    order-pure tiles of both specialised orders and any-order tiles; J and grad J as well."""
    prob = mixed_problem(name)
    eng = _engine(prob, threads_per_block=tpb)
    orders = np.concatenate([eng.phase_tile_orders(p) for p in range(len(prob.phases))])
    assert {0, 4, 6} <= set(int(o) for o in orders)
    rng = np.random.default_rng(11)
    W = rng.uniform(0.5, 2.0, eng.layout.num_ocp_c)
    eng.set_scaling(eng.V_ocp, eng.r_ocp, W, 1.7)
    ora = OracleNlp(prob, tab, V_ocp=eng.V_ocp, r_ocp=eng.r_ocp, W_ocp=W, w_J=1.7)
    x, lam = _point(eng, name)
    _assert_oracle(eng, ora, x, 0.6, lam, eng.evaluate_all(x, 0.6, lam))
    assert abs(eng.evaluate_J(x) - ora.J(x)) <= 1e-10 * max(1.0, abs(ora.J(x)))
    g = eng.evaluate_g(x)
    assert np.max(np.abs(g - ora.grad_J(x))) <= 1e-10 * max(1.0, np.max(np.abs(ora.grad_J(x))))
    eng.close()


@pytest.mark.parametrize("name", ["shuttle", "delta_iii", "two_phase_transfer", "time_coupled_transfer"])
def test_mixed_build_waves_per_tile(built, tab, monkeypatch, name):
    """1, 2 or 4 waves per tile (replica index compiled in where the code object has that kernel): the same bits."""
    prob = mixed_problem(name)
    outs = {}
    for wpt in (1, 2, 4):
        monkeypatch.setenv("PYCOLLO_AMD_WPT", str(wpt))
        eng = _engine(prob, threads_per_block=64)
        assert eng.info["waves_per_tile"] == wpt
        x, lam = _point(eng, name)
        outs[wpt] = eng.evaluate_all(x, 0.6, lam)
        if wpt == 1:
            ora = OracleNlp(prob, tab, V_ocp=eng.V_ocp, r_ocp=eng.r_ocp, W_ocp=eng.W_ocp, w_J=1.0)
            _assert_oracle(eng, ora, x, 0.6, lam, outs[1])
        eng.close()
    for wpt in (2, 4):
        for a, b in zip(outs[1], outs[wpt]):
            np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("name", ["hypersensitive", "delta_iii", "time_coupled_transfer"])
def test_mixed_build_one_launch_equals_two(built, monkeypatch, name):
    """The resident tail (one launch) and bulk + pc_tail (two launches) of the mixed build write the same bits."""
    prob = mixed_problem(name)
    outs = {}
    for res in (1, 0):
        monkeypatch.setenv("PYCOLLO_AMD_RESIDENT", str(res))
        eng = _engine(prob)
        assert eng.info["n_launches"] == (1 if res else 2)
        x, lam = _point(eng, name)
        outs[res] = list(eng.evaluate_all(x, 0.6, lam)) + [np.array([eng.evaluate_J(x)]), eng.evaluate_g(x)]
        eng.close()
    for a, b in zip(outs[1], outs[0]):
        np.testing.assert_array_equal(a, b)


def test_mixed_build_against_any_order_kernel(built, tab):
    """The same mesh through the any-order kernel alone: values agree to rounding (the bodies fold h_k into the A
    column at different places), the tiling differs, the patterns do not."""
    from pycollo_amd.engine import NlpEngine
    prob = mixed_problem("shuttle")
    a = _engine(prob)
    b = NlpEngine(prob, device=0, mixed=None)
    assert not np.array_equal(a.phase_tiles(0)[0], b.phase_tiles(0)[0])
    x, lam = _point(a, "shuttle")
    ora = OracleNlp(prob, tab, V_ocp=a.V_ocp, r_ocp=a.r_ocp, W_ocp=a.W_ocp, w_J=1.0)
    for eng in (a, b):
        _assert_oracle(eng, ora, x, 0.6, lam, eng.evaluate_all(x, 0.6, lam))
    a.close()
    b.close()


@pytest.mark.parametrize("name,world", [("delta_iii", 3), ("hypersensitive", 4)])
def test_mixed_build_sharded_ranks_reassemble_bitwise(built, name, world):
    """Tile ranges of the mixed build on emulated ranks (the tile records follow the launched ranges): the merged
    result equals the unsharded evaluation bit for bit."""
    import torch
    from pycollo_amd.sharding import ShardPlan
    prob = mixed_problem(name)
    eng = _engine(prob)
    x, lam = _point(eng, name)
    c, G, H = (a.copy() for a in eng.evaluate_all(x, 0.9, lam))
    plan = ShardPlan(eng, world)
    dev = torch.device("cuda", 0)
    dx, dl = torch.from_numpy(x).to(dev), torch.from_numpy(lam).to(dev)
    oG, oH = plan.num_c, plan.num_c + plan.nnz_G
    s = torch.cuda.Stream(device=dev)
    merged = torch.full((plan.total,), float("nan"), dtype=torch.float64, device=dev)
    with torch.cuda.stream(s):
        for r in range(world):
            buf = torch.full((plan.total,), float("nan"), dtype=torch.float64, device=dev)
            for ip, ((k0, nred), off) in enumerate(zip(plan.tiles, plan.part_off)):
                if nred:
                    eng.set_partials_buffer(ip, buf[off:off + (len(k0) - 1) * nred])
                eng.set_tile_range(ip, *plan.tile_ranges[r][ip])
            eng.launch_bulk_only(dx, dl, buf[:oG], buf[oG:oH], buf[oH:oH + plan.nnz_H], s.cuda_stream)
            idx = torch.from_numpy(plan.index[r]).to(dev)
            merged[idx] = buf[idx]
            s.synchronize()
        for ip, ((k0, nred), off) in enumerate(zip(plan.tiles, plan.part_off)):
            if nred:
                eng.set_partials_buffer(ip, merged[off:off + (len(k0) - 1) * nred])
            eng.set_tile_range(ip, 0, len(k0) - 1)
        eng.launch_tail_only(dx, 0.9, dl, merged[:oG], merged[oG:oH], merged[oH:oH + plan.nnz_H], s.cuda_stream)
        s.synchronize()
    out = merged.cpu().numpy()
    assert np.array_equal(out[:oG], c)
    assert np.array_equal(out[oG:oH], G)
    assert np.array_equal(out[oH:oH + plan.nnz_H], H)
    for ip in range(len(plan.tiles)):
        eng.set_partials_buffer(ip, 0)
    eng.close()


def test_auto_mixed_on_a_refined_mesh(built, tab):
    """NlpEngine picks the mixed build by itself for a large mesh whose orders come in runs (the ph rule iterated on a
    synthetic error field, refinement.synthetic_refined_mesh): most tiles are order-pure, and every entry matches."""
    from pycollo_amd.engine import NlpEngine
    prob = problems.REGISTRY["hypersensitive"]()
    sizes, nodes = synthetic_refined_mesh(30000, seed=3)
    ph = prob.phases[0]
    ph.mesh.number_mesh_sections, ph.mesh.mesh_section_sizes, ph.mesh.number_mesh_section_nodes = nodes.size, sizes, nodes
    eng = NlpEngine(prob, device=0)
    assert eng.mixed[0], "the refined mesh has runs: the engine should have chosen the mixed build"
    od = eng.phase_tile_orders(0)
    assert np.mean(od > 0) > 0.8
    x, lam = _point(eng, "hypersensitive")
    ora = OracleNlp(prob, tab, V_ocp=eng.V_ocp, r_ocp=eng.r_ocp, W_ocp=eng.W_ocp, w_J=1.0)
    _assert_oracle(eng, ora, x, 1.0, lam, eng.evaluate_all(x, 1.0, lam))
    eng.close()
