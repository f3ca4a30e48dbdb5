"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against the
CPU oracle on identical seeded inputs, the reference's known answers, and size-independent properties
at BASELINE.json's full sizes.

Tolerance (north_star): index arrays bit-exact; floating-point values within 1e-10 relative.
* G~ and H~ are compared ENTRY BY ENTRY (entry_err in conftest.py): |got - ref| <= 1e-10 |ref| + 64 eps S, S = the
  sum of the magnitudes of the terms the entry is made of (OracleNlp.G_mag / H_mag: the oracle's assembly rerun on
  magnitudes) -- the floor any evaluation order of a sum of products has.
* c~ is compared entry by entry the same way, S = OracleNlp.c_mag: a defect is a difference of large terms
  (y_0 - y_j + stretch h A f) -- near a solution pure cancellation -- so a row is held to the rounding of ITS terms,
  never to the largest row of the vector (a path row 1e6 times smaller than a defect must still be right).
The oracle is built on the reference's own quadrature tables (tests/golden/quadrature_tables.npz, orders 2..20);
the product computes its tables itself (pycollo_amd/quadrature.py).
"""
import numpy as np
import pytest

from conftest import entry_err, golden_tables, vec_err
from oracle.ref_numpy import OracleNlp
from pycollo_amd import problems
from pycollo_amd.quadrature import QuadratureTables

pytestmark = pytest.mark.gpu
TOL = 1e-10


@pytest.fixture(scope="module")
def tab():
    return golden_tables("lobatto")


def _engine(prob, **kw):
    from pycollo_amd.engine import NlpEngine
    return NlpEngine(prob, device=0, **kw)


def _check_all(eng, ora, seed=1, xlo=-0.45, xhi=0.45):
    rng = np.random.default_rng(seed)
    x = rng.uniform(xlo, xhi, eng.num_x)
    lam = rng.normal(size=eng.num_c)
    sigma = 0.6
    c, G, H = eng.evaluate_all(x, sigma, lam)
    cr, Gr, Hr = ora.c(x), ora.G(x), ora.H(x, sigma, lam)
    Gm, Hm, cm = ora.G_mag(x), ora.H_mag(x, sigma, lam), ora.c_mag(x)
    assert entry_err(c, cr, cm) <= 1.0
    assert entry_err(G, Gr, Gm) <= 1.0
    assert entry_err(H, Hr, Hm) <= 1.0
    # the separate IPOPT callbacks agree with the fused call
    assert entry_err(eng.evaluate_c(x), cr, cm) <= 1.0
    assert entry_err(eng.evaluate_G_nonzeros(x, new_x=False), Gr, Gm) <= 1.0
    assert entry_err(eng.evaluate_H_nonzeros(x, sigma, lam), Hr, Hm) <= 1.0
    assert abs(eng.evaluate_J(x) - ora.J(x)) <= TOL * max(1.0, abs(ora.J(x)))
    assert vec_err(eng.evaluate_g(x), ora.grad_J(x)) <= 1.0
    for got, ref in ((eng.evaluate_G_structure(), ora.G_structure()), (eng.evaluate_H_structure(), ora.H_structure())):
        np.testing.assert_array_equal(got[0], ref[0])
        np.testing.assert_array_equal(got[1], ref[1])
    return x, lam


CASES = [("brachistochrone", {}), ("hypersensitive", dict(K=2000, order=6)), ("cart_pole", dict(K=500, order=4)),
         ("shuttle", dict(K=60, order=5)), ("double_pendulum", {}), ("two_phase_transfer", {}),
         ("delta_iii", dict(K=9, order=4)), ("free_flying_robot", dict(K=33, order=5)), ("sliding_mass", dict(num_phases=3, K=7, order=4)),
         ("tumour_anti_angiogenesis", dict(K=21, order=6)), ("space_station", dict(K=12, order=4)),
         ("time_coupled_transfer", {}), ("time_coupled_transfer", dict(K=300, order=5))]   # q / t0 / tF inside f, p, g


@pytest.mark.parametrize("tpb", [64, 256])
@pytest.mark.parametrize("name,kw", CASES)
def test_parity_with_oracle(built, tab, name, kw, tpb):
    prob = problems.REGISTRY[name](**kw)
    eng = _engine(prob, threads_per_block=tpb)
    rng = np.random.default_rng(11)
    W = rng.uniform(0.5, 2.0, eng.layout.num_ocp_c)
    eng.set_scaling(eng.V_ocp, eng.r_ocp, W, 1.7)
    ora = OracleNlp(prob, tab, V_ocp=eng.V_ocp, r_ocp=eng.r_ocp, W_ocp=W, w_J=1.7)
    lo, hi = (0.05, 0.3) if name == "delta_iii" else (-0.45, 0.45)   # keep |r| away from 0 for mu/r^3
    _check_all(eng, ora, xlo=lo, xhi=hi)
    eng.close()


@pytest.mark.parametrize("name,kw", [("shuttle", dict(K=60, order=5)), ("delta_iii", dict(K=9, order=4)),
                                     ("two_phase_transfer", {}), ("time_coupled_transfer", dict(K=12, order=4))])
def test_waves_per_tile_replicas(built, tab, monkeypatch, name, kw):
    """64-node tiles shared by 1, 2 or 4 waves (pc::bulk, `wpt`): every split writes the same bits, and the
    automatic choice is one of them."""
    prob = problems.REGISTRY[name](**kw)
    lo, hi = (0.05, 0.3) if name == "delta_iii" else (-0.45, 0.45)
    outs = {}
    for wpt in (1, 2, 4):
        monkeypatch.setenv("PYCOLLO_AMD_WPT", str(wpt))
        eng = _engine(prob, threads_per_block=64)
        assert eng.info["waves_per_tile"] == wpt
        if wpt == 1:
            ora = OracleNlp(prob, tab, V_ocp=eng.V_ocp, r_ocp=eng.r_ocp, W_ocp=eng.W_ocp, w_J=1.0)
        x, lam = _check_all(eng, ora, seed=3, xlo=lo, xhi=hi)
        outs[wpt] = eng.evaluate_all(x, 0.6, lam)
        eng.close()
    for wpt in (2, 4):
        for a, b in zip(outs[1], outs[wpt]):
            np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("name,order,K", [("hypersensitive", 20, 7), ("hypersensitive", 13, 11), ("cart_pole", 20, 5),
                                          ("brachistochrone", 2, 9), ("shuttle", 11, 1)])
def test_extreme_orders(built, tab, name, order, K):
    """The largest supported section order (20 nodes, pycollo/quadrature.py order range), orders past the
    coefficient-hoisting limit, the smallest one (2 nodes: every node is a section boundary) and a single section;
    both the order-specialised and the any-mesh kernel."""
    for specialise in (True, False):
        prob = problems.REGISTRY[name](K=K, order=order)
        eng = _engine(prob, threads_per_block=64, specialise=specialise)
        ora = OracleNlp(prob, tab, V_ocp=eng.V_ocp, r_ocp=eng.r_ocp, W_ocp=eng.W_ocp, w_J=1.0)
        _check_all(eng, ora, seed=order + K)
        eng.close()


def test_bound_device_call(built):
    """NlpEngine.bind_device: the pre-resolved form of evaluate_all_device writes the same bits."""
    import torch
    prob = problems.cart_pole(K=200, order=4)
    eng = _engine(prob)
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(21)
    x = torch.from_numpy(rng.uniform(-0.4, 0.4, eng.num_x)).to(dev)
    lam = torch.from_numpy(rng.normal(size=eng.num_c)).to(dev)
    out = [[torch.full((n,), float("nan"), dtype=torch.float64, device=dev) for n in (eng.num_c, eng.nnz_jac, eng.nnz_hess)]
           for _ in range(2)]
    s = torch.cuda.Stream(device=dev)
    eng.evaluate_all_device(x, 0.7, lam, *out[0], s.cuda_stream)
    eng.bind_device(x, lam, *out[1], s.cuda_stream)(0.7)
    s.synchronize()
    for a, b in zip(*out):
        assert torch.equal(a, b) and not torch.isnan(a).any()
    eng.close()


def test_known_answers_brachistochrone(built, known_answers):
    """tests/unit/test_iteration.py:305-318, 339-354, 371-385 evaluated by the HIP path."""
    eng = _engine(problems.brachistochrone())
    x = known_answers["EXPECT_X_TILDE_BR"]
    np.testing.assert_almost_equal(eng.evaluate_J(x), 0.8243386694458454)
    g = np.zeros(125); g[124] = 10
    np.testing.assert_allclose(eng.evaluate_g(x), g)
    c = eng.evaluate_c(x)
    np.testing.assert_allclose(c, np.zeros(90), atol=10e-2)
    assert np.max(np.abs(c)) < 1e-8
    eng.close()


def test_known_answers_double_pendulum(built, known_answers):
    """tests/unit/test_iteration.py:290-302, 321-336."""
    eng = _engine(problems.double_pendulum())
    x = known_answers["EXPECT_X_TILDE_DP"]
    assert eng.evaluate_J(x) == 100
    g = np.zeros(190); g[186] = 1000
    np.testing.assert_allclose(eng.evaluate_g(x), g)
    eng.close()


def test_ragged_and_mixed_order_mesh(built, tab):
    """ph-refined style mesh: non-uniform section sizes, orders 2..10 mixed, tile boundaries anywhere."""
    rng = np.random.default_rng(5)
    K = 157
    prob = problems.two_phase_transfer()
    A, B = prob.phases
    A.mesh.number_mesh_sections = K
    A.mesh.mesh_section_sizes = rng.uniform(0.2, 1.0, K)
    A.mesh.number_mesh_section_nodes = rng.integers(2, 11, K)
    B.mesh.number_mesh_sections = 3
    B.mesh.mesh_section_sizes = [0.2, 0.5, 0.3]
    B.mesh.number_mesh_section_nodes = [10, 2, 7]
    for tpb in (64, 128):
        eng = _engine(prob, threads_per_block=tpb)
        ora = OracleNlp(prob, tab, V_ocp=eng.V_ocp, r_ocp=eng.r_ocp, W_ocp=eng.W_ocp, w_J=1.0)
        _check_all(eng, ora, seed=tpb)
        eng.close()


def test_smallest_meshes(built, tab):
    """K = 1 with n = 2 (N = 2: both nodes are endpoints) and K = 1, n = 3."""
    for K, n in ((1, 2), (1, 3), (2, 2)):
        prob = problems.two_phase_transfer(K=K, order=n)
        prob.phases[1].mesh.number_mesh_sections = K
        prob.phases[1].mesh.number_mesh_section_nodes = n
        eng = _engine(prob)
        ora = OracleNlp(prob, tab, V_ocp=eng.V_ocp, r_ocp=eng.r_ocp, W_ocp=eng.W_ocp, w_J=1.0)
        _check_all(eng, ora, seed=K * 10 + n)
        eng.close()


def test_radau_tables_as_data(built):
    """tables are data -- the Radau scheme (weights summing to 2) runs through unchanged."""
    prob = problems.cart_pole(K=12, order=5)
    prob.quadrature_method = "radau"
    eng = _engine(prob)
    ora = OracleNlp(prob, golden_tables("radau"), V_ocp=eng.V_ocp, r_ocp=eng.r_ocp, W_ocp=eng.W_ocp, w_J=1.0)
    _check_all(eng, ora)
    eng.close()


def test_scaling_none_and_reset(built, tab):
    prob = problems.cart_pole(K=20, order=4)
    prob.scaling_method = None
    eng = _engine(prob)
    assert np.all(eng.V_ocp == 1) and np.all(eng.r_ocp == 0)
    ora = OracleNlp(prob, tab, V_ocp=eng.V_ocp, r_ocp=eng.r_ocp, W_ocp=eng.W_ocp, w_J=1.0)
    x, lam = _check_all(eng, ora)
    # changing the scaling invalidates the cached c / G of the previous x
    W2 = 2.0 * eng.W_ocp
    c1 = eng.evaluate_c(x)
    eng.set_scaling(eng.V_ocp, eng.r_ocp, W2, 1.0)
    np.testing.assert_array_equal(eng.evaluate_c(x, new_x=False), 2.0 * c1)   # a factor 2 is exact
    eng.close()


def test_row_norms_and_constraint_scaling(built, tab):
    """pycollo/scaling.py:370-430: W from the row norms of G evaluated with W = 1, without densifying."""
    from pycollo_amd.scaling import constraint_scaling
    prob = problems.two_phase_transfer(K=9, order=4)
    eng = _engine(prob)
    ones = np.ones(eng.layout.num_ocp_c)
    eng.set_scaling(eng.V_ocp, eng.r_ocp, ones, 1.0)
    ora = OracleNlp(prob, tab, V_ocp=eng.V_ocp, r_ocp=eng.r_ocp, W_ocp=ones, w_J=1.0)
    x = np.random.default_rng(2).uniform(-0.4, 0.4, eng.num_x)
    rn = eng.G_row_norms(x)
    assert vec_err(rn, ora.G_row_norms(x)) <= 1.0   # row by row
    W = constraint_scaling(eng, x)
    # reference formula restated with the oracle's norms
    lay = eng.layout
    ref = np.empty(lay.num_ocp_c)
    for pl, pm in zip(lay.phases, eng.model.phases):
        o = pl.ocp_c_off
        ref[o:o + pm.n_y] = 1.0 / eng.V_ocp[pl.ocp_x_off:pl.ocp_x_off + pm.n_y]
        pn = ora.G_row_norms(x)[pl.c_path_off:pl.c_int_off].reshape(pm.n_p, pl.N)
        ref[o + pm.n_y:o + pm.n_y + pm.n_p] = 1.0 / pn.mean(axis=1)
        ref[o + pm.n_y + pm.n_p:o + pm.n_y + pm.n_p + pm.n_q] = 1.0 / eng.V_ocp[pl.ocp_x_off + pm.n_z:pl.ocp_x_off + pm.n_z + pm.n_q]
    ref[lay.ocp_c_end_off:] = 1.0 / ora.G_row_norms(x)[lay.c_end_off:]
    np.testing.assert_allclose(W, ref, rtol=1e-10)
    eng.close()


def test_nan_inf_pass_through(built):
    """IPOPT semantics: NaN / Inf are written through, never trapped."""
    eng = _engine(problems.hypersensitive(K=8, order=4))
    x = np.zeros(eng.num_x)
    x[3] = np.nan
    c = eng.evaluate_c(x)
    assert np.isnan(c).any() and np.isfinite(c).any()
    eng.close()


def test_device_resident_api_matches_host_api(built):
    import torch
    eng = _engine(problems.cart_pole(K=300, order=4))
    rng = np.random.default_rng(9)
    x = rng.uniform(-0.4, 0.4, eng.num_x)
    lam = rng.normal(size=eng.num_c)
    c, G, H = eng.evaluate_all(x, 0.3, lam)
    dev = torch.device("cuda", 0)
    dx, dl = torch.from_numpy(x).to(dev), torch.from_numpy(lam).to(dev)
    dc = torch.empty(eng.num_c, dtype=torch.float64, device=dev)
    dG = torch.empty(eng.nnz_jac, dtype=torch.float64, device=dev)
    dH = torch.empty(eng.nnz_hess, dtype=torch.float64, device=dev)
    s = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    eng.evaluate_all_device(dx, 0.3, dl, dc, dG, dH, s.cuda_stream)
    s.synchronize()
    # bitwise: same kernels, same inputs, fixed reduction order
    assert np.array_equal(dc.cpu().numpy(), c) and np.array_equal(dG.cpu().numpy(), G) and np.array_equal(dH.cpu().numpy(), H)
    eng.close()


# ---- BASELINE.json full sizes: size-independent properties --------------------------------------
def _ph_refined(prob, K, seed=7, lo=4, hi=9):
    """A mesh as ph refinement leaves it (pycollo/mesh_refinement.py:250-392): section widths and orders differ from
    section to section (widths U(0.5, 1.5) before normalisation, lo..hi-1 nodes)."""
    rr = np.random.default_rng(seed)
    for ph in prob.phases:
        ph.mesh.mesh_section_sizes = rr.uniform(0.5, 1.5, K)
        ph.mesh.number_mesh_section_nodes = rr.integers(lo, hi, K)
    return prob


def _full_problem(name, kw):
    kw = dict(kw)
    refined = kw.pop("refined", False)
    ph_nodes = kw.pop("ph_nodes", 0)
    prob = problems.REGISTRY[name](**kw)
    if ph_nodes:   # the ph rule iterated on a synthetic error field: orders in runs -> the mixed build (tests/test_gpu_mixed.py)
        return problems.with_refined_mesh(prob, ph_nodes)
    return _ph_refined(prob, kw["K"]) if refined else prob


# configs[1..4] of BASELINE.json at full size; config 5 (Delta III, 4 phases, ~50 k nodes) twice: a uniform
# 4 x 3125 x 5 mesh (4 x 12 501 nodes, the merged multi-phase launch at 800+ tiles) and a ph-refined mesh
# (4 x 2500 sections of 4..8 nodes, ~50 k nodes: the any-order kernels)
FULL = [("hypersensitive", dict(K=2000, order=6)), ("cart_pole", dict(K=5000, order=4)),
        ("shuttle", dict(K=20000, order=4)), ("delta_iii", dict(K=3125, order=5)),
        ("delta_iii", dict(K=2500, order=4, refined=True)),      # orders 4..8 at random, section by section: any-order kernels
        ("delta_iii", dict(ph_nodes=12500))]                      # a mesh as the ph rule leaves it, ~50 k nodes: the mixed build


@pytest.mark.parametrize("name,kw", FULL)
def test_full_size_properties(built, tab, name, kw):
    """BASELINE.json's configs at full size: (1) parity with the vectorised oracle (it still finishes in seconds),
    (2) run-to-run bit reproducibility, (3) linearity of H in (sigma, lambda), (4) G^T lambda contraction
    equals the directional derivative of lambda.c (checksum of the whole Jacobian against c alone)."""
    prob = _full_problem(name, kw)
    eng = _engine(prob)
    if name == "delta_iii":
        assert sum(pl.N for pl in eng.layout.phases) > 49000
    if "ph_nodes" in kw:
        assert all(eng.mixed) and eng.info["waves_per_tile"] == 2      # mixed build, two waves per tile
        od = np.concatenate([eng.phase_tile_orders(p) for p in range(len(prob.phases))])
        assert np.mean(od > 0) > 0.85
    ora = OracleNlp(prob, tab, V_ocp=eng.V_ocp, r_ocp=eng.r_ocp, W_ocp=eng.W_ocp, w_J=1.0)
    rng = np.random.default_rng(1234)
    lo, hi = (0.05, 0.3) if name == "delta_iii" else (-0.45, 0.45)   # keep |r| away from 0 for mu/r^3
    x = rng.uniform(lo, hi, eng.num_x)
    lam = np.random.default_rng(1235).normal(size=eng.num_c)
    c, G, H = eng.evaluate_all(x, 1.0, lam)
    assert entry_err(c, ora.c(x), ora.c_mag(x)) <= 1.0
    assert entry_err(G, ora.G(x), ora.G_mag(x)) <= 1.0
    Hm = ora.H_mag(x, 1.0, lam)
    assert entry_err(H, ora.H(x, 1.0, lam), Hm) <= 1.0
    for got, ref in ((eng.evaluate_G_structure(), ora.G_structure()), (eng.evaluate_H_structure(), ora.H_structure())):
        np.testing.assert_array_equal(got[0], ref[0])
        np.testing.assert_array_equal(got[1], ref[1])
    c2, G2, H2 = eng.evaluate_all(x, 1.0, lam)
    assert np.array_equal(c, c2) and np.array_equal(G, G2) and np.array_equal(H, H2)
    # linearity: H(2 sigma, 2 lam) = 2 H(sigma, lam);  H(s1+s2, l1+l2) = H(s1,l1) + H(s2,l2)
    lam2 = rng.normal(size=eng.num_c)
    Ha = eng.evaluate_H_nonzeros(x, 0.25, lam2)
    Hs = eng.evaluate_H_nonzeros(x, 1.25, lam + lam2)
    # entry by entry, each held to the rounding of its own terms (their magnitudes add: |lam| + |lam2|, 1 + 0.25)
    assert entry_err(Hs, H + Ha, Hm + ora.H_mag(x, 0.25, lam2), rtol=1e-12, ulps=256) <= 1.0
    # checksum: d/d eps [lam . c(x + eps d)] = lam^T G d
    import scipy.sparse as sp
    r, cc = eng.evaluate_G_structure()
    d = rng.normal(size=eng.num_x)
    Gm = sp.csr_matrix((G, (r, cc)), shape=(eng.num_c, eng.num_x))
    eps = 1e-7
    fd = (lam @ eng.evaluate_c(x + eps * d) - lam @ eng.evaluate_c(x - eps * d)) / (2 * eps)
    an = lam @ (Gm @ d)
    assert abs(fd - an) <= 1e-5 * max(1.0, abs(an)) + 1e3 * np.finfo(float).eps * np.abs(lam @ c) / eps
    eng.close()


@pytest.mark.parametrize("name,kw,world", [("two_phase_transfer", dict(K=40, order=4), 3),
                                           ("time_coupled_transfer", dict(K=40, order=4), 3),
                                           ("hypersensitive", dict(K=2000, order=6), 8),
                                           ("delta_iii", dict(K=40, order=4), 2),
                                           ("shuttle", dict(K=20000, order=4), 8),              # config 4 as BASELINE shards it
                                           ("delta_iii", dict(K=3125, order=5), 8),             # config 5, uniform mesh
                                           ("delta_iii", dict(K=2500, order=4, refined=True), 8),    # config 5, random orders
                                           ("delta_iii", dict(ph_nodes=12500), 8)])                  # config 5, ph-refined: mixed build
def test_sharded_ranks_reassemble_bitwise(built, name, kw, world):
    """Emulate `world` ranks on one GPU: each rank's bulk kernels run over its tile range into a NaN-filled
    buffer, the plan's segments are merged (what the all-gather + unpack do), the tail runs on the merged
    buffer -- the result must equal the unsharded evaluation bit for bit."""
    import torch
    from pycollo_amd.sharding import ShardPlan
    prob = _full_problem(name, kw)
    eng = _engine(prob)
    rng = np.random.default_rng(4)
    lo, hi = (0.05, 0.3) if name == "delta_iii" else (-0.45, 0.45)
    x = rng.uniform(lo, hi, eng.num_x)
    lam = rng.normal(size=eng.num_c)
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


@pytest.mark.parametrize("name,kw,world", [("shuttle", dict(K=300, order=5), 3), ("delta_iii", dict(K=40, order=4), 2)])
def test_exchange_run_copy_tables(built, name, kw, world):
    """The GPU pack / unpack of SegmentExchange (pc_copy_runs over run tables) against the index form of the same
    plan: every rank packs its runs into its send buffer, the buffers are copied into the ranks' receive buffers
    (what the all-gather does), every rank unpacks -- all ranks must hold the same complete buffer."""
    import torch
    from pycollo_amd.sharding import SegmentExchange, ShardPlan
    prob = problems.REGISTRY[name](**kw)
    eng = _engine(prob)
    plan = ShardPlan(eng, world)
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(12)
    truth = torch.from_numpy(rng.normal(size=plan.total)).to(dev)
    covered = np.zeros(plan.total, bool)
    for r in range(world):
        covered[plan.index[r]] = True
    bufs, exs = [], []
    for r in range(world):
        b = torch.full((plan.total,), float("nan"), dtype=torch.float64, device=dev)
        idx = torch.from_numpy(plan.index[r]).to(dev)
        b[idx] = truth[idx]                       # what rank r's bulk kernels would have produced
        bufs.append(b)
        exs.append(SegmentExchange(plan, r, dev))
    for r in range(world):
        exs[r]._copy_runs(bufs[r], exs[r].send, exs[r].pack_tab)
    torch.cuda.synchronize()
    ml = exs[0].maxlen
    for r in range(world):                        # the all-gather, by hand
        for q in range(world):
            exs[r].recv[q * ml:(q + 1) * ml] = exs[q].send
    for r in range(world):
        exs[r]._copy_runs(exs[r].recv, bufs[r], exs[r].unpack_tab)
    torch.cuda.synchronize()
    want = truth.cpu().numpy()
    for r in range(world):
        got = bufs[r].cpu().numpy()
        assert np.array_equal(got[covered], want[covered])
        assert np.isnan(got[~covered]).all()
    eng.close()


def test_sharded_world1_nccl(built):
    """ShardedNlp end to end with a 1-rank RCCL group (the only group size one GPU allows)."""
    import os
    import torch
    import torch.distributed as dist
    from pycollo_amd.sharding import ShardedNlp
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29571")
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        prob = problems.cart_pole(K=400, order=4)
        sh = ShardedNlp(prob, device=0)
        eng = _engine(prob)
        rng = np.random.default_rng(8)
        x, lam = rng.uniform(-0.4, 0.4, eng.num_x), rng.normal(size=eng.num_c)
        c, G, H = eng.evaluate_all(x, 1.0, lam)
        dc, dG, dH = sh.evaluate_all_device(torch.from_numpy(x).to(dev), 1.0, torch.from_numpy(lam).to(dev))
        torch.cuda.synchronize()
        assert np.array_equal(dc.cpu().numpy(), c) and np.array_equal(dG.cpu().numpy(), G) and np.array_equal(dH.cpu().numpy(), H)
    finally:
        dist.destroy_process_group()


RES_CASES = [("hypersensitive", dict(K=2000, order=6), 0), ("hypersensitive", dict(K=700, order=6), 256),
             ("double_pendulum", {}, 0), ("delta_iii", dict(K=40, order=4), 0), ("delta_iii", dict(K=9, order=4), 128),
             ("two_phase_transfer", {}, 0), ("time_coupled_transfer", dict(K=30, order=4), 0), ("space_station", dict(K=12, order=4), 0), ("shuttle", dict(K=60, order=5), 0),
             ("tumour_anti_angiogenesis", dict(K=21, order=6), 64), ("sliding_mass", dict(num_phases=3, K=7, order=4), 0)]


@pytest.mark.parametrize("name,kw,tpb", RES_CASES)
def test_resident_tail_writes_the_same_bits_as_two_launches(built, monkeypatch, name, kw, tpb):
    """One launch per evaluation (the tail as block 0 of the bulk launch, fed by granules: pc_kernels.hpp RES) against
    bulk launch + pc_tail (PYCOLLO_AMD_RESIDENT=0): identical bits for c~, G~, H~, J, grad J on every flag
    combination the callbacks use, repeatedly (the granule tags change from launch to launch)."""
    prob = problems.REGISTRY[name](**kw)
    lo, hi = (0.05, 0.3) if name == "delta_iii" else (-0.45, 0.45)
    outs = {}
    for res in (1, 0):
        monkeypatch.setenv("PYCOLLO_AMD_RESIDENT", str(res))
        eng = _engine(prob, threads_per_block=tpb)
        assert eng.info["n_launches"] == (1 if res else 2)
        rng = np.random.default_rng(17)
        got = []
        for rep in range(3):
            x = rng.uniform(lo, hi, eng.num_x)
            lam = rng.normal(size=eng.num_c)
            got += list(eng.evaluate_all(x, 0.6, lam))
            got += [eng.evaluate_c(x), eng.evaluate_G_nonzeros(x, new_x=False), np.array([eng.evaluate_J(x, new_x=False)]),
                    eng.evaluate_g(x, new_x=False), eng.evaluate_H_nonzeros(x, 0.3, lam, new_x=False)]
        outs[res] = got
        eng.close()
    for a, b in zip(outs[1], outs[0]):
        np.testing.assert_array_equal(a, b)
