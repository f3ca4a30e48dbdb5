"""Pin the CPU oracle: reference known answers, symbolic assembly, finite differences."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, entry_err, rel_err
from oracle import symbolic
from oracle.ref_numpy import GoldenTables, OracleNlp
from pycollo_amd import problems
from pycollo_amd.quadrature import QuadratureTables


@pytest.fixture(scope="module")
def golden_tab():
    return GoldenTables(os.path.join(GOLDEN, "quadrature_tables.npz"))


def test_brachistochrone_known_answers(golden_tab, known_answers):
    """tests/unit/test_iteration.py:305-318 (J), :339-354 (grad J), :371-385 (c) at EXPECT_X_TILDE_BR."""
    o = OracleNlp(problems.brachistochrone(), golden_tab)
    x = known_answers["EXPECT_X_TILDE_BR"]
    assert (o.num_x, o.num_c) == (125, 90)
    np.testing.assert_almost_equal(o.J(x), 0.8243386694458454)          # 7 decimals, as in the reference
    expect_g = np.zeros(125); expect_g[124] = 10
    np.testing.assert_allclose(o.grad_J(x), expect_g)
    np.testing.assert_allclose(o.c(x), np.zeros(90), atol=10e-2)          # the reference's own tolerance
    assert np.max(np.abs(o.c(x))) < 1e-8                                   # 8.1e-10


def test_double_pendulum_known_answers(golden_tab, known_answers):
    """tests/unit/test_iteration.py:290-302 (J == 100), :321-336 (grad J), :192-234 (counts / slices)."""
    o = OracleNlp(problems.double_pendulum(), golden_tab)
    x = known_answers["EXPECT_X_TILDE_DP"]
    assert (o.num_x, o.num_c) == (190, 121)
    P = o.P[0]
    assert (P.x_off, P.q_off, P.t_off, o.s_off) == (0, 186, 187, 188)
    assert (P.c_off, P.c_path, P.c_int, o.c_end) == (0, 120, 120, 121)
    assert o.J(x) == 100
    expect_g = np.zeros(190); expect_g[186] = 1000
    np.testing.assert_allclose(o.grad_J(x), expect_g)


@pytest.mark.parametrize("tag,prob", [("BR", problems.brachistochrone), ("DP", problems.double_pendulum)])
def test_scaling_vectors_known_answers(tag, prob, golden_tab, known_answers):
    """tests/unit/test_iteration_scaling.py:101-168: V, r expanded to the mesh; x = V x~ + r."""
    from pycollo_amd.engine import NlpEngine  # layout only (structure-only handle needs the built library)
    from pycollo_amd.layout import NlpLayout
    from pycollo_amd.mesh import build_phase_mesh
    from pycollo_amd.model import compile_model
    p = prob()
    model = compile_model(p)
    lay = NlpLayout(model, [build_phase_mesh(QuadratureTables("lobatto"), *p.phases[0].mesh.resolved())])
    V, r = lay.base_variable_scaling()
    np.testing.assert_allclose(lay.expand_x(V), known_answers[f"EXPECT_V_{tag}"])
    np.testing.assert_allclose(lay.expand_x(r), known_answers[f"EXPECT_R_{tag}"])
    np.testing.assert_allclose(1.0 / lay.expand_x(V), known_answers[f"EXPECT_V_INV_{tag}"])
    x = lay.expand_x(V) * known_answers[f"EXPECT_X_TILDE_{tag}"] + lay.expand_x(r)
    # EXPECT_X_TILDE is stored with 9 significant digits in the reference data file
    np.testing.assert_allclose(x, known_answers[f"EXPECT_X_{tag}"], rtol=1e-4, atol=1e-7)
    o = OracleNlp(p, golden_tab)
    np.testing.assert_allclose(o.V_ocp, V)
    np.testing.assert_allclose(o.r_ocp, r)


SYMBOLIC_CASES = [("brachistochrone", dict(K=2, order=3)), ("hypersensitive", dict(K=3, order=3)),
                  ("cart_pole", dict(K=2, order=3)), ("two_phase_transfer", dict(K=2, order=3)),
                  ("time_coupled_transfer", dict(K=2, order=3))]


@pytest.mark.parametrize("name,kw", SYMBOLIC_CASES)
def test_oracle_matches_symbolic_assembly(name, kw):
    """Values AND structural patterns of G / H equal the jacobian / hessian of the symbolically
    assembled c~(x~) -- what ca.jacobian(c_iter, x_var_iter) computes (backend.py:1676)."""
    tab = QuadratureTables("lobatto")
    prob = problems.REGISTRY[name](**kw)
    o = OracleNlp(prob, tab)
    rng = np.random.default_rng(7)
    o.W_ocp = rng.uniform(0.5, 2.0, o.num_ocp_c)
    o.w_J = 1.3
    x = rng.uniform(-0.45, 0.45, o.num_x)
    lam = rng.normal(size=o.num_c)
    sigma = 0.7
    xs, J, c = symbolic.assemble(prob, tab, o.V_ocp, o.r_ocp, o.W_ocp, o.w_J)
    assert len(xs) == o.num_x and len(c) == o.num_c
    sub = dict(zip(xs, x))
    assert entry_err(o.c(x), [float(e.subs(sub)) for e in c], o.c_mag(x), rtol=1e-12) <= 1.0   # entry by entry
    assert abs(float(J.subs(sub)) - o.J(x)) < 1e-12 * max(1.0, abs(o.J(x)))
    r, cc, v = symbolic.jacobian_triplets(xs, c, x)
    ro, co = o.G_structure()
    np.testing.assert_array_equal(r, ro)
    np.testing.assert_array_equal(cc, co)
    assert entry_err(o.G(x), v, o.G_mag(x), rtol=1e-12) <= 1.0
    r, cc, v = symbolic.hessian_triplets(xs, J, c, x, sigma, lam)
    ro, co = o.H_structure()
    np.testing.assert_array_equal(r, ro)
    np.testing.assert_array_equal(cc, co)
    assert entry_err(o.H(x, sigma, lam), v, o.H_mag(x, sigma, lam), rtol=1e-12) <= 1.0


@pytest.mark.parametrize("name,kw", [("shuttle", dict(K=4, order=4)), ("double_pendulum", dict(K=3, order=4)),
                                     ("delta_iii", dict(K=2, order=3))])
def test_oracle_finite_differences(name, kw):
    """Central differences of c~ (for G) and of G~^T lambda + sigma grad J (for H) at a benign point."""
    tab = QuadratureTables("lobatto")
    prob = problems.REGISTRY[name](**kw)
    o = OracleNlp(prob, tab)
    rng = np.random.default_rng(3)
    x = rng.uniform(0.05, 0.3, o.num_x)
    lam = rng.normal(size=o.num_c)
    sigma = 0.8
    import scipy.sparse as sp
    r, c = o.G_structure()
    G = sp.coo_matrix((o.G(x), (r, c)), shape=(o.num_c, o.num_x)).tocsc()
    hr, hc = o.H_structure()
    Hl = sp.coo_matrix((o.H(x, sigma, lam), (hr, hc)), shape=(o.num_x, o.num_x)).toarray()
    H = Hl + np.tril(Hl, -1).T
    cols = rng.choice(o.num_x, size=min(12, o.num_x), replace=False)
    for j in cols:
        e = np.zeros(o.num_x); e[j] = 1.0
        step = 1e-6
        dc = (o.c(x + step * e) - o.c(x - step * e)) / (2 * step)
        col = G[:, j].toarray().ravel()
        # truncation term + round-off of the difference quotient (eps * |c| / step)
        noise = 8 * np.finfo(float).eps * np.max(np.abs(o.c(x))) / step
        assert np.max(np.abs(dc - col)) <= 2e-5 * max(1.0, np.max(np.abs(col))) + noise, (name, j)

        def gradL(xx):
            rr, cc_ = o.G_structure()
            Gx = sp.coo_matrix((o.G(xx), (rr, cc_)), shape=(o.num_c, o.num_x)).tocsr()
            return Gx.T @ lam + sigma * o.grad_J(xx)
        dg = (gradL(x + step * e) - gradL(x - step * e)) / (2 * step)
        noise = 8 * np.finfo(float).eps * np.max(np.abs(gradL(x))) / step
        assert np.max(np.abs(dg - H[:, j])) <= 2e-4 * max(1.0, np.max(np.abs(H[:, j]))) + noise, (name, j)


def test_oracle_row_norms():
    o = OracleNlp(problems.cart_pole(K=3, order=4), QuadratureTables("lobatto"))
    x = np.random.default_rng(0).uniform(-0.4, 0.4, o.num_x)
    import scipy.sparse as sp
    r, c = o.G_structure()
    G = sp.coo_matrix((o.G(x), (r, c)), shape=(o.num_c, o.num_x)).toarray()
    np.testing.assert_allclose(o.G_row_norms(x), np.sqrt((G ** 2).sum(axis=1)), rtol=1e-13)
