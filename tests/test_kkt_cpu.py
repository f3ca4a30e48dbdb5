"""The block elimination order of the GPU KKT solver (pycollo_amd/kkt.py tables) executed with NumPy
(oracle/ref_kkt.py) against a general sparse solve: assembly, inertia and solution.  No GPU."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from conftest import golden_tables
from oracle.ref_kkt import RefKkt
from oracle.ref_numpy import OracleNlp
from pycollo_amd import kkt, problems


def kkt_case(name, kw, seed=0, device=None):
    """A KKT system of the shape the interior-point solver builds: some rows inequalities (slacks), some variables
    fixed, row scaling, Sigma on the primal diagonal."""
    from pycollo_amd.engine import NlpEngine
    prob = problems.REGISTRY[name](**kw)
    eng = NlpEngine(prob, device=device)
    ora = OracleNlp(prob, golden_tables(), V_ocp=eng.V_ocp, r_ocp=eng.r_ocp, W_ocp=eng.W_ocp)
    n, m = eng.num_x, eng.num_c
    rng = np.random.default_rng(seed)
    x = rng.uniform(-0.3, 0.3, n)
    lam = 0.1 * rng.normal(size=m)
    lay = eng.layout
    ineq = []
    for pl, pm in zip(lay.phases, eng.model.phases):
        ineq += list(range(pl.c_path_off, pl.c_path_off + pm.n_p * pl.N))
    ineq += list(range(lay.c_end_off, m, 2))
    ineq = np.array(sorted(ineq), dtype=np.int64)
    ns = len(ineq)
    fixed = np.zeros(n + ns, bool)
    fixed[rng.choice(n, size=max(1, n // 40), replace=False)] = True
    sc = rng.uniform(0.5, 1.0, m)
    dvec = np.concatenate([rng.uniform(0.5, 2.0, n + ns) + 50.0, -1e-8 * np.ones(m)])
    return eng, ora, x, lam, ineq, fixed, sc, dvec


def reference_matrix(eng, G, H, ineq, fixed, sc, dvec):
    n, m, ns = eng.num_x, eng.num_c, len(ineq)
    hr, hc = eng.evaluate_H_structure()
    jr, jc = eng.evaluate_G_structure()
    Hm = sp.coo_matrix((H, (hr, hc)), shape=(n, n))
    Hm = Hm + sp.triu(Hm.T, 1)
    W = sp.block_diag([Hm, sp.csr_matrix((ns, ns))])
    J = sp.hstack([sp.csr_matrix((G * sc[jr], (jr, jc)), shape=(m, n)),
                   sp.csr_matrix((-np.ones(ns), (ineq, np.arange(ns))), shape=(m, ns))]).tocsr()
    K = sp.bmat([[W + sp.diags(dvec[:n + ns]), J.T], [J, sp.diags(dvec[n + ns:])]]).tocsr()
    mask = np.ones(K.shape[0])
    mask[np.nonzero(fixed)[0]] = 0
    return (sp.diags(mask) @ K @ sp.diags(mask) + sp.diags(1 - mask)).tocsc()


CASES = [("hypersensitive", dict(K=30, order=6)), ("double_pendulum", {}), ("two_phase_transfer", {}),
         ("hypersensitive", dict(K=7, order=2)), ("brachistochrone", {}), ("sliding_mass", dict(num_phases=3, K=4, order=4)),
         ("free_flying_robot", dict(K=5, order=5)), ("shuttle", dict(K=6, order=4)), ("time_coupled_transfer", dict(K=9, order=4)),
         ("hypersensitive", dict(K=1, order=4)), ("hypersensitive", dict(K=2, order=3)), ("two_phase_transfer", dict(K=1, order=3))]


@pytest.mark.parametrize("group", [1, 3, None])
@pytest.mark.parametrize("name,kw", CASES)
def test_block_elimination_matches_a_general_sparse_solve(built, name, kw, group):
    """``group``: mesh sections per leaf (1: one leaf per section; None: the default by leaf size)."""
    eng, ora, x, lam, ineq, fixed, sc, dvec = kkt_case(name, kw)
    G, H = ora.G(x), ora.H(x, 1.0, lam)
    T = kkt.build_tables(eng, ineq, fixed, sc, group)
    assert T.n_primal == eng.num_x + len(ineq) and T.n_dual == eng.num_c
    R = RefKkt(T)
    K = reference_matrix(eng, G, H, ineq, fixed, sc, dvec)
    assert abs(R.assemble(G, H, dvec) - K).max() == 0.0
    npos, nneg = R.factor(G, H, dvec)
    ev = np.linalg.eigvalsh(K.toarray())
    assert (npos, nneg) == (int((ev > 0).sum()), int((ev < 0).sum()))     # Sylvester: pivot signs = inertia
    rng = np.random.default_rng(1)
    rhs = rng.normal(size=T.nu)
    rhs[np.nonzero(fixed)[0]] = 0.0
    # (the -dc I block makes the system ill-conditioned by design, ~1e9: both solvers get one step of iterative
    #  refinement, as the interior-point solver gives every step)
    xs = R.solve(rhs)
    xs = xs + R.solve(rhs - K @ xs)
    lu = spla.splu(K)
    xr = lu.solve(rhs)
    xr = xr + lu.solve(rhs - K @ xr)
    assert np.max(np.abs(xs - xr)) <= 1e-9 * np.max(np.abs(xr))
    np.testing.assert_allclose(R.matvec(G, H, dvec, xs), rhs, atol=1e-7 * np.max(np.abs(rhs)))
    eng.close()


@pytest.mark.parametrize("name,kw", CASES[:9])
def test_plan_positions_in_the_library_follow_the_stated_rule(built, name, kw):
    """``pc_kkt_plan_entries`` / ``pc_kkt_plan_positions`` (host C++, what ``build_tables`` uses) against the vectorised
    statement of the same rules kept beside them in ``kkt.py``: every table identical in value and type, for two leaf
    sizes -- the whole entry tables from the library, and the position rule alone."""
    import dataclasses
    eng, _, _, _, ineq, fixed, sc, _ = kkt_case(name, kw)
    for group in (1, None):
        B = kkt.build_tables(eng, ineq, fixed, sc, group, positions="numpy")
        for mode in ("library", "positions"):
            A = kkt.build_tables(eng, ineq, fixed, sc, group, positions=mode)
            for f in dataclasses.fields(A):
                a, b = getattr(A, f.name), getattr(B, f.name)
                assert np.array_equal(a, b), (mode, f.name)
                if isinstance(b, np.ndarray):
                    assert a.dtype == b.dtype, (mode, f.name, a.dtype, b.dtype)
