"""Row N4 on the GPU: the block L D L^T kernels (pc_kkt_*) against a general sparse solve and against the NumPy
execution of the same tables, and the interior-point solver with its linear algebra on the device against the same
solver with SuperLU on the host (reference: the linear solver inside IPOPT, pycollo/backend.py:1703-1711)."""
import numpy as np
import pytest
import scipy.sparse.linalg as spla

from oracle.ref_kkt import RefKkt
from pycollo_amd import problems
from test_kkt_cpu import kkt_case, reference_matrix

pytestmark = pytest.mark.gpu

CASES = [("hypersensitive", dict(K=2000, order=6)), ("double_pendulum", {}), ("two_phase_transfer", {}),
         ("hypersensitive", dict(K=7, order=2)), ("sliding_mass", dict(num_phases=3, K=4, order=4)),
         ("free_flying_robot", dict(K=33, order=5)), ("shuttle", dict(K=60, order=4)), ("cart_pole", dict(K=500, order=4)),
         ("time_coupled_transfer", dict(K=9, order=4)),
         ("hypersensitive", dict(K=1, order=4)), ("hypersensitive", dict(K=2, order=3)), ("two_phase_transfer", dict(K=1, order=3))]


@pytest.mark.parametrize("name,kw", CASES)
def test_factor_and_solve_match_superlu(built, name, kw):
    from pycollo_amd.kkt import GpuKkt
    eng, ora, x, lam, ineq, fixed, sc, dvec = kkt_case(name, kw, device=0)
    _, _, _ = eng.evaluate_resident(x, 1.0, lam)              # c~, G~, H~ stay on the device
    c, G, H = eng.evaluate_all(x, 1.0, lam)                   # host copies of the same bits, for the reference matrix
    eng.evaluate_resident(x, 1.0, lam)
    k = GpuKkt(eng, ineq, fixed, sc)
    K = reference_matrix(eng, G, H, ineq, fixed, sc, dvec)
    npos, nneg = k.factor(dvec)
    rng = np.random.default_rng(1)
    rhs = rng.normal(size=k.nu)
    rhs[np.nonzero(fixed)[0]] = 0.0
    # products
    y = k.matvec(dvec, rhs)
    assert np.max(np.abs(y - K @ rhs)) <= 1e-12 * np.max(np.abs(K @ rhs))
    # one step of refinement on both sides (the -dc I block makes the system ill-conditioned by design)
    xs = k.solve(rhs)
    xs = xs + k.solve(rhs - k.matvec(dvec, xs))
    lu = spla.splu(K)
    xr = lu.solve(rhs)
    xr = xr + lu.solve(rhs - K @ xr)
    assert np.max(np.abs(xs - xr)) <= 1e-9 * np.max(np.abs(xr))
    # inertia: the pivot signs of the same elimination order on the CPU
    if k.nu < 6000:
        assert (npos, nneg) == RefKkt(k.tables).factor(G, H, dvec)
    # bit-reproducible: no atomics, fixed order
    k.factor(dvec)
    np.testing.assert_array_equal(k.solve(rhs), k.solve(rhs))
    k.close()
    eng.close()


@pytest.mark.parametrize("name,kw", [("hypersensitive", dict(K=300, order=6)), ("two_phase_transfer", {}), ("shuttle", dict(K=60, order=4))])
def test_cyclic_reduction_and_sequential_chain_agree(built, monkeypatch, name, kw):
    """The chain eliminated level by level (default) and node by node (PYCOLLO_AMD_KKT_CR=0, the fallback for blocks
    too large for the level kernels' LDS): same inertia, same solution to rounding."""
    from pycollo_amd.kkt import GpuKkt
    eng, ora, x, lam, ineq, fixed, sc, dvec = kkt_case(name, kw, device=0)
    eng.evaluate_resident(x, 1.0, lam)
    rhs = np.random.default_rng(2).normal(size=eng.num_x + len(ineq) + eng.num_c)
    rhs[np.nonzero(fixed)[0]] = 0.0
    out = []
    for cr in ("1", "0"):
        monkeypatch.setenv("PYCOLLO_AMD_KKT_CR", cr)
        k = GpuKkt(eng, ineq, fixed, sc)
        inertia = k.factor(dvec)
        xs = k.solve(rhs)
        xs = xs + k.solve(rhs - k.matvec(dvec, xs))
        out.append((inertia, xs))
        k.close()
    assert out[0][0] == out[1][0]
    assert np.max(np.abs(out[0][1] - out[1][1])) <= 1e-9 * np.max(np.abs(out[1][1]))
    eng.close()


@pytest.mark.parametrize("name,kw", [("brachistochrone", {}), ("hypersensitive", dict(K=40, order=5)),
                                     ("free_flying_robot", dict(K=10, order=5))])
def test_gpu_linear_algebra_reproduces_the_host_solve(built, name, kw):
    """The same interior-point run with the KKT systems on the GPU and on the host: same iterates to solver tolerance."""
    from pycollo_amd.iteration import MeshIteration
    prob = problems.REGISTRY[name](**kw)
    res = {}
    for ls in ("host", "gpu"):
        it = MeshIteration(prob, device=0)
        r = it.solve_with_ipm(max_iter=300, tol=1e-8, linear_solver=ls)
        assert r.success, (ls, r.status)
        res[ls] = (it.objective, r.x.copy(), r.iterations, r.evaluations)
        it.engine.close()
    assert abs(res["gpu"][0] - res["host"][0]) <= 1e-7 * max(1.0, abs(res["host"][0]))
    assert np.max(np.abs(res["gpu"][1] - res["host"][1])) <= 1e-5
    assert "gpu_seconds" in res["gpu"][3]
