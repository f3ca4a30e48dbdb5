"""Row N2: ph mesh-error estimate on the GPU against the NumPy restatement of
pycollo/mesh_refinement.py:63-240, and the refine -> re-solve loop."""
import numpy as np
import pytest

from conftest import golden_tables
from oracle.ref_numpy import OracleNlp
from oracle.ref_refine import mesh_error as oracle_mesh_error
from pycollo_amd import problems
from pycollo_amd.quadrature import QuadratureTables

pytestmark = pytest.mark.gpu


def _ragged(prob, seed=3, K=23):
    rng = np.random.default_rng(seed)
    ph = prob.phases[0]
    ph.mesh.number_mesh_sections = K
    ph.mesh.mesh_section_sizes = rng.uniform(0.3, 1.0, K)
    ph.mesh.number_mesh_section_nodes = rng.integers(3, 9, K)
    return prob


@pytest.mark.parametrize("name,kw,ragged", [("hypersensitive", dict(K=40, order=5), False),
                                            ("cart_pole", dict(K=300, order=4), False),
                                            ("shuttle", dict(K=12, order=6), False),
                                            ("two_phase_transfer", dict(K=6, order=4), False),
                                            ("time_coupled_transfer", dict(K=6, order=4), False),
                                            ("cart_pole", dict(K=10, order=4), True),
                                            ("double_pendulum", dict(K=10, order=4), True)])
def test_mesh_error_matches_oracle(built, name, kw, ragged):
    from pycollo_amd.engine import NlpEngine
    from pycollo_amd.refinement import mesh_error
    prob = problems.REGISTRY[name](**kw)
    if ragged:
        prob = _ragged(prob)
    eng = NlpEngine(prob, device=0)
    ora = OracleNlp(prob, golden_tables("lobatto"), V_ocp=eng.V_ocp, r_ocp=eng.r_ocp, W_ocp=eng.W_ocp)
    # a smooth "solution": low-order polynomial in tau per variable, so the estimate is small but not zero
    rng = np.random.default_rng(5)
    x = np.zeros(eng.num_x)
    for pl, mesh in zip(eng.layout.phases, eng.meshes):
        for b in range(pl.n_z):
            cf = rng.uniform(-0.15, 0.15, 4)
            x[pl.x_off + b * pl.N:pl.x_off + (b + 1) * pl.N] = np.polynomial.polynomial.polyval(mesh.tau, cf)
        x[pl.q_off:pl.q_off + pl.n_q + pl.n_t] = rng.uniform(0.1, 0.3, pl.n_q + pl.n_t)
    x[eng.layout.s_off:] = rng.uniform(-0.2, 0.2, eng.layout.n_s)
    got = mesh_error(eng, x)
    ref = oracle_mesh_error(ora, x)
    for (rel, ab), (ref_abs, ref_rel), pm in zip(got, ref, eng.model.phases):
        scale = max(1.0, np.max(np.abs(ref_abs)))
        # the estimate is a difference of O(|y|) numbers: absolute tolerance relative to the state magnitude
        y_mag = 1.0 + np.max(np.abs(eng.V_ocp))
        np.testing.assert_allclose(ab, ref_abs.max(axis=2), rtol=1e-8, atol=1e-11 * y_mag * scale)
        np.testing.assert_allclose(rel, ref_rel, rtol=1e-8, atol=1e-11 * scale)
    eng.close()


def test_refine_and_resolve_brachistochrone(built):
    """solve -> estimate -> refine -> carry the solution over: a coarse mesh (K = 5) converges to the known
    objective but misses the 1e-7 mesh tolerance, the rules ask for more nodes, the reference's default mesh
    (K = 10) meets the tolerance, and the carried-over solution is a far better starting point than the user
    guess.  (scipy's trust-constr stands in for IPOPT; it is only asked to solve meshes it is known to handle.)"""
    from pycollo_amd.iteration import MeshIteration
    from pycollo_amd.refinement import mesh_error, next_phase_mesh
    it5 = MeshIteration(problems.brachistochrone(K=5, order=4))
    res5 = it5.solve_with_scipy(maxiter=600)
    assert res5.constr_violation < 1e-8
    np.testing.assert_allclose(it5.objective, 0.82434, rtol=1e-4)     # tests/integration/test_brachistochrone.py:159-166
    (rel5, _), = mesh_error(it5.engine, it5.x_tilde)
    assert rel5.shape == (5,) and np.all(rel5 >= 0) and np.max(rel5) > 1e-7
    sizes, nodes, done = next_phase_mesh(it5.meshes[0].sizes, it5.meshes[0].n, rel5)
    assert not done and abs(sizes.sum() - 1) < 1e-12 and nodes.min() >= 4 and nodes.max() <= 10
    assert nodes.sum() > it5.meshes[0].n.sum()
    it10 = MeshIteration(problems.brachistochrone())
    it10.solve_with_scipy(maxiter=600)
    (rel10, _), = mesh_error(it10.engine, it10.x_tilde)
    assert np.max(rel10) < 1e-7 < np.max(rel5)
    assert next_phase_mesh(it10.meshes[0].sizes, it10.meshes[0].n, rel10)[2]
    np.testing.assert_allclose(it10.objective, 0.82434, rtol=1e-4)
    # carry the K = 5 solution to the refined mesh (iteration.py:528-583 -> 86-194)
    x = it5.V * it5.x_tilde + it5.r
    pl = it5.layout.phases[0]
    prev = ([it5.meshes[0].tau], [x[pl.x_off:pl.x_off + 3 * pl.N].reshape(3, -1)], [x[pl.x_off + 3 * pl.N:pl.q_off].reshape(1, -1)],
            [np.zeros(0)], [x[pl.t_off:pl.t_off + 1]], np.zeros(0))
    prob2 = problems.brachistochrone()
    prob2.phases[0].mesh.number_mesh_sections = len(nodes)
    prob2.phases[0].mesh.mesh_section_sizes = sizes
    prob2.phases[0].mesh.number_mesh_section_nodes = nodes
    warm = MeshIteration(prob2, prev=prev)
    cold = MeshIteration(prob2)
    viol_warm = np.max(np.abs(warm.engine.evaluate_c(warm.guess_x_tilde)))
    viol_cold = np.max(np.abs(cold.engine.evaluate_c(cold.guess_x_tilde)))
    assert viol_warm < 1e-2 * viol_cold
    assert abs(warm.engine.evaluate_J(warm.guess_x_tilde) - 0.82434) < 1e-4
