"""The pycollo-side adapter (pycollo_amd/pycollo_backend.py) driven the way a live pycollo drives its backend, on
stand-in objects with the reference's attribute names (tests/pycollo_stub.py): problem lowering, counts and slices
against tests/unit/test_iteration.py:192-234, CCS order of the Jacobian probes (backend.py:1738-1761), and -- on the
GPU -- the reference's known answers through ``Mi355x.evaluate_*`` and an end-to-end ``solve_nlp``."""
import numpy as np
import pytest

import pycollo_stub as stub
from conftest import golden_tables
from pycollo_amd import problems
from pycollo_amd.model import compile_model
from pycollo_amd.pycollo_backend import Mi355x, phase_mesh, to_problem_spec
from pycollo_amd.quadrature import QuadratureTables


def _ocp_scaling(expanded, lay):
    """Per-OCP-variable vector from the reference's per-NLP-variable EXPECT_V / EXPECT_R arrays."""
    out = []
    for pl in lay.phases:
        out += [expanded[pl.x_off + b * pl.N] for b in range(pl.n_z)]
        out += list(expanded[pl.q_off:pl.q_off + pl.n_q + pl.n_t])
    out += list(expanded[lay.s_off:])
    return np.array(out)


def _backend(make, known_answers, tag, device):
    ocp, counts = make()
    be = Mi355x(ocp, device=device)
    be.p, be.num_s_var, be.num_b_con = counts.p, counts.num_s_var, counts.num_b_con   # what the live BackendABC holds
    spec = to_problem_spec(ocp)
    from pycollo_amd.layout import NlpLayout
    from pycollo_amd.mesh import build_phase_mesh
    lay = NlpLayout(compile_model(spec), [build_phase_mesh(QuadratureTables("lobatto"), *ph.mesh.resolved()) for ph in spec.phases])
    V = _ocp_scaling(known_answers[f"EXPECT_V_{tag}"], lay)
    r = _ocp_scaling(known_answers[f"EXPECT_R_{tag}"], lay)
    it = stub.iteration(ocp, V, r, np.ones(lay.num_ocp_c), 1.0, num_x=len(known_answers[f"EXPECT_V_{tag}"]))
    return be, it, lay


@pytest.mark.parametrize("make,ours", [(stub.brachistochrone, problems.brachistochrone), (stub.double_pendulum, problems.double_pendulum)])
def test_lowered_problem_is_the_registered_one(make, ours):
    """to_problem_spec(reference-shaped OCP) compiles to the same model (same digest: same equations after auxiliary
    substitution, same bounds / needed masks) as the hand-written problem definition the parity tests use."""
    ocp, counts = make()
    spec = to_problem_spec(ocp)
    m1, m2 = compile_model(spec), compile_model(ours())
    assert m1.digest == m2.digest
    pm = m1.phases[0]
    assert (pm.n_y, pm.n_u, pm.n_q, int(pm.t_free[0]) + int(pm.t_free[1])) == counts.p[0].num_each_var
    assert m1.n_s == counts.num_s_var
    # pycollo's generated symbols are gone from every expression
    gone = {ocp.phases[0].initial_time_variable, ocp.phases[0].final_time_variable, *ocp.phases[0].initial_state_variables,
            *ocp.phases[0].final_state_variables, *ocp.phases[0].integral_variables}
    assert not (spec.objective_function.free_symbols & gone)


def test_phase_mesh_checks_the_iterations_tables():
    ocp, _ = stub.brachistochrone()
    quad = QuadratureTables("lobatto")
    it = stub.iteration(ocp, np.ones(5), np.zeros(5), np.ones(3), 1.0)
    m = phase_mesh(it.mesh, 0, quad)
    assert (m.N, m.K) == (31, 10)
    it.mesh.N = [32]
    with pytest.raises(ValueError):
        phase_mesh(it.mesh, 0, quad)
    it.mesh.N = [31]
    it.mesh.tau = [np.linspace(-1, 1, 31)]      # not the Lobatto nodes
    with pytest.raises(ValueError):
        phase_mesh(it.mesh, 0, quad)
    it.mesh.tau = [golden_mesh_tau(quad)]
    assert phase_mesh(it.mesh, 0, quad).N == 31


def golden_mesh_tau(quad):
    from pycollo_amd.mesh import build_phase_mesh
    return build_phase_mesh(quad, np.ones(10) / 10, np.full(10, 4)).tau


def test_counts_slices_and_ccs_order_structure_only(built, known_answers):
    """tests/unit/test_iteration.py:192-234 (double pendulum: 190 variables, 121 constraints, slices) through the
    adapter on a structure-only engine; Jacobian probes in CasADi's column-major order (backend.py:1747-1761)."""
    be, it, lay = _backend(stub.double_pendulum, known_answers, "DP", None)
    it.num_c = 121
    be.generate_nlp_function_callables(it)
    e = be.engine
    assert (e.num_x, e.num_c) == (190, 121)
    pl = e.layout.phases[0]
    assert (pl.x_off, pl.x_off + pl.n_y * pl.N, pl.q_off, pl.t_off, e.layout.s_off) == (0, 124, 186, 187, 188)
    assert (pl.c_off, pl.c_path_off, pl.c_int_off, e.layout.c_end_off) == (0, 120, 120, 121)
    r, c = be.evaluate_G_structure()
    assert len(r) == be.evaluate_G_num_nonzero()
    key = c.astype(np.int64) * e.num_c + r
    assert np.all(np.diff(key) > 0)              # columns ascending, rows ascending inside a column: CCS
    # a live backend whose counts differ must be refused, not silently mis-mapped
    be.p[0].num_each_var = (4, 2, 1, 2)
    with pytest.raises(RuntimeError):
        be.generate_nlp_function_callables(it)


@pytest.mark.gpu
def test_known_answers_through_the_backend(built, known_answers):
    """EXPECT_* of tests/unit/test_iteration.py:290-354, 371-385 via Mi355x.evaluate_J / g / c after
    generate_nlp_function_callables + create_nlp_solver, i.e. along the call path of OptimalControlProblem.solve()."""
    be, it, _ = _backend(stub.brachistochrone, known_answers, "BR", 0)
    be.generate_nlp_function_callables(it)
    be.create_nlp_solver()
    x = known_answers["EXPECT_X_TILDE_BR"]
    np.testing.assert_almost_equal(be.evaluate_J(x), 0.8243386694458454)
    g = np.zeros(125); g[124] = 10
    np.testing.assert_allclose(be.evaluate_g(x), g)
    c = be.evaluate_c(x)
    np.testing.assert_allclose(c, np.zeros(90), atol=10e-2)
    assert np.max(np.abs(c)) < 1e-8
    # the probes agree with each other: COO matrix == structure + CCS-ordered values
    G = be.evaluate_G(x).tocsr()
    r, cc = be.evaluate_G_structure()
    np.testing.assert_array_equal(np.asarray(G[r, cc]).ravel(), be.evaluate_G_nonzeros(x))
    # scaling-time callables (scaling.py:360-363, 392-395): gradient with w_J = 3, dense G with W = 2
    np.testing.assert_allclose(be.g_iter_scale_callable(np.concatenate([x, [3.0]])), 3 * g)
    Gd = be.G_iter_scale_callable(np.concatenate([x, 2 * np.ones(3)]))
    np.testing.assert_allclose(Gd, 2 * G.toarray())
    np.testing.assert_allclose(be.constraint_row_norms(x), np.sqrt((G.toarray() ** 2).sum(axis=1)), rtol=1e-12)
    # state derivatives at the nodes vs the oracle's f
    from oracle.ref_numpy import OracleNlp
    ora = OracleNlp(problems.brachistochrone(), golden_tables("lobatto"))
    z, q, stretch, _, w = ora._unpack(ora.P[0], x)
    ref = np.concatenate([np.broadcast_to(fn(*ora._args(ora.P[0], z, w)), (31,)) for fn in ora.P[0].F_fn[:3]])
    np.testing.assert_allclose(be.dy_iter_callable(x), ref, rtol=1e-12)

    be2, it2, _ = _backend(stub.double_pendulum, known_answers, "DP", 0)
    be2.generate_nlp_function_callables(it2)
    be2.create_nlp_solver()
    xd = known_answers["EXPECT_X_TILDE_DP"]
    assert be2.evaluate_J(xd) == 100
    gd = np.zeros(190); gd[186] = 1000
    np.testing.assert_allclose(be2.evaluate_g(xd), gd)
    be.engine.close(); be2.engine.close()


@pytest.mark.gpu
def test_solve_nlp_through_the_backend(built, known_answers):
    """generate -> create_nlp_solver -> solve_nlp with the iteration's guess and bounds (backend.py:1807-1827): the
    brachistochrone NLP on its K = 10, n = 4 mesh converges to the reference's pinned objective."""
    from pycollo_amd.iteration import MeshIteration
    be, it, _ = _backend(stub.brachistochrone, known_answers, "BR", 0)
    mi = MeshIteration(problems.brachistochrone())         # the build's own setup: scaled guess and bounds
    it.guess_x, it.x_bnd_l, it.x_bnd_u, it.c_bnd_l, it.c_bnd_u = mi.guess_x_tilde, mi.x_bnd_l, mi.x_bnd_u, mi.c_bnd_l, mi.c_bnd_u
    it.scaling.W_ocp, it.scaling.w = mi.W_ocp, mi.w
    be.generate_nlp_function_callables(it)
    be.create_nlp_solver()
    res = be.solve_nlp()
    assert res.solution["status"] == 0, res.solution["status_msg"]
    J = be.evaluate_J(res.solution["x"]) / it.scaling.w
    np.testing.assert_allclose(J, 0.8243386694458454, rtol=1e-6)
    np.testing.assert_allclose(res.solution["x"], known_answers["EXPECT_X_TILDE_BR"], atol=2e-3)
    be.engine.close()
