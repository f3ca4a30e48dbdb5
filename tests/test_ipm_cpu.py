"""The interior-point stand-in for IPOPT (pycollo_amd/ipm.py) on small analytic NLPs, through the cyipopt
``problem_obj`` protocol it shares with PycolloGpuProblem.  No GPU: plain NumPy problem objects."""
import numpy as np

from pycollo_amd.ipm import InteriorPointSolver, solve_nlp


class HS071:
    """Hock-Schittkowski 71 (IPOPT's own tutorial problem): f* = 17.0140173."""
    n, m = 4, 2

    def objective(self, x):
        return x[0] * x[3] * (x[0] + x[1] + x[2]) + x[2]

    def gradient(self, x):
        return np.array([x[3] * (2 * x[0] + x[1] + x[2]), x[0] * x[3], x[0] * x[3] + 1.0, x[0] * (x[0] + x[1] + x[2])])

    def constraints(self, x):
        return np.array([np.prod(x), np.dot(x, x)])

    def jacobianstructure(self):
        return np.repeat([0, 1], 4), np.tile(np.arange(4), 2)

    def jacobian(self, x):
        return np.concatenate([np.prod(x) / x, 2 * x])

    def hessianstructure(self):
        return np.tril_indices(4)

    def hessian(self, x, lam, s):
        H = s * np.array([[2 * x[3], 0, 0, 0], [x[3], 0, 0, 0], [x[3], 0, 0, 0], [2 * x[0] + x[1] + x[2], x[0], x[0], 0]], float)
        H += lam[0] * np.array([[0, 0, 0, 0], [x[2] * x[3], 0, 0, 0], [x[1] * x[3], x[0] * x[3], 0, 0], [x[1] * x[2], x[0] * x[2], x[0] * x[1], 0]], float)
        H += lam[1] * 2 * np.eye(4)
        return H[np.tril_indices(4)]


def test_hs071():
    p = HS071()
    res = solve_nlp(p, np.array([1.0, 5.0, 5.0, 1.0]), np.ones(4), 5 * np.ones(4), np.array([25.0, 40.0]), np.array([2e19, 40.0]))
    assert res.success, res.status
    np.testing.assert_allclose(res.objective, 17.0140173, rtol=1e-7)
    np.testing.assert_allclose(res.x, [1.0, 4.74299963, 3.82114998, 1.37940829], rtol=1e-6)
    assert res.inf_pr < 1e-8


def test_warm_start_keeps_the_starting_point_closer():
    """pycollo's ``warm_start`` setting = IPOPT's warm_start_init_point (backend.py:1703-1709): the bound push of the
    initial point drops from 1e-2 to 1e-3; same optimum, and a start on a bound moves a tenth as far."""
    from pycollo_amd.ipm import InteriorPointSolver
    p = HS071()
    args = (p, 4, 2, np.ones(4), 5 * np.ones(4), np.array([25.0, 40.0]), np.array([2e19, 40.0]))
    x0 = np.array([1.0, 4.743, 3.821, 1.379])
    cold, warm = InteriorPointSolver(*args), InteriorPointSolver(*args, warm_start=True)
    v = np.concatenate([x0, [30.0]])
    assert abs(cold._push_interior(v)[0] - 1.0 - 1e-2) < 1e-12
    assert abs(warm._push_interior(v, 1e-3, 1e-3)[0] - 1.0 - 1e-3) < 1e-12
    rc, rw = cold.solve(x0), warm.solve(x0)
    assert rc.success and rw.success
    np.testing.assert_allclose(rw.objective, 17.0140173, rtol=1e-7)
    np.testing.assert_allclose(rw.x, rc.x, rtol=1e-6)


class Rosenbrock:
    """Unconstrained in c (m = 0) with one fixed variable and one active bound."""
    n, m = 3, 0

    def objective(self, x):
        return 100 * (x[1] - x[0] ** 2) ** 2 + (1 - x[0]) ** 2 + (x[2] - 3.0) ** 2

    def gradient(self, x):
        return np.array([-400 * x[0] * (x[1] - x[0] ** 2) - 2 * (1 - x[0]), 200 * (x[1] - x[0] ** 2), 2 * (x[2] - 3.0)])

    def constraints(self, x):
        return np.zeros(0)

    def jacobianstructure(self):
        return np.zeros(0, int), np.zeros(0, int)

    def jacobian(self, x):
        return np.zeros(0)

    def hessianstructure(self):
        return np.array([0, 1, 1, 2]), np.array([0, 0, 1, 2])

    def hessian(self, x, lam, s):
        return s * np.array([1200 * x[0] ** 2 - 400 * x[1] + 2, -400 * x[0], 200.0, 2.0])


def test_bounds_and_fixed_variables():
    p = Rosenbrock()
    s = InteriorPointSolver(p, 3, 0, [-2, -2, 1.0], [0.5, 2, 1.0], [], [])
    res = s.solve(np.array([-1.0, 1.0, 1.0]))
    assert res.success, res.status
    assert res.x[2] == 1.0                                  # fixed variable untouched
    np.testing.assert_allclose(res.x[:2], [0.5, 0.25], atol=1e-6)   # x0 at its upper bound
    np.testing.assert_allclose(res.objective, 0.25 + 4.0, atol=1e-6)


def test_cyipopt_shaped_interface_with_user_scaling():
    """The four calls the reference makes on ``ipopt.problem`` (pycollo/nlp.py:84-115) against the adapter: options,
    user scaling, solve -> (x, info).  Scaling must not move the solution, and the multipliers it reports are those of
    the unscaled problem (stationarity checked with the unscaled derivatives)."""
    from pycollo_amd import ipopt_api as ipopt
    p = HS071()
    x0 = np.array([1.0, 5.0, 5.0, 1.0])
    out = []
    for scaled in (False, True):
        nlp = ipopt.problem(n=4, m=2, problem_obj=p, lb=np.ones(4), ub=5 * np.ones(4), cl=np.array([25.0, 40.0]), cu=np.array([2e19, 40.0]))
        nlp.addOption('mu_strategy', 'adaptive')
        nlp.addOption('tol', 1e-9)
        nlp.addOption('max_iter', 500)
        nlp.addOption('print_level', 0)
        if scaled:
            nlp.addOption('nlp_scaling_method', 'user-scaling')
            nlp.setProblemScaling(0.05, np.array([2.0, 0.5, 1.0, 4.0]), np.array([0.1, 3.0]))
        x, info = nlp.solve(x0)
        assert info["status"] == 0, info["status_msg"]
        np.testing.assert_allclose(info["obj_val"], 17.0140173, rtol=1e-7)
        np.testing.assert_allclose(x, [1.0, 4.74299963, 3.82114998, 1.37940829], rtol=1e-6)
        np.testing.assert_allclose(info["g"], p.constraints(x))
        jr, jc = p.jacobianstructure()
        J = np.zeros((2, 4)); J[jr, jc] = p.jacobian(x)
        stat = p.gradient(x) + J.T @ info["mult_g"] - info["mult_x_L"] + info["mult_x_U"]
        assert np.max(np.abs(stat)) < 1e-6
        out.append(info)
    np.testing.assert_allclose(out[0]["mult_g"], out[1]["mult_g"], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(out[0]["mult_x_L"], out[1]["mult_x_L"], rtol=1e-4, atol=1e-7)


class Maratos:
    """min 2 (x^2 + y^2 - 1) - x  s.t.  x^2 + y^2 = 1 (Powell's example of the Maratos effect): from a feasible point the
    full Newton step increases both the objective and the violation, whatever the step's quality."""
    n, m = 2, 1

    def objective(self, x):
        return 2.0 * (x[0]**2 + x[1]**2 - 1.0) - x[0]

    def gradient(self, x):
        return np.array([4.0 * x[0] - 1.0, 4.0 * x[1]])

    def constraints(self, x):
        return np.array([x[0]**2 + x[1]**2 - 1.0])

    def jacobianstructure(self):
        return np.array([0, 0]), np.array([0, 1])

    def jacobian(self, x):
        return np.array([2.0 * x[0], 2.0 * x[1]])

    def hessianstructure(self):
        return np.array([0, 1]), np.array([0, 1])

    def hessian(self, x, lam, s):
        return np.array([4.0 * s + 2.0 * lam[0], 4.0 * s + 2.0 * lam[0]])


def test_second_order_correction_takes_the_full_step():
    """IPOPT's second-order correction (A-5.7 .. A-5.9), which the stand-in applies like the solver behind
    pycollo/backend.py:1711: on the Maratos example the corrected full steps are accepted where the plain line search
    backtracks; same optimum either way, fewer iterations with it."""
    p = Maratos()
    t = 1.2
    x0 = np.array([np.cos(t), np.sin(t)])
    args = (p, 2, 1, np.full(2, -2e19), np.full(2, 2e19), np.zeros(1), np.zeros(1))
    with_soc = InteriorPointSolver(*args, tol=1e-10)
    without = InteriorPointSolver(*args, tol=1e-10, second_order_correction=False)
    a, b = with_soc.solve(x0), without.solve(x0)
    assert a.success and b.success
    np.testing.assert_allclose(a.x, [1.0, 0.0], atol=1e-7)
    np.testing.assert_allclose(b.x, [1.0, 0.0], atol=1e-7)
    assert with_soc.counts.get("second_order_steps", 0) >= 1
    assert "second_order_steps" not in without.counts
    assert a.iterations <= b.iterations
