"""Stand-ins for the pycollo objects the MI355X backend adapter reads (TEST INFRASTRUCTURE).

pycollo cannot be imported in this image (casadi / pyproprop are absent).  These classes carry exactly the attribute
names of the reference's objects -- nothing of their behaviour -- so that ``pycollo_amd.pycollo_backend`` can be
exercised the way a live pycollo would drive it:

* ``OptimalControlProblem`` / ``Phase`` (user-facing; pycollo/optimal_control_problem.py:113-305, phase.py:303-565):
  ``phases, parameter_variables, objective_function, endpoint_constraints, auxiliary_data, bounds, guess, settings``;
  per phase ``state_variables, control_variables, state_equations, path_constraints, integrand_functions,
  auxiliary_data, initial/final_time_variable, initial/final_state_variables, integral_variables, bounds, guess, mesh``
  with pycollo's generated symbol names (``t0_P0``, ``x_P0(tF)``, ``q0_P0``; phase.py:379-412, 519-520).
* the iteration's ``Mesh`` (mesh.py:110-235): ``p, N, K, N_K, tau``; ``IterationScaling`` (scaling.py:166-169,
  273-281): ``V_ocp, r_ocp, W_ocp, w``; ``Iteration``: ``mesh, scaling, num_x, num_c, guess_x, x_bnd_l/u, c_bnd_l/u``.
* the live backend's counts (backend.py:632-816, 1212-1306): ``p[i].num_each_var, num_y_eqn, num_p_con, num_q_fnc``,
  ``num_s_var, num_b_con``.

The two problems are the reference's unit-test fixtures (tests/unit/conftest.py:14-190), restated as data.
"""
from types import SimpleNamespace

import numpy as np
import sympy as sym


class Phase:
    def __init__(self, ocp, name, number):
        self.name, self.phase_number, self.optimal_control_problem = name, number, ocp
        self._y, self._u, self._q_fnc = (), (), ()
        self.state_equations, self.path_constraints, self.auxiliary_data = (), (), {}
        self.initial_time_variable = sym.Symbol(f"t0_P{number}")
        self.final_time_variable = sym.Symbol(f"tF_P{number}")
        self.initial_state_variables = self.final_state_variables = self.integral_variables = ()
        self.bounds = SimpleNamespace(initial_time=None, final_time=None, state_variables=None, control_variables=None,
                                      integral_variables=None, path_constraints=None, initial_state_constraints=None,
                                      final_state_constraints=None)
        self.guess = SimpleNamespace(time=None, state_variables=None, control_variables=None, integral_variables=None)
        self.mesh = SimpleNamespace(number_mesh_sections=10, mesh_section_sizes=None, number_mesh_section_nodes=4)

    @property
    def state_variables(self):
        return self._y

    @state_variables.setter
    def state_variables(self, ys):
        self._y = tuple(ys)
        n = self.phase_number
        self.initial_state_variables = tuple(sym.Symbol(f"{y}_P{n}(t0)") for y in self._y)
        self.final_state_variables = tuple(sym.Symbol(f"{y}_P{n}(tF)") for y in self._y)

    @property
    def control_variables(self):
        return self._u

    @control_variables.setter
    def control_variables(self, us):
        self._u = tuple(us) if isinstance(us, (list, tuple)) else (us,)

    @property
    def integrand_functions(self):
        return self._q_fnc

    @integrand_functions.setter
    def integrand_functions(self, fs):
        self._q_fnc = tuple(fs)
        self.integral_variables = tuple(sym.Symbol(f"q{i}_P{self.phase_number}") for i in range(len(self._q_fnc)))


class OptimalControlProblem:
    def __init__(self, name):
        self.name, self.phases = name, ()
        self.parameter_variables, self.endpoint_constraints, self.auxiliary_data = (), (), {}
        self.objective_function = None
        self.bounds = SimpleNamespace(parameter_variables=None, endpoint_constraints=None)
        self.guess = SimpleNamespace(parameter_variables=None)
        self.settings = SimpleNamespace(scaling_method="bounds", quadrature_method="lobatto", nlp_tolerance=1e-10,
                                        max_nlp_iterations=2000, linear_solver="mumps", warm_start=False,
                                        mesh_tolerance=1e-7, max_mesh_iterations=10, collocation_points_min=4,
                                        collocation_points_max=10, derivative_level=2)

    def new_phase(self, name):
        ph = Phase(self, name, len(self.phases))
        self.phases = self.phases + (ph,)
        return ph


def brachistochrone():
    """tests/unit/conftest.py:14-76."""
    x, y, v, u = sym.symbols("x y v u")
    problem = OptimalControlProblem(name="Brachistochrone")
    phase = problem.new_phase(name="A")
    phase.state_variables = [x, y, v]
    phase.control_variables = u
    phase.state_equations = [v * sym.sin(u), v * sym.cos(u), 9.81 * sym.cos(u)]
    problem.objective_function = phase.final_time_variable
    phase.bounds.initial_time = 0.0
    phase.bounds.final_time = [0, 10]
    phase.bounds.state_variables = [[0, 10], [0, 10], [-50, 50]]
    phase.bounds.control_variables = [[-np.pi / 2, np.pi / 2]]
    phase.bounds.initial_state_constraints = {x: 0, y: 0, v: 0}
    phase.bounds.final_state_constraints = {x: 2, y: 2}
    phase.guess.time = np.array([0, 10])
    phase.guess.state_variables = np.array([[0, 2], [0, 2], [0, 0]])
    phase.guess.control_variables = np.array([[0, np.pi / 2]])
    problem.settings.max_mesh_iterations = 10
    # what the live backend would have counted (tests/unit/test_iteration.py: 3 states, 1 control, tF free)
    counts = SimpleNamespace(p=[SimpleNamespace(num_each_var=(3, 1, 0, 1), num_y_eqn=3, num_p_con=0, num_q_fnc=0)],
                             num_s_var=0, num_b_con=0)
    return problem, counts


def double_pendulum():
    """tests/unit/conftest.py:79-190."""
    a0, a1, v0, v1, T0, T1 = sym.symbols("a0 a1 v0 v1 T0 T1")
    g = sym.symbols("g")
    m0, p0, d0, l0, k0, I0 = sym.symbols("m0 p0 d0 l0 k0 I0")
    m1, p1, d1, l1, k1, I1 = sym.symbols("m1 p1 d1 l1 k1 I1")
    c0, s0, c1, s1 = sym.symbols("c0 s0 c1 s1")
    M00, M01, M10, M11, K0, K1, detM = sym.symbols("M00 M01 M10 M11 K0 K1 detM")
    K0_eqn = T0 + g * (m0 * p0 + m1 * l0) * c0 + m1 * p1 * l0 * (s1 * c0 - s0 * c1) * v1 ** 2
    K1_eqn = T1 + g * m1 * p1 * c1 + m1 * p1 * l0 * (s0 * c1 - s1 * c0) * v0 ** 2
    problem = OptimalControlProblem(name="Double Pendulum Swing-Up")
    phase = problem.new_phase(name="A")
    phase.state_variables = [a0, a1, v0, v1]
    phase.control_variables = [T0, T1]
    phase.state_equations = [v0, v1, (M11 * K0 - M01 * K1) / detM, (M00 * K1 - M10 * K0) / detM]
    phase.integrand_functions = [(T0**2 + T1**2)]
    phase.auxiliary_data = {g: -9.81, k1: 1 / 12, I0: m0 * (k0 ** 2 + p0 ** 2), I1: m1 * (k1 ** 2 + p1 ** 2),
                            s0: sym.sin(a0), c1: sym.cos(a1)}
    problem.parameter_variables = [m0, p0]
    problem.objective_function = phase.integral_variables[0]
    problem.auxiliary_data = {g: 0, d0: 0.5, k0: 1 / 12, m1: 1.0, p1: 0.5, d1: 0.5, l0: p0 + d0, l1: p1 + d1,
                              I0: m0 * (k0 ** 2 + p0 ** 2), I1: m1 * (k1 ** 2 + p1 ** 2),
                              c0: sym.cos(a0), s0: sym.sin(a0), c1: sym.cos(a1), s1: sym.sin(a1),
                              M00: I0 + m1 * l0 ** 2, M01: m1 * p1 * l0 * (s0 * s1 + c0 * c1), M10: M01, M11: I1,
                              K0: K0_eqn, K1: K1_eqn, detM: M00 * M11 - M01 * M10}
    phase.bounds.initial_time = 0
    phase.bounds.final_time = [1, 3]
    phase.bounds.state_variables = [[-np.pi, np.pi], [-np.pi, np.pi], [-10, 10], [-10, 10]]
    phase.bounds.control_variables = [[-15, 15], [-15, 15]]
    phase.bounds.integral_variables = [0, 1000]
    phase.bounds.initial_state_constraints = [[-0.5 * np.pi, -0.5 * np.pi], [-0.5 * np.pi, -0.5 * np.pi], [0, 0], [0, 0]]
    phase.bounds.final_state_constraints = [[0.5 * np.pi, 0.5 * np.pi], [0.5 * np.pi, 0.5 * np.pi], [0, 0], [0, 0]]
    problem.bounds.parameter_variables = [[0.5, 1.5], [0.5, 1.5]]
    phase.guess.time = [0, 2]
    phase.guess.state_variables = [[-0.5 * np.pi, 0.5 * np.pi], [-0.5 * np.pi, 0.5 * np.pi], [0, 0], [0, 0]]
    phase.guess.control_variables = [[0, 0], [0, 0]]
    phase.guess.integral_variables = [100]
    problem.guess.parameter_variables = [1.0, 1.0]
    # tests/unit/test_iteration.py:192-234: 4 states, 2 controls, 1 integral, tF free, 2 parameters; 121 constraints
    counts = SimpleNamespace(p=[SimpleNamespace(num_each_var=(4, 2, 1, 1), num_y_eqn=4, num_p_con=0, num_q_fnc=1)],
                             num_s_var=2, num_b_con=0)
    return problem, counts


def iteration(problem, V_ocp, r_ocp, W_ocp, w, tau=None, num_x=None, num_c=None):
    """An ``Iteration`` as the backend sees it on the problem's initial mesh: ``mesh`` (phase descriptions + the
    generated N / K / N_K), ``scaling`` (V_ocp, r_ocp, W_ocp, w) and the totals."""
    Ns, Ks, NKs, ps = [], [], [], []
    for ph in problem.phases:
        K = int(ph.mesh.number_mesh_sections)
        nodes = ph.mesh.number_mesh_section_nodes
        nk = np.full(K, int(nodes)) if np.ndim(nodes) == 0 else np.asarray(nodes)
        sizes = ph.mesh.mesh_section_sizes
        ps.append(SimpleNamespace(number_mesh_sections=K, mesh_section_sizes=np.ones(K) / K if sizes is None else sizes,
                                  number_mesh_section_nodes=nk))
        Ns.append(int(nk.sum() - K + 1)); Ks.append(K); NKs.append(nk)
    mesh = SimpleNamespace(p=ps, N=Ns, K=Ks, N_K=NKs)
    if tau is not None:
        mesh.tau = tau
    scaling = SimpleNamespace(V_ocp=np.asarray(V_ocp, float), r_ocp=np.asarray(r_ocp, float), W_ocp=np.asarray(W_ocp, float), w=float(w))
    it = SimpleNamespace(mesh=mesh, scaling=scaling)
    if num_x is not None:
        it.num_x = num_x
    if num_c is not None:
        it.num_c = num_c
    return it
