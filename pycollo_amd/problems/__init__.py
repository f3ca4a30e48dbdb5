"""Benchmark / fixture problem definitions (equations, constants and bounds only).

Each function returns a fresh :class:`pycollo_amd.problem.ProblemSpec`.  Dynamics and bounds are
restated from the reference's example scripts and unit-test fixtures (cited per function); none of
the reference's front-end machinery is involved.  ``K``/``order`` set a uniform initial mesh.
"""
from __future__ import annotations

import numpy as np
import sympy as sym

from ..problem import ProblemSpec


def _mesh(ph, K, order):
    ph.mesh.number_mesh_sections = K
    ph.mesh.number_mesh_section_nodes = order


def brachistochrone(K: int = 10, order: int = 4) -> ProblemSpec:
    """Betts ex. 4.10; examples/brachistochrone/brachistochrone.py:14-50,
    tests/unit/conftest.py:14-76.  3 states, 1 control, tF free, objective tF."""
    x, y, v, u = sym.symbols("x y v u")
    prob = ProblemSpec("Brachistochrone")
    ph = prob.new_phase("A")
    ph.state_variables = [x, y, v]
    ph.control_variables = [u]
    grav = 9.81
    ph.state_equations = [v * sym.sin(u), v * sym.cos(u), grav * sym.cos(u)]
    prob.objective_function = ph.final_time_variable
    ph.bounds.initial_time = 0.0
    ph.bounds.final_time = [0, 10]
    ph.bounds.state_variables = [[0, 10], [0, 10], [-50, 50]]
    ph.bounds.control_variables = [[-np.pi / 2, np.pi / 2]]
    ph.bounds.initial_state_constraints = {x: 0, y: 0, v: 0}
    ph.bounds.final_state_constraints = {x: 2, y: 2}
    ph.guess.time = np.array([0.0, 10.0])
    ph.guess.state_variables = np.array([[0.0, 2.0], [0.0, 2.0], [0.0, 0.0]])
    ph.guess.control_variables = np.array([[0.0, np.pi / 2]])
    _mesh(ph, K, order)
    return prob


def hypersensitive(K: int = 10, order: int = 4, *, test_fixture_bounds: bool = False) -> ProblemSpec:
    """Betts ex. 4.4; examples/hypersensitive_problem/hypersensitive_problem.py:14-35.
    ``test_fixture_bounds`` switches to the bounds of tests/unit/conftest.py:193-230."""
    y, u = sym.symbols("y u")
    prob = ProblemSpec("Hypersensitive problem")
    ph = prob.new_phase("A")
    ph.state_variables = [y]
    ph.control_variables = [u]
    ph.state_equations = [-y**3 + u]
    ph.integrand_functions = [0.5 * (y**2 + u**2)]
    prob.objective_function = ph.integral_variables[0]
    ph.bounds.initial_time = 0.0
    ph.bounds.final_time = 10000.0
    if test_fixture_bounds:
        ph.bounds.state_variables = [[0, 2]]
        ph.bounds.control_variables = [[-1, 8]]
        ph.bounds.integral_variables = [[0, 2000]]
    else:
        ph.bounds.state_variables = [[-50, 50]]
        ph.bounds.control_variables = [[-50, 50]]
        ph.bounds.integral_variables = [[0, 100000]]
    ph.bounds.initial_state_constraints = [[1.0, 1.0]]
    ph.bounds.final_state_constraints = [[1.5, 1.5]]
    ph.guess.time = np.array([0.0, 10000.0])
    ph.guess.state_variables = np.array([[1.0, 1.5]])
    ph.guess.control_variables = np.array([[0.0, 0.0]])
    ph.guess.integral_variables = np.array([4.0])
    _mesh(ph, K, order)
    return prob


def cart_pole(K: int = 10, order: int = 4) -> ProblemSpec:
    """Kelly (2017) cart-pole swing-up; examples/cart_pole_swing_up/cart_pole_swing_up_explicit.py:22-77."""
    q1, q2, q1d, q2d, q1dd, q2dd, F = sym.symbols("q1 q2 q1d q2d q1dd q2dd F")
    m1, m2, l, g = sym.symbols("m1 m2 l g")
    prob = ProblemSpec("Cart-Pole Swing-Up")
    ph = prob.new_phase("A")
    ph.state_variables = [q1, q2, q1d, q2d]
    ph.control_variables = [F]
    ph.state_equations = [q1d, q2d, q1dd, q2dd]
    ph.integrand_functions = [F**2]
    ph.bounds.initial_time = 0
    ph.bounds.final_time = 2.0
    ph.bounds.state_variables = {q1: [-2.0, 2.0], q2: [-10, 10], q1d: [-10, 10], q2d: [-10, 10]}
    ph.bounds.control_variables = {F: [-20.0, 20.0]}
    ph.bounds.integral_variables = [[0, 100]]
    ph.bounds.initial_state_constraints = {q1: 0, q2: 0, q1d: 0, q2d: 0}
    ph.bounds.final_state_constraints = {q1: 1.0, q2: np.pi, q1d: 0, q2d: 0}
    ph.guess.time = np.array([0.0, 2.0])
    ph.guess.state_variables = np.array([[0.0, 1.0], [0.0, np.pi], [0.0, 0.0], [0.0, 0.0]])
    ph.guess.control_variables = np.array([[0.0, 0.0]])
    ph.guess.integral_variables = np.array([0.0])
    s2, c2 = sym.sin(q2), sym.cos(q2)
    prob.objective_function = ph.integral_variables[0]
    prob.auxiliary_data = {
        g: 9.81, l: 0.5, m1: 1.0, m2: 0.3,
        q1dd: (l * m2 * s2 * q2d**2 + F + m2 * g * c2 * s2) / (m1 + m2 * (1 - c2**2)),
        q2dd: -(l * m2 * c2 * s2 * q2d**2 + F * c2 + (m1 + m2) * g * s2) / (l * m1 + l * m2 * (1 - c2**2)),
    }
    _mesh(ph, K, order)
    return prob


def shuttle(K: int = 10, order: int = 4) -> ProblemSpec:
    """Betts ex. 6.1 maximum crossrange;
    examples/space_shuttle_reentry_trajectory/space_shuttle_reentry_trajectory_maximum_crossrange.py:117-200.
    6 states, 2 controls, tF free, objective -theta(tF)."""
    h, phi, theta, nu, gamma, psi, alpha, beta = sym.symbols("h phi theta nu gamma psi alpha beta")
    D, L, g, r, rho, rho_0, h_r, c_L, c_D = sym.symbols("D L g r rho rho_0 h_r c_L c_D")
    Re, S, cl0, cl1, mu, cd0, cd1, cd2, m = sym.symbols("Re S c_lift_0 c_lift_1 mu c_drag_0 c_drag_1 c_drag_2 m")
    prob = ProblemSpec("Space shuttle reentry trajectory maximum crossrange")
    ph = prob.new_phase("A")
    ph.state_variables = [h, phi, theta, nu, gamma, psi]
    ph.control_variables = [alpha, beta]
    cg, sg = sym.cos(gamma), sym.sin(gamma)
    ph.state_equations = {
        h: nu * sg,
        phi: nu * cg * sym.sin(psi) / (r * sym.cos(theta)),
        theta: nu * cg * sym.cos(psi) / r,
        nu: -(D / m) - g * sg,
        gamma: L * sym.cos(beta) / (m * nu) + cg * ((nu / r) - (g / nu)),
        psi: L * sym.sin(beta) / (m * nu * cg) + nu * cg * sym.sin(psi) * sym.sin(theta) / (r * sym.cos(theta)),
    }
    prob.objective_function = -ph.final_state_variables[2]
    prob.auxiliary_data = {
        rho_0: 1.225570827014494, h_r: 7254.24, Re: 6371203.92, S: 249.9091776,
        cl0: -0.2070, cl1: 1.6756, mu: 3.986031954093051e14,
        cd0: 0.07854, cd1: -0.3529, cd2: 2.0400, m: 92079.2525560557,
        D: 0.5 * c_D * S * rho * nu**2,
        L: 0.5 * c_L * S * rho * nu**2,
        g: mu / (r**2),
        r: Re + h,
        rho: rho_0 * sym.exp(-h / h_r),
        c_L: cl0 + cl1 * alpha,
        c_D: cd0 + cd1 * alpha + cd2 * alpha**2,
    }
    deg = np.pi / 180
    ph.bounds.initial_time = [0.0, 0.0]
    ph.bounds.final_time = [0.0, 3000.0]
    ph.bounds.state_variables = {h: [0, 300000], phi: [-np.pi, np.pi], theta: [-70 * deg, 70 * deg],
                                 nu: [10, 45000], gamma: [-80 * deg, 80 * deg], psi: [-np.pi, np.pi]}
    ph.bounds.control_variables = {alpha: [-np.pi / 2, np.pi / 2], beta: [-np.pi / 2, deg]}
    ph.bounds.initial_state_constraints = {h: 79248, phi: 0, theta: 0, nu: 7802.88, gamma: -1 * deg, psi: 90 * deg}
    ph.bounds.final_state_constraints = {h: [24384, 24384], nu: [762, 762], gamma: [-5 * deg, -5 * deg]}
    ph.guess.time = np.array([0.0, 1000.0])
    ph.guess.state_variables = np.array([[79248, 24384], [0, 10 * deg], [0, 10 * deg], [7802.88, 762],
                                         [-1 * deg, -5 * deg], [90 * deg, -90 * deg]], dtype=float)
    ph.guess.control_variables = np.array([[0.0, 0.0], [0.0, 0.0]])
    _mesh(ph, K, order)
    return prob


def delta_iii(K: int = 10, order: int = 4, *, burnout_mass: bool = False) -> ProblemSpec:
    """Betts ex. 6.15 Delta III ascent, 4 phases, 18 linkage constraints;
    examples/delta_iii_launch_vehicle/delta_iii_launch_vehicle.py:198-459.
    7 states, 3 controls, 2 path constraints per phase; all phase times fixed.

    As the example stands the NLP is **infeasible**: phase D's mass is pinned to 23 464 kg at 261 s and to the payload's
    4 164 kg at 961 s, while its state equation burns T / (g_0 I) = 24.03 kg/s for those 700 s -- 16 820 kg, the second
    stage's propellant, not the 19 300 kg between the two pins (the difference is the stage's 2 480 kg of structure,
    which Betts' problem drops only because there the burn ends at a *free* final time).  Every mass defect row of phase
    D keeps the same residual whatever the solver does (2 480 kg; phases A-C are off by about a gram, of either sign: the
    rounding of the tabulated propellant masses against thrust / specific impulse).  ``burnout_mass=True`` is the minimal repair: the
    final mass of a phase is what its burn leaves -- bounded by the example's figure (less 1 kg) from below instead of
    being pinned to it; same functions, same code object."""
    r_x, r_y, r_z, v_x, v_y, v_z, m = sym.symbols("r_x r_y r_z v_x v_y v_z m")
    u_x, u_y, u_z = sym.symbols("u_x u_y u_z")
    D_x, D_y, D_z, T, xi, C_D, S, omega_E = sym.symbols("D_x D_y D_z T xi C_D S omega_E")
    v_r_x, v_r_y, v_r_z, oxr_x, oxr_y, oxr_z = sym.symbols("v_r_x v_r_y v_r_z omega_x_r_x omega_x_r_y omega_x_r_z")
    mu, R_E, psi_L, g_0, h_0, h, rho, rho_0 = sym.symbols("mu R_E psi_L g_0 h_0 h rho rho_0")
    r_n, u_n, v_n, v_r_n, T_over_m = sym.symbols("r_vec_norm u_vec_norm v_vec_norm v_r_vec_norm T_over_m")

    t_launch, t_sep_S, t_sep_1, t_sep_2, t_orbit = 0.0, 75.2, 150.4, 261.0, 961.0
    m_tot_S, m_tot_1, m_tot_2 = 19290, 104380, 19300
    m_prop_S, m_prop_1 = 17010, 95550
    m_struct_S, m_struct_1 = 2280, 8830
    T_eng_S, T_eng_1, T_eng_2 = 628500, 1083100, 110094
    I_S, I_1, I_2 = 283.33364, 301.68776, 467.21311
    tau_S, tau_1 = 75.2, 261
    m_payload = 4164
    m0_A = 9 * m_tot_S + m_tot_1 + m_tot_2 + m_payload
    mF_A = m0_A - 6 * m_prop_S - (tau_S / tau_1) * m_prop_1
    m0_B = mF_A - 6 * m_struct_S
    mF_B = m0_B - 3 * m_prop_S - (tau_S / tau_1) * m_prop_1
    m0_C = mF_B - 3 * m_struct_S
    mF_C = m0_C - (1 - 2 * (tau_S / tau_1)) * m_prop_1
    m0_D = mF_C - m_struct_1
    mF_D = m_payload
    R_E_val, psi_L_val, omega_val = 6378145.0, (28.5 / 180) * np.pi, 7.29211585e-5
    g0_val = 9.80665

    prob = ProblemSpec("Delta III Launch Vehicle Ascent Problem")
    grav = -mu / (r_n**3)
    phase_data = [
        ("A", t_launch, t_sep_S, m0_A, mF_A, 6 * T_eng_S + T_eng_1, (1 / g0_val) * (6 * (T_eng_S / I_S) + T_eng_1 / I_1)),
        ("B", t_sep_S, t_sep_1, m0_B, mF_B, 3 * T_eng_S + T_eng_1, (1 / g0_val) * (3 * (T_eng_S / I_S) + T_eng_1 / I_1)),
        ("C", t_sep_1, t_sep_2, m0_C, mF_C, T_eng_1, T_eng_1 / (g0_val * I_1)),
        ("D", t_sep_2, t_orbit, m0_D, mF_D, T_eng_2, T_eng_2 / (g0_val * I_2)),
    ]
    phases = []
    for name, ta, tb, m_a, m_b, thrust, mdot in phase_data:
        ph = prob.new_phase(name)
        ph.state_variables = [r_x, r_y, r_z, v_x, v_y, v_z, m]
        ph.control_variables = [u_x, u_y, u_z]
        ph.state_equations = {r_x: v_x, r_y: v_y, r_z: v_z,
                              v_x: grav * r_x + T_over_m * u_x + D_x / m,
                              v_y: grav * r_y + T_over_m * u_y + D_y / m,
                              v_z: grav * r_z + T_over_m * u_z + D_z / m,
                              m: -xi}
        ph.path_constraints = [u_n - 1, r_n - R_E]
        ph.auxiliary_data = {T: thrust, xi: mdot}
        ph.bounds.initial_time = ta
        ph.bounds.final_time = tb
        ph.bounds.state_variables = {r_x: [-2 * R_E_val, 2 * R_E_val], r_y: [-2 * R_E_val, 2 * R_E_val],
                                     r_z: [-2 * R_E_val, 2 * R_E_val], v_x: [-10000, 10000],
                                     v_y: [-10000, 10000], v_z: [-10000, 10000], m: [m_b - (1.0 if burnout_mass else 0.0), m_a]}
        ph.bounds.control_variables = {u_x: [-1.1, 1.1], u_y: [-1.1, 1.1], u_z: [-1.1, 1.1]}
        ph.bounds.path_constraints = [[0, 0], [0, "inf"]]
        ph.bounds.initial_state_constraints = {m: m_a}
        ph.bounds.final_state_constraints = {m: [m_b - 1.0, m_a] if burnout_mass else m_b}
        # guess: at rest on the pad, mass linear, constant steering (delta_iii_launch_vehicle.py:273-281; the
        # example's later phases reuse phase B's time span in their guess, here each phase gets its own)
        vy0 = omega_val * R_E_val * np.cos(psi_L_val)
        ph.guess.time = np.array([ta, tb])
        ph.guess.state_variables = np.array([[R_E_val * np.cos(psi_L_val)] * 2, [0.0, 0.0], [R_E_val * np.sin(psi_L_val)] * 2,
                                             [0.0, 0.0], [vy0, vy0], [0.0, 0.0], [m_a, m_b]], dtype=float)
        ph.guess.control_variables = np.array([[0.9, 0.9], [0.05, 0.05], [0.45, 0.45]])
        _mesh(ph, K, order)
        phases.append(ph)
    A = phases[0]
    A.bounds.initial_state_constraints = {r_x: R_E_val * np.cos(psi_L_val), r_y: 0, r_z: R_E_val * np.sin(psi_L_val),
                                          v_x: 0, v_y: omega_val * R_E_val * np.cos(psi_L_val), v_z: 0, m: m0_A}
    Dp = phases[3]
    rf = Dp.final_state_variables
    prob.objective_function = -(sym.sqrt(rf[0]**2 + rf[1]**2 + rf[2]**2) - R_E)
    link = []
    for a, b in zip(phases[:-1], phases[1:]):
        for i in range(6):
            link.append(a.final_state_variables[i] - b.initial_state_variables[i])
    prob.endpoint_constraints = link
    prob.bounds.endpoint_constraints = [0] * len(link)
    prob.auxiliary_data = {
        mu: 3.986012e14, R_E: R_E_val,
        r_n: sym.sqrt(r_x**2 + r_y**2 + r_z**2),
        v_n: sym.sqrt(v_x**2 + v_y**2 + v_z**2),
        u_n: sym.sqrt(u_x**2 + u_y**2 + u_z**2),
        D_x: -0.5 * C_D * S * rho * v_r_n * v_r_x,
        D_y: -0.5 * C_D * S * rho * v_r_n * v_r_y,
        D_z: -0.5 * C_D * S * rho * v_r_n * v_r_z,
        C_D: 0.5, S: 4 * np.pi,
        v_r_n: sym.sqrt(v_r_x**2 + v_r_y**2 + v_r_z**2),
        v_r_x: v_x - oxr_x, v_r_y: v_y - oxr_y, v_r_z: v_z - oxr_z,
        oxr_x: -omega_E * r_y, oxr_y: omega_E * r_x, oxr_z: 0,
        g_0: g0_val, h_0: 7200, h: r_n - R_E, rho: rho_0 * sym.exp(-h / h_0), rho_0: 1.225,
        omega_E: omega_val, T_over_m: T / m, psi_L: psi_L_val,
    }
    return prob


def double_pendulum(K: int = 10, order: int = 4) -> ProblemSpec:
    """Double-pendulum swing-up test fixture (tests/unit/conftest.py:79-190): 4 states, 2 controls,
    1 integral, 2 static parameters (m0, p0), tF free."""
    a0, a1, v0, v1, T0, T1 = sym.symbols("a0 a1 v0 v1 T0 T1")
    g = sym.symbols("g")
    m0, p0, d0, l0, k0, I0 = sym.symbols("m0 p0 d0 l0 k0 I0")
    m1, p1, d1, l1, k1, I1 = sym.symbols("m1 p1 d1 l1 k1 I1")
    c0, s0, c1, s1 = sym.symbols("c0 s0 c1 s1")
    M00, M01, M10, M11, K0, K1, detM = sym.symbols("M00 M01 M10 M11 K0 K1 detM")
    prob = ProblemSpec("Double Pendulum Swing-Up")
    ph = prob.new_phase("A")
    ph.state_variables = [a0, a1, v0, v1]
    ph.control_variables = [T0, T1]
    ph.state_equations = [v0, v1, (M11 * K0 - M01 * K1) / detM, (M00 * K1 - M10 * K0) / detM]
    ph.integrand_functions = [T0**2 + T1**2]
    ph.auxiliary_data = {g: -9.81, k1: 1 / 12, I0: m0 * (k0**2 + p0**2), I1: m1 * (k1**2 + p1**2),
                         s0: sym.sin(a0), c1: sym.cos(a1)}
    prob.parameter_variables = [m0, p0]
    prob.objective_function = ph.integral_variables[0]
    prob.auxiliary_data = {
        g: 0, d0: 0.5, k0: 1 / 12, m1: 1.0, p1: 0.5, d1: 0.5, l0: p0 + d0, l1: p1 + d1,
        I0: m0 * (k0**2 + p0**2), I1: m1 * (k1**2 + p1**2),
        c0: sym.cos(a0), s0: sym.sin(a0), c1: sym.cos(a1), s1: sym.sin(a1),
        M00: I0 + m1 * l0**2, M01: m1 * p1 * l0 * (s0 * s1 + c0 * c1), M10: M01, M11: I1,
        K0: T0 + g * (m0 * p0 + m1 * l0) * c0 + m1 * p1 * l0 * (s1 * c0 - s0 * c1) * v1**2,
        K1: T1 + g * m1 * p1 * c1 + m1 * p1 * l0 * (s0 * c1 - s1 * c0) * v0**2,
        detM: M00 * M11 - M01 * M10,
    }
    ph.bounds.initial_time = 0
    ph.bounds.final_time = [1, 3]
    ph.bounds.state_variables = [[-np.pi, np.pi], [-np.pi, np.pi], [-10, 10], [-10, 10]]
    ph.bounds.control_variables = [[-15, 15], [-15, 15]]
    ph.bounds.integral_variables = [0, 1000]
    ph.bounds.initial_state_constraints = [[-0.5 * np.pi] * 2, [-0.5 * np.pi] * 2, [0, 0], [0, 0]]
    ph.bounds.final_state_constraints = [[0.5 * np.pi] * 2, [0.5 * np.pi] * 2, [0, 0], [0, 0]]
    prob.bounds.parameter_variables = [[0.5, 1.5], [0.5, 1.5]]
    ph.guess.time = np.array([0.0, 2.0])
    ph.guess.state_variables = np.array([[-0.5 * np.pi, 0.5 * np.pi], [-0.5 * np.pi, 0.5 * np.pi], [0, 0], [0, 0]], dtype=float)
    ph.guess.control_variables = np.array([[0.0, 0.0], [0.0, 0.0]])
    ph.guess.integral_variables = np.array([100.0])
    prob.guess.parameter_variables = np.array([1.0, 1.0])
    _mesh(ph, K, order)
    return prob


def two_phase_transfer(K: int = 4, order: int = 3) -> ProblemSpec:
    """Small synthetic multi-phase problem exercising every block type at once: two phases with
    different dynamics, free interior time, a path constraint, integrals, a static parameter that
    enters dynamics / path / integrand, linkage endpoint constraints and a nonlinear objective.
    (Not from the reference; it exists so that parity tests cover q/t/s coupling and phase offsets.)"""
    x, v, u, k = sym.symbols("x v u k")
    prob = ProblemSpec("two-phase transfer")
    prob.parameter_variables = [k]
    prob.bounds.parameter_variables = [[0.5, 2.0]]
    A = prob.new_phase("A")
    A.state_variables = [x, v]
    A.control_variables = [u]
    A.state_equations = [v, u - k * sym.sin(x) * v]
    A.path_constraints = [u**2 + k * x]
    A.integrand_functions = [u**2 + k * x**2]
    A.bounds.initial_time = 0.0
    A.bounds.final_time = [0.5, 2.0]
    A.bounds.state_variables = [[-2, 2], [-3, 3]]
    A.bounds.control_variables = [[-4, 4]]
    A.bounds.integral_variables = [[0, 50]]
    A.bounds.path_constraints = [[-1, 20]]
    A.bounds.initial_state_constraints = {x: 0, v: 0}
    B = prob.new_phase("B")
    B.state_variables = [x, v]
    B.control_variables = [u]
    B.state_equations = [v * sym.cos(x), -k * x + u * sym.exp(-v**2)]
    B.integrand_functions = [sym.sqrt(1 + u**2), x * v * k]
    B.bounds.initial_time = [0.5, 2.0]
    B.bounds.final_time = [2.5, 4.0]
    B.bounds.state_variables = [[-2, 2], [-3, 3]]
    B.bounds.control_variables = [[-4, 4]]
    B.bounds.integral_variables = [[0, 50], [-10, 10]]
    B.bounds.final_state_constraints = {x: 1.0, v: 0.0}
    prob.objective_function = (A.integral_variables[0] + B.integral_variables[0] * B.final_time_variable
                               + k * B.final_state_variables[1]**2 * A.initial_state_variables[0])
    prob.endpoint_constraints = [A.final_state_variables[0] - B.initial_state_variables[0],
                                 A.final_state_variables[1] - B.initial_state_variables[1],
                                 A.final_time_variable - B.initial_time_variable,
                                 B.integral_variables[1] * k - sym.sin(A.final_state_variables[0]) * B.final_time_variable]
    prob.bounds.endpoint_constraints = [0, 0, 0, [-1, 1]]
    _mesh(A, K, order)
    _mesh(B, K + 1, order + 1)
    return prob


def time_coupled_transfer(K: int = 4, order: int = 3) -> ProblemSpec:
    """``two_phase_transfer`` with dynamics, path constraints and integrands that also depend on the phase's
    integral variables and on its initial / final time *variables* -- the live reference keeps q, t0, tF as global
    symbols inside f, p, g (pycollo/backend.py:1526-1539 substitutes y and u per node only).  Covers: q / t columns
    of defect, path and integral rows, (q, z) Hessian strips, second partials merged into the stretch strips and
    sums, a control that meets a time variable in a path row only, an integrand that depends on its own integral.
    (Not from the reference.)"""
    x, v, u, w, k = sym.symbols("x v u w k")
    prob = ProblemSpec("time-coupled transfer")
    prob.parameter_variables = [k]
    prob.bounds.parameter_variables = [[0.5, 2.0]]
    A = prob.new_phase("A")
    A.state_variables = [x, v]
    A.control_variables = [u, w]
    A.integrand_functions = [0, 0]          # placeholders: the integral symbols must exist before they are used
    qA = A.integral_variables
    tFA = A.final_time_variable
    A.state_equations = [v * (1 + sym.Rational(1, 10) * tFA), u - k * sym.sin(x) * v + sym.Rational(1, 20) * qA[0] * x]
    A.path_constraints = [u**2 + k * x + w * tFA + sym.Rational(1, 10) * qA[0]**2]
    A.integrand_functions = [u**2 + k * x**2 + sym.Rational(1, 5) * qA[0] * v + sym.Rational(1, 100) * tFA**2 * x,
                             w**2 + qA[0] * tFA * sym.Rational(1, 50)]
    A.bounds.initial_time = 0.0
    A.bounds.final_time = [0.5, 2.0]
    A.bounds.state_variables = [[-2, 2], [-3, 3]]
    A.bounds.control_variables = [[-4, 4], [-1, 1]]
    A.bounds.integral_variables = [[0, 50], [0, 10]]
    A.bounds.path_constraints = [[-1, 40]]
    A.bounds.initial_state_constraints = {x: 0, v: 0}
    B = prob.new_phase("B")
    B.state_variables = [x, v]
    B.control_variables = [u]
    B.integrand_functions = [0, 0]
    qB = B.integral_variables
    t0B, tFB = B.initial_time_variable, B.final_time_variable
    B.state_equations = [v * sym.cos(x) + sym.Rational(1, 10) * (tFB - t0B) * qB[1],
                         -k * x + u * sym.exp(-v**2) * t0B]
    B.path_constraints = [x * qB[1] + t0B * k]
    B.integrand_functions = [sym.sqrt(1 + u**2) + sym.Rational(1, 10) * qB[1] * qB[0] * x,
                             x * v * k + t0B * tFB * u + sym.Rational(3, 10) * qB[0]]
    B.bounds.initial_time = [0.5, 2.0]
    B.bounds.final_time = [2.5, 4.0]
    B.bounds.state_variables = [[-2, 2], [-3, 3]]
    B.bounds.control_variables = [[-4, 4]]
    B.bounds.integral_variables = [[0, 50], [-10, 10]]
    B.bounds.path_constraints = [[-30, 30]]
    B.bounds.final_state_constraints = {x: 1.0, v: 0.0}
    # (the q x y(tF) and t x y(t0) products put endpoint terms on the ends of a q strip and of a t strip)
    prob.objective_function = (qA[0] + qB[0] * tFB + k * B.final_state_variables[1]**2 * A.initial_state_variables[0]
                               + sym.Rational(1, 10) * qA[0] * A.final_state_variables[1])
    prob.endpoint_constraints = [A.final_state_variables[0] - B.initial_state_variables[0],
                                 A.final_state_variables[1] - B.initial_state_variables[1],
                                 tFA - t0B,
                                 qB[1] * k - sym.sin(A.final_state_variables[0]) * tFB + qA[1]
                                 + t0B * B.initial_state_variables[0] * qB[1]]
    prob.bounds.endpoint_constraints = [0, 0, 0, [-1, 1]]
    _mesh(A, K, order)
    _mesh(B, K + 1, order + 1)
    return prob


def time_scaled_transfer(K: int = 10, order: int = 4) -> ProblemSpec:
    """Rest-to-rest transfer of a unit mass over unit distance whose actuator weakens with the manoeuvre's duration:
    ``dv/dt = u / tF`` -- the final-time *variable* inside a state equation -- with objective ``q + tF``, ``q`` the
    integral of ``u**2``, and an (inactive) path constraint that involves ``q``.  With a = u / tF the minimum of
    int a^2 over rest-to-rest transfers in time T is 12 / T^3, so q = 12 / T and J = 12 / T + T: T* = sqrt(12),
    J* = 2 sqrt(12) = 6.92820323...  (Not from the reference: it exists to solve an NLP whose node functions depend on
    tF and q end to end; the live reference allows both, pycollo/backend.py:1526-1539.)"""
    x, v, u = sym.symbols("x v u")
    prob = ProblemSpec("time-scaled transfer")
    ph = prob.new_phase("A")
    ph.state_variables = [x, v]
    ph.control_variables = [u]
    ph.integrand_functions = [u**2]
    q0, tF = ph.integral_variables[0], ph.final_time_variable
    ph.state_equations = [v, u / tF]
    ph.path_constraints = [x - q0 / 200]
    prob.objective_function = q0 + tF
    ph.bounds.initial_time = 0.0
    ph.bounds.final_time = [1.0, 8.0]
    ph.bounds.state_variables = [[-1, 2], [-5, 5]]
    ph.bounds.control_variables = [[-20, 20]]
    ph.bounds.integral_variables = [[0, 100]]
    ph.bounds.path_constraints = [[-5, 1.5]]
    ph.bounds.initial_state_constraints = {x: 0, v: 0}
    ph.bounds.final_state_constraints = {x: 1, v: 0}
    ph.guess.time = np.array([0.0, 3.0])
    ph.guess.state_variables = np.array([[0.0, 1.0], [0.0, 0.0]])
    ph.guess.control_variables = np.array([[0.0, 0.0]])
    ph.guess.integral_variables = np.array([4.0])
    _mesh(ph, K, order)
    return prob


def sliding_mass(num_phases: int = 2, K: int = 10, order: int = 4) -> ProblemSpec:
    """Unit mass slid from x = 0 to x = 1 in minimum time, at rest at both ends, split into ``num_phases`` phases
    of equal distance with velocity / time continuity as endpoint constraints -- the problem of the reference's
    multi-phase integration test (tests/integration/test_multiphase.py:25-74; expected objective 0.4472136 =
    2 / sqrt(20) for every phase count)."""
    x, v, f = sym.symbols("x v f")
    MAX_T, MAX_V, MAX_F = 1.0, 10.0, 20.0
    prob = ProblemSpec(f"{num_phases}-phase sliding mass")
    for i in range(num_phases):
        x0, x1 = i / num_phases, (i + 1) / num_phases
        ph = prob.new_phase("ABCDEFGH"[i])
        ph.state_variables = [x, v]
        ph.control_variables = [f]
        ph.state_equations = [v, f]
        ph.bounds.initial_time = [0, MAX_T] if i else 0.0
        ph.bounds.final_time = [0, MAX_T]
        ph.bounds.initial_state_constraints = {x: x0, v: [0, MAX_V] if i else 0}
        ph.bounds.state_variables = [[x0, x1], [0, MAX_V]]
        ph.bounds.final_state_constraints = {x: x1, v: [0, MAX_V] if (i + 1) != num_phases else 0}
        ph.bounds.control_variables = [[-MAX_F, MAX_F]]
        ph.guess.time = np.array([x0 * MAX_T, x1 * MAX_T])
        ph.guess.state_variables = np.array([[x0, x1], [0.0, 0.0]])
        ph.guess.control_variables = np.array([[0.0, 0.0]])
        _mesh(ph, K, order)
    if num_phases >= 2:
        cons = []
        for p1, p2 in zip(prob.phases[:-1], prob.phases[1:]):
            cons.append(p1.final_state_variables[1] - p2.initial_state_variables[1])
            cons.append(p1.final_time_variable - p2.initial_time_variable)
        prob.endpoint_constraints = cons
        prob.bounds.endpoint_constraints = [0] * len(cons)
    prob.objective_function = prob.phases[-1].final_time_variable
    return prob


def free_flying_robot(K: int = 10, order: int = 4) -> ProblemSpec:
    """Sakawa free-flying robot, 6 states, 4 one-sided thruster controls, 2 path inequality rows, fuel integral;
    the problem of the reference's integration test (tests/integration/test_free_flying_robot.py:20-186; expected
    objective 7.9101902 (GPOPS-II) / 7.910154646 (SOS), mesh tolerance 1e-5, at most 15 mesh iterations)."""
    r_x, r_y, theta, v_x, v_y, omega = sym.symbols("r_x r_y theta v_x v_y omega")
    uxp, uxn, uyp, uyn = sym.symbols("u_x_pos u_x_neg u_y_pos u_y_neg")
    T_x, T_y, I_xx, I_yy = sym.symbols("T_x T_y I_xx I_yy")
    prob = ProblemSpec("Free-Flying Robot")
    ph = prob.new_phase("A")
    ph.state_variables = [r_x, r_y, theta, v_x, v_y, omega]
    ph.control_variables = [uxp, uxn, uyp, uyn]
    ph.state_equations = {r_x: v_x, r_y: v_y, theta: omega,
                          v_x: (T_x + T_y) * sym.cos(theta), v_y: (T_x + T_y) * sym.sin(theta),
                          omega: I_xx * T_x - I_yy * T_y}
    ph.integrand_functions = [uxp + uxn + uyp + uyn]
    ph.path_constraints = [uxp + uxn, uyp + uyn]
    prob.objective_function = ph.integral_variables[0]
    prob.auxiliary_data = {I_xx: 0.2, I_yy: 0.2, T_x: uxp - uxn, T_y: uyp - uyn}
    ph.bounds.initial_time = 0.0
    ph.bounds.final_time = 12.0
    ph.bounds.state_variables = {r_x: [-10, 10], r_y: [-10, 10], theta: [-np.pi, np.pi], v_x: [-2, 2], v_y: [-2, 2],
                                 omega: [-1, 1]}
    ph.bounds.initial_state_constraints = {r_x: -10, r_y: -10, theta: np.pi / 2, v_x: 0, v_y: 0, omega: 0}
    ph.bounds.final_state_constraints = {r_x: 0, r_y: 0, theta: 0, v_x: 0, v_y: 0, omega: 0}
    ph.bounds.control_variables = [[0, 1000]] * 4
    ph.bounds.integral_variables = [[0, 100]]
    ph.bounds.path_constraints = [[-1000, 1], [-1000, 1]]
    ph.guess.time = np.array([0.0, 12.0])
    ph.guess.state_variables = np.array([[-10, 0], [-10, 0], [np.pi / 2, 0], [0, 0], [0, 0], [0, 0]], dtype=float)
    ph.guess.control_variables = np.zeros((4, 2))
    ph.guess.integral_variables = np.array([0.0])
    _mesh(ph, K, order)
    return prob


def tumour_anti_angiogenesis(K: int = 10, order: int = 4) -> ProblemSpec:
    """Ledzewicz-Schaettler tumour anti-angiogenesis: 2 states, 1 control, 1 integral (total drug) with an upper
    bound, free final time, Mayer objective p(tF); log and fractional-power dynamics.  The problem of the reference's
    integration test (tests/integration/test_tumour_anti_angiogenesis.py:20-111; expected objective 7571.66986
    (GPOPS-II) / 7571.6831 (SOS), rtol 1e-5)."""
    p_, q_, u_ = sym.symbols("p q u")
    xi, b, d, G, mu = sym.symbols("xi b d G mu")
    b_v, mu_v, d_v, a_v, A_v = 5.85, 0.02, 0.00873, 75.0, 15.0
    p_max = ((b_v - mu_v) / d_v) ** 1.5
    p_min = 0.1
    p_t0, q_t0 = p_max / 2, p_max / 4
    prob = ProblemSpec("Tumour Anti-Angiogenesis")
    ph = prob.new_phase("A")
    ph.state_variables = [p_, q_]
    ph.control_variables = [u_]
    ph.state_equations = {p_: -xi * p_ * sym.log(p_ / q_), q_: q_ * (b - (mu + d * p_ ** (2 / 3) + G * u_))}
    ph.integrand_functions = [u_]
    prob.objective_function = ph.final_state_variables[0]
    prob.auxiliary_data = {xi: 0.084, b: b_v, d: d_v, G: 0.15, mu: mu_v}
    ph.bounds.initial_time = 0.0
    ph.bounds.final_time = [0.1, 5.0]
    ph.bounds.state_variables = {p_: [p_min, p_max], q_: [p_min, p_max]}
    ph.bounds.control_variables = [[0.0, a_v]]
    ph.bounds.integral_variables = [[0.0, A_v]]
    ph.bounds.initial_state_constraints = {p_: p_t0, q_: q_t0}
    ph.guess.time = np.array([0.0, 1.0])
    ph.guess.state_variables = np.array([[p_t0, p_max], [q_t0, p_max]])
    ph.guess.control_variables = np.array([[a_v, a_v]])
    ph.guess.integral_variables = np.array([7.5])
    _mesh(ph, K, order)
    return prob


def space_station(K: int = 10, order: int = 4) -> ProblemSpec:
    """Betts' space-station attitude control: 9 states (angular velocity, Euler-Rodrigues parameters, control-moment-
    gyro momentum), 3 torque controls, momentum-magnitude path row, control-effort integral, six nonlinear endpoint
    rows (zero angular and attitude acceleration at tF).  The problem of the reference's integration test
    (tests/integration/test_space_station_attitute_control.py:24-285; expected objective 3.58675 (GPOPS-II) /
    3.58688 (SOS), rtol 1e-4)."""
    wx, wy, wz, rx, ry, rz, hx, hy, hz = sym.symbols("omega_x omega_y omega_z r_x r_y r_z h_x h_y h_z")
    ux, uy, uz = sym.symbols("u_x u_y u_z")
    Jm = sym.Matrix([[2.80701911616e7, 4.822509936e5, -1.71675094448e7],
                     [4.822509936e5, 9.5144639344e7, 6.02604448e4],
                     [-1.71675094448e7, 6.02604448e4, 7.6594401336e7]])
    Jinv = Jm.inv()
    w_orb = 0.06511 * np.pi / 180
    h_max = 10000.0

    def skew(vv):
        return sym.Matrix([[0, -vv[2], vv[1]], [vv[2], 0, -vv[0]], [-vv[1], vv[0], 0]])

    def rates(om, r, h, u):
        I3 = sym.eye(3)
        rs = skew(r)
        C = I3 + (2 / (1 + r.dot(r))) * (rs * rs - rs)
        tau_gg = 3 * w_orb**2 * skew(C[:, 2]) * (Jm * C[:, 2])
        dom = Jinv * (tau_gg - skew(om) * (Jm * om + h) - u)
        om0 = -w_orb * C[:, 1]
        dr = sym.Rational(1, 2) * (r * r.T + I3 + rs) * (om - om0)
        return dom, dr

    prob = ProblemSpec("Space Station Attitude Control")
    ph = prob.new_phase("A")
    ph.state_variables = [wx, wy, wz, rx, ry, rz, hx, hy, hz]
    ph.control_variables = [ux, uy, uz]
    dom, dr = rates(sym.Matrix([wx, wy, wz]), sym.Matrix([rx, ry, rz]), sym.Matrix([hx, hy, hz]), sym.Matrix([ux, uy, uz]))
    ph.state_equations = [dom[0], dom[1], dom[2], dr[0], dr[1], dr[2], ux, uy, uz]
    ph.path_constraints = [hx**2 + hy**2 + hz**2]
    ph.integrand_functions = [1e-6 * (ux**2 + uy**2 + uz**2)]
    prob.objective_function = ph.integral_variables[0]
    yF = ph.final_state_variables
    domF, drF = rates(sym.Matrix(yF[0:3]), sym.Matrix(yF[3:6]), sym.Matrix(yF[6:9]), sym.zeros(3, 1))
    prob.endpoint_constraints = [domF[0], domF[1], domF[2], drF[0], drF[1], drF[2]]
    prob.bounds.endpoint_constraints = [0] * 6
    ph.bounds.initial_time = 0.0
    ph.bounds.final_time = 1800.0
    ph.bounds.state_variables = [[-2e-3, 2e-3]] * 3 + [[-1, 1]] * 3 + [[-15000, 15000]] * 3
    y0 = [-9.5380685844896e-6, -1.1363312657036e-3, 5.3472801108427e-6, 2.9963689649816e-3, 1.5334477761054e-1,
          3.8359805613992e-3, 5000.0, 5000.0, 5000.0]
    ph.bounds.initial_state_constraints = dict(zip(ph.state_variables, y0))
    ph.bounds.final_state_constraints = {hx: 0, hy: 0, hz: 0}
    ph.bounds.control_variables = [[-150, 150]] * 3
    ph.bounds.integral_variables = [[0, 10]]
    ph.bounds.path_constraints = [[0, h_max**2]]
    ph.guess.time = np.array([0.0, 1800.0])
    ph.guess.state_variables = np.array([[v, v] for v in y0])
    ph.guess.control_variables = np.zeros((3, 2))
    ph.guess.integral_variables = np.array([10.0])
    _mesh(ph, K, order)
    return prob


REGISTRY = {
    "brachistochrone": brachistochrone,
    "hypersensitive": hypersensitive,
    "cart_pole": cart_pole,
    "shuttle": shuttle,
    "delta_iii": delta_iii,
    "double_pendulum": double_pendulum,
    "two_phase_transfer": two_phase_transfer,
    "time_coupled_transfer": time_coupled_transfer,
    "time_scaled_transfer": time_scaled_transfer,
    "sliding_mass": sliding_mass,
    "free_flying_robot": free_flying_robot,
    "tumour_anti_angiogenesis": tumour_anti_angiogenesis,
    "space_station": space_station,
}


def with_refined_mesh(prob: ProblemSpec, nodes_per_phase: int, seeds=None) -> ProblemSpec:
    """Give every phase of ``prob`` a mesh as ph refinement leaves it (refinement.synthetic_refined_mesh: the ph rule
    iterated on a synthetic error field), phase i from seed ``seeds[i]`` (default 7, 8, ...).  The workload of bench.py
    --refined and of the refined-mesh parity tests (BASELINE.json configs[4]: "ph-adaptive mesh refinement to ~50k nodes")."""
    from ..refinement import synthetic_refined_mesh
    seeds = tuple(seeds) if seeds is not None else tuple(7 + i for i in range(len(prob.phases)))
    for ph, seed in zip(prob.phases, seeds):
        sizes, nodes = synthetic_refined_mesh(int(nodes_per_phase), int(seed))
        ph.mesh.number_mesh_sections = int(nodes.size)
        ph.mesh.mesh_section_sizes = sizes
        ph.mesh.number_mesh_section_nodes = nodes
    return prob


def delta_iii_flown_guess(prob: ProblemSpec, points: int = 60) -> ProblemSpec:
    """Replace the guess of :func:`delta_iii` by a trajectory that was actually flown: the equations of motion of the
    example (delta_iii_launch_vehicle.py:198-459) integrated phase after phase from the pad with the thrust along the
    local vertical, ``points`` samples per phase.  The example's own guess -- the vehicle standing on the pad for all
    four phases -- has zero air speed under the drag term's square root, i.e. NaN partial derivatives at every node,
    and violates the dynamics everywhere; this one satisfies dynamics, mass schedule, linkages and path constraints, so
    an interior-point solve starts feasible."""
    from scipy.integrate import solve_ivp
    mu, R_E, om, h_0, rho_0, C_D, S = 3.986012e14, 6378145.0, 7.29211585e-5, 7200.0, 1.225, 0.5, 4 * np.pi
    T, xi = sym.symbols("T xi")

    def rhs(thrust, mdot):
        def f(t, y):
            r, v, m = y[:3], y[3:6], y[6]
            rn = np.linalg.norm(r)
            vr = v - np.array([-om * r[1], om * r[0], 0.0])
            drag = -0.5 * C_D * S * rho_0 * np.exp(-(rn - R_E) / h_0) * np.linalg.norm(vr) * vr
            return np.concatenate([v, -mu / rn**3 * r + thrust / m * (r / rn) + drag / m, [-mdot]])
        return f

    y_end = None
    for ph in prob.phases:
        ta, tb = float(ph.bounds.initial_time), float(ph.bounds.final_time)
        m_a = float(ph.guess.state_variables[6][0])
        y0 = np.array([ph.guess.state_variables[i][0] for i in range(7)], float) if y_end is None else np.append(y_end[:6], m_a)
        thrust, mdot = float(ph.auxiliary_data[T]), float(ph.auxiliary_data[xi])
        t = np.linspace(ta, tb, points)
        sol = solve_ivp(rhs(thrust, mdot), (ta, tb), y0, t_eval=t, rtol=1e-10, atol=1e-6)
        if not sol.success:
            raise RuntimeError(f"phase {ph.name}: the guess trajectory could not be integrated: {sol.message}")
        y = sol.y
        ph.guess.time = t
        ph.guess.state_variables = y
        ph.guess.control_variables = y[:3] / np.linalg.norm(y[:3], axis=0)
        y_end = y[:, -1]
    return prob
