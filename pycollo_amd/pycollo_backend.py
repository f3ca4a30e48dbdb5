"""The pycollo-side backend: ``OptimalControlProblem`` -> MI355X callback engine, behind pycollo's own backend surface.

What it replaces in the reference (``/root/reference/pycollo/backend.py``):

* ``Casadi.generate_nlp_function_callables`` (:1403-1411) -> :meth:`Mi355x.generate_nlp_function_callables`: instead
  of building ``ca.SX`` graphs for J, g, c, G per mesh iteration, the user's problem is lowered once
  (:func:`to_problem_spec`), the iteration's mesh is read (:func:`phase_mesh`) and an :class:`NlpEngine` is created
  (``pc_create``: patterns + code object).
* ``Casadi.create_nlp_solver`` / ``solve_nlp`` (:1681-1711, :1807-1827) -> the iteration's scaling is handed over
  (``pc_set_scaling``), the cyipopt-protocol object (pycollo/nlp.py:47-76) is built and given to IPOPT where cyipopt
  exists, else to the stand-in behind the same four calls (``pycollo_amd.ipopt_api``).
* the probes ``evaluate_J / g / c / G / G_nonzeros / G_structure / G_num_nonzero`` (:1713-1771) and the unimplemented
  ``evaluate_H*`` (:1773-1805), plus the three scaling-time callables ``g_iter_scale_callable``,
  ``G_iter_scale_callable``, ``dy_iter_callable`` (:1511, :1679, :1666-1668).

A pycollo maintainer registers it next to ``"casadi"`` (backend.py:1925-1927)::

    from pycollo_amd.pycollo_backend import Mi355x as _Mi355xMixin
    class Mi355x(_Mi355xMixin, BackendABC): pass
    BACKENDS = Options(("casadi", "mi355x", ...), handles=(Casadi, Mi355x, ...))

pycollo itself cannot be imported in this image (casadi, pyproprop are absent), so nothing here imports it: the
adapter is duck-typed on the attribute names the reference's objects expose -- the user-facing
``OptimalControlProblem`` / ``Phase`` / ``PhaseBounds`` / ``PhaseMesh`` (phase.py:303-565, optimal_control_problem.py:
113-305, mesh.py:10-107), the iteration's ``Mesh`` (mesh.py:110-235), ``IterationScaling`` (scaling.py:166-169,
273-281) and the counts of the live backend (``p[i].num_each_var``, ``num_s_var``, ``num_c``; backend.py:632-816,
1212-1306) -- and tested against stub objects carrying exactly those names (tests/pycollo_stub.py).  Expressions are
read from the USER's problem (SymPy, as the user wrote them), never from the backend's ``ca.SX`` copies.
"""
from __future__ import annotations

import time
from types import SimpleNamespace

import numpy as np
import sympy as sym

from .engine import NlpEngine, PycolloGpuProblem
from .mesh import build_phase_mesh
from .problem import ProblemSpec
from .quadrature import QuadratureTables


# --------------------------------------------------------------------------------------------------
# user problem -> ProblemSpec
# --------------------------------------------------------------------------------------------------
def _seq(x):
    if x is None:
        return []
    if isinstance(x, (sym.Basic, int, float)):
        return [x]
    if isinstance(x, dict):
        return list(x.values())
    return list(x)


def _copy_bounds(b):
    """A bounds entry as the user gave it: dict keyed by symbol, sequence in variable order, scalar, or None."""
    if b is None:
        return None
    if isinstance(b, dict):
        return dict(b)
    if isinstance(b, np.ndarray):
        return b.copy()
    if isinstance(b, (list, tuple)):
        return [(_copy_bounds(e) if isinstance(e, (list, tuple, np.ndarray)) else e) for e in b]
    return b


def to_problem_spec(ocp) -> ProblemSpec:
    """Lower a pycollo ``OptimalControlProblem`` (or anything with its attribute names) to the engine's problem
    description.  Reads: ``ocp.name, phases, parameter_variables, objective_function, endpoint_constraints,
    auxiliary_data, bounds.parameter_variables / endpoint_constraints, guess.parameter_variables,
    settings.scaling_method / quadrature_method`` and, per phase, ``name, state_variables, control_variables,
    state_equations, path_constraints, integrand_functions, auxiliary_data, initial / final_time_variable, initial /
    final_state_variables, integral_variables, bounds.*, guess.*, mesh.number_mesh_sections / mesh_section_sizes /
    number_mesh_section_nodes``.  The user's own state / control / parameter symbols are kept; pycollo's generated
    endpoint symbols (``t0_P0``, ``y_P0(tF)``, the integral variables) are renamed to the engine's, everywhere they
    occur (objective, endpoint constraints, auxiliary data, and -- the live reference allows it, backend.py:1526-1539 --
    inside state equations, path constraints and integrands)."""
    spec = ProblemSpec(str(getattr(ocp, "name", "ocp")))
    spec.parameter_variables = _seq(getattr(ocp, "parameter_variables", None))
    rename: dict = {}
    pairs = []
    for ph in ocp.phases:
        new = spec.new_phase(str(getattr(ph, "name", len(spec.phases))))
        new.state_variables = _seq(ph.state_variables)
        new.control_variables = _seq(getattr(ph, "control_variables", None))
        eqns = ph.state_equations
        new.state_equations = dict(eqns) if isinstance(eqns, dict) else _seq(eqns)
        new.path_constraints = [sym.sympify(e) for e in _seq(getattr(ph, "path_constraints", None))]
        new.integrand_functions = _seq(getattr(ph, "integrand_functions", None))
        rename[ph.initial_time_variable] = new.initial_time_variable
        rename[ph.final_time_variable] = new.final_time_variable
        for a, b in zip(_seq(ph.initial_state_variables), new.initial_state_variables):
            rename[a] = b
        for a, b in zip(_seq(ph.final_state_variables), new.final_state_variables):
            rename[a] = b
        q_ref = _seq(getattr(ph, "integral_variables", None))
        if len(q_ref) != len(new.integral_variables):
            raise ValueError(f"phase {new.name}: {len(q_ref)} integral variables for {len(new.integral_variables)} integrands")
        for a, b in zip(q_ref, new.integral_variables):
            rename[a] = b
        pairs.append((ph, new))
    rename = {k: v for k, v in rename.items() if k != v}

    def rn(e):
        return sym.sympify(e).xreplace(rename)

    for ph, new in pairs:
        new._y_eqn = [rn(e) for e in new._y_eqn]
        new.path_constraints = [rn(e) for e in new.path_constraints]
        new._q_fnc = [rn(e) for e in new._q_fnc]
        new.auxiliary_data = {rn(k): rn(v) for k, v in dict(getattr(ph, "auxiliary_data", None) or {}).items()}
        for name in ("initial_time", "final_time", "state_variables", "control_variables", "integral_variables",
                     "path_constraints", "initial_state_constraints", "final_state_constraints"):
            setattr(new.bounds, name, _copy_bounds(getattr(ph.bounds, name, None)))
        g = getattr(ph, "guess", None)
        for name in ("time", "state_variables", "control_variables", "integral_variables"):
            v = getattr(g, name, None) if g is not None else None
            setattr(new.guess, name, None if v is None else np.array(v, dtype=float))
        m = ph.mesh
        new.mesh.number_mesh_sections = int(m.number_mesh_sections)
        sizes = getattr(m, "mesh_section_sizes", None)
        new.mesh.mesh_section_sizes = None if sizes is None else np.array(sizes, dtype=float)
        nodes = m.number_mesh_section_nodes
        new.mesh.number_mesh_section_nodes = int(nodes) if np.ndim(nodes) == 0 else np.array(nodes, dtype=np.int64)
    spec.objective_function = rn(ocp.objective_function)
    spec.endpoint_constraints = [rn(e) for e in _seq(getattr(ocp, "endpoint_constraints", None))]
    spec.auxiliary_data = {rn(k): rn(v) for k, v in dict(getattr(ocp, "auxiliary_data", None) or {}).items()}
    b = getattr(ocp, "bounds", None)
    spec.bounds.parameter_variables = _copy_bounds(getattr(b, "parameter_variables", None))
    spec.bounds.endpoint_constraints = _copy_bounds(getattr(b, "endpoint_constraints", None))
    g = getattr(ocp, "guess", None)
    pv = getattr(g, "parameter_variables", None) if g is not None else None
    spec.guess.parameter_variables = None if pv is None else np.array(pv, dtype=float)
    st = getattr(ocp, "settings", None)
    if st is not None:
        method = getattr(st, "scaling_method", "bounds")
        spec.scaling_method = None if method in (None, "none") else str(method)
        spec.quadrature_method = str(getattr(st, "quadrature_method", "lobatto")).lower()
    return spec


def phase_mesh(mesh, p: int, quad: QuadratureTables):
    """Phase ``p`` of an iteration's ``Mesh`` (pycollo/mesh.py:110-235) as the engine's mesh tables.  Reads the phase
    mesh description ``mesh.p[p]`` (``mesh_section_sizes``, ``number_mesh_section_nodes``) and, where the generated
    arrays are present (``mesh.N / K / N_K / tau``), checks that the engine's tables describe the same mesh."""
    pm = mesh.p[p]
    K = int(pm.number_mesh_sections)
    sizes = getattr(pm, "mesh_section_sizes", None)
    sizes = np.ones(K) / K if sizes is None else np.asarray(sizes, dtype=float)
    nodes = pm.number_mesh_section_nodes
    nodes = np.full(K, int(nodes), dtype=np.int64) if np.ndim(nodes) == 0 else np.asarray(nodes, dtype=np.int64)
    if len(sizes) != K or len(nodes) != K:
        raise ValueError(f"phase {p}: mesh description is inconsistent with number_mesh_sections")
    out = build_phase_mesh(quad, sizes / sizes.sum(), nodes)
    for name, mine in (("N", out.N), ("K", out.K)):
        ref = getattr(mesh, name, None)
        if ref is not None and len(ref) > p and int(ref[p]) != int(mine):
            raise ValueError(f"phase {p}: the iteration's mesh has {name} = {ref[p]}, the engine's tables {mine}")
    ref_nk = getattr(mesh, "N_K", None)
    if ref_nk is not None and len(ref_nk) > p and not np.array_equal(np.asarray(ref_nk[p], dtype=np.int64), nodes):
        raise ValueError(f"phase {p}: nodes per section differ from the iteration's N_K")
    ref_tau = getattr(mesh, "tau", None)
    if ref_tau is not None and len(ref_tau) > p and not np.allclose(np.asarray(ref_tau[p], float), out.tau, rtol=0, atol=1e-12):
        raise ValueError(f"phase {p}: node positions differ from the iteration's tau")
    return out


# --------------------------------------------------------------------------------------------------
# the backend
# --------------------------------------------------------------------------------------------------
class NlpResult(SimpleNamespace):
    """(solution, info, solve_time) -- the named tuple of pycollo/backend.py:57-61 as attributes."""


class Mi355x:
    """MI355X backend: method names, argument meaning and return shapes of the reference's ``Casadi`` backend for the
    path this work covers.  Mix it in front of ``BackendABC`` inside pycollo (module docstring); on its own it only
    needs ``ocp`` (and, when present, checks itself against the live backend's counts)."""

    def __init__(self, ocp=None, *, device: int | None = 0):
        if ocp is not None:
            self.ocp = ocp
        self.device = device
        self.engine: NlpEngine | None = None
        self.problem_obj: PycolloGpuProblem | None = None
        self.current_iteration = None
        self._spec = None

    # ---- per mesh iteration (backend.py:1403-1411) ------------------------------------------------
    def generate_nlp_function_callables(self, iteration):
        self.current_iteration = iteration
        if self._spec is None:
            self._spec = to_problem_spec(self.ocp)
            self._quad = QuadratureTables(self._spec.quadrature_method)
        meshes = [phase_mesh(iteration.mesh, i, self._quad) for i in range(len(self._spec.phases))]
        if self.engine is not None:
            self.engine.close()
        self.engine = NlpEngine(self._spec, meshes, device=self.device, quad=self._quad)
        self._check_against_live_backend()
        self.problem_obj = None
        # the three scaling-time callables the reference builds here (backend.py:1506-1511, 1666-1679)
        self.g_iter_scale_callable = self._g_scale
        self.G_iter_scale_callable = self._G_scale
        self.dy_iter_callable = self._dy

    def _check_against_live_backend(self):
        """Where the object also is a live pycollo backend (``self.p`` = PycolloPhaseData list, backend.py:632-816),
        its counts must be the engine's: same needed variables in the same order, same constraint counts."""
        lay = self.engine.layout
        phases = getattr(self, "p", None)
        if phases is not None:
            for i, (p, pl) in enumerate(zip(phases, lay.phases)):
                each = tuple(int(v) for v in p.num_each_var)
                if each != (pl.n_y, pl.n_u, pl.n_q, pl.n_t):
                    raise RuntimeError(f"phase {i}: pycollo counts (y, u, q, t) = {each}, engine {(pl.n_y, pl.n_u, pl.n_q, pl.n_t)}")
                for name, mine in (("num_y_eqn", pl.n_y), ("num_p_con", pl.n_p), ("num_q_fnc", pl.n_q)):
                    ref = getattr(p, name, None)
                    if ref is not None and int(ref) != mine:
                        raise RuntimeError(f"phase {i}: pycollo {name} = {ref}, engine {mine}")
        for name, mine in (("num_s_var", lay.n_s), ("num_b_con", lay.n_b), ("num_c", lay.num_ocp_c), ("num_var", lay.num_ocp_x)):
            ref = getattr(self, name, None)
            if ref is not None and int(ref) != mine:
                raise RuntimeError(f"pycollo {name} = {ref}, engine {mine}")
        it = self.current_iteration
        for name, mine in (("num_x", self.engine.num_x), ("num_c", self.engine.num_c)):
            ref = getattr(it, name, None)
            if ref is not None and int(ref) != mine:
                raise RuntimeError(f"iteration {name} = {ref}, engine {mine}")

    # ---- solver (backend.py:1681-1711, 1807-1827) ---------------------------------------------------
    def create_nlp_solver(self):
        s = self.current_iteration.scaling
        self.engine.set_scaling(np.asarray(s.V_ocp, float), np.asarray(s.r_ocp, float), np.asarray(s.W_ocp, float), float(s.w))
        self.problem_obj = PycolloGpuProblem(self.engine)
        self.nlp_solver = self._nlp_solver

    def create_nlp_solver_settings(self):
        st = self.ocp.settings
        return {"tol": st.nlp_tolerance, "max_iter": st.max_nlp_iterations, "linear_solver": getattr(st, "linear_solver", "mumps"),
                "mu_strategy": "adaptive", "mu_min": 1e-11, "warm_start_init_point": "yes" if getattr(st, "warm_start", False) else "no"}

    def _nlp_solver(self, x0, lbx, ubx, lbg, ubg):
        """``ca.nlpsol``'s call signature and result keys (backend.py:1815-1819, solution/casadi_solution.py)."""
        try:
            import ipopt                       # cyipopt's legacy module name, as pycollo/nlp.py imports it
            have_ipopt = True
        except ImportError:
            from . import ipopt_api as ipopt   # same four calls over the stand-in interior-point method
            have_ipopt = False
        nlp = ipopt.problem(n=self.engine.num_x, m=self.engine.num_c, problem_obj=self.problem_obj,
                            lb=np.asarray(lbx, float), ub=np.asarray(ubx, float), cl=np.asarray(lbg, float), cu=np.asarray(ubg, float))
        for k, v in self.create_nlp_solver_settings().items():
            if k == "linear_solver" and not have_ipopt:
                continue                       # the stand-in's KKT systems are factorised on the GPU (pc_kkt_*)
            nlp.addOption(k, v)
        nlp.addOption("print_level", 0)
        x, info = nlp.solve(np.asarray(x0, float))
        return {"x": np.asarray(x), "f": float(info["obj_val"]), "g": np.asarray(info["g"]), "lam_g": np.asarray(info["mult_g"]),
                "lam_x": np.asarray(info["mult_x_U"]) - np.asarray(info["mult_x_L"]), "status": info["status"],
                "status_msg": info["status_msg"]}

    def solve_nlp(self):
        it = self.current_iteration
        t0 = time.perf_counter()
        out = self.nlp_solver(x0=it.guess_x, lbx=it.x_bnd_l, ubx=it.x_bnd_u, lbg=it.c_bnd_l, ubg=it.c_bnd_u)
        return NlpResult(solution=out, info=None, solve_time=time.perf_counter() - t0)

    # ---- probes (backend.py:1713-1805) ---------------------------------------------------------------
    def evaluate_J(self, x):
        return float(self.engine.evaluate_J(x))

    def evaluate_g(self, x):
        return np.asarray(self.engine.evaluate_g(x)).squeeze()

    def evaluate_c(self, x):
        return np.asarray(self.engine.evaluate_c(x)).squeeze()

    def evaluate_G(self, x):
        return self.engine.evaluate_G(x)           # scipy COO, as backend.py:1727-1736

    def evaluate_G_nonzeros(self, x):
        """In CasADi's column-major (CCS) order, which is what backend.py:1738-1745 returns."""
        return self.engine.evaluate_G_nonzeros(x)[self._ccs()]

    def evaluate_G_structure(self):
        """(row_indices, col_indices) in CCS order (backend.py:1747-1761)."""
        r, c = self.engine.evaluate_G_structure()
        p = self._ccs()
        return r[p], c[p]

    def evaluate_G_num_nonzero(self):
        return self.engine.evaluate_G_num_nonzero()

    def evaluate_H(self, x, obj, l):
        return self.engine.evaluate_H(x, obj, l)

    def evaluate_H_nonzeros(self, x, obj=1.0, l=None):
        l = np.zeros(self.engine.num_c) if l is None else l
        return self.engine.evaluate_H_nonzeros(x, obj, l)

    def evaluate_H_structure(self):
        return self.engine.evaluate_H_structure()

    def evaluate_H_num_nonzero(self):
        return self.engine.evaluate_H_num_nonzero()

    def _ccs(self):
        if getattr(self, "_ccs_perm_engine", None) is not self.engine:
            self._ccs_perm, self._ccs_perm_engine = self.engine.csr_to_ccs_permutation(), self.engine
        return self._ccs_perm

    # ---- scaling-time callables ----------------------------------------------------------------------
    def _with_scaling(self, W_ocp, w_J, fn):
        """Evaluate ``fn`` with the iteration's V, r and the given W, w_J (the reference passes them as trailing
        arguments of the callable), then restore the engine's own."""
        e = self.engine
        s = self.current_iteration.scaling
        keep = (e.V_ocp.copy(), e.r_ocp.copy(), e.W_ocp.copy(), e.w_J)
        e.set_scaling(np.asarray(s.V_ocp, float), np.asarray(s.r_ocp, float), W_ocp, w_J)
        try:
            return fn()
        finally:
            e.set_scaling(*keep)

    def _g_scale(self, args):
        """``g_iter_scale_callable([x; w_J])`` (backend.py:1506-1511; caller scaling.py:360-363)."""
        args = np.asarray(args, float).reshape(-1)
        x, w = args[:self.engine.num_x], float(args[self.engine.num_x])
        return self._with_scaling(self.engine.W_ocp, w, lambda: self.engine.evaluate_g(x))

    def constraint_row_norms(self, x, W_ocp=None):
        """2-norm of every row of G at x with constraint scaling ``W_ocp`` (default 1): what
        ``IterationScaling._calculate_constraint_scaling`` computes by densifying G (scaling.py:392-395), from the
        GPU's CSR values (``pc_row_norms_jac``)."""
        W = np.ones(self.engine.layout.num_ocp_c) if W_ocp is None else np.asarray(W_ocp, float)
        return self._with_scaling(W, self.engine.w_J, lambda: self.engine.G_row_norms(x))

    def _G_scale(self, args):
        """``G_iter_scale_callable([x; W])`` (backend.py:1674-1679; caller scaling.py:392-394) returns a DENSE matrix in
        the reference, so this one does too -- up to 4 M entries; past that use :meth:`constraint_row_norms`."""
        args = np.asarray(args, float).reshape(-1)
        n = self.engine.num_x
        x, W = args[:n], args[n:]
        if self.engine.num_c * n > 4_000_000:
            raise MemoryError("G_iter_scale_callable densifies G; use Mi355x.constraint_row_norms(x) at this size")
        return self._with_scaling(W, self.engine.w_J, lambda: self.engine.evaluate_G(x).toarray())

    def _dy(self, x):
        """``dy_iter_callable(x)`` (backend.py:1551-1556, 1666-1668): the state derivatives f(y, u, q, t, s) at every
        node of every phase, phase after phase, state after state -- solution post-processing
        (solution/casadi_solution.py:71), evaluated on the host from the model's expressions."""
        e = self.engine
        x = np.asarray(x, float).reshape(-1)
        out = []
        fns = getattr(self, "_dy_fns", None)
        if fns is None or getattr(self, "_dy_model", None) is not e.model:
            fns = []
            for pm in e.model.phases:
                consts = dict(pm.consts)
                args = list(pm.z) + list(pm.s)
                fns.append([sym.lambdify(args, sym.sympify(f).xreplace(consts), "numpy") for f in pm.f])
            self._dy_fns, self._dy_model = fns, e.model
        lay = e.layout
        s_vals = e.V_ocp[lay.ocp_s_off:] * x[lay.s_off:] + e.r_ocp[lay.ocp_s_off:]
        for pm, pl, fn_list in zip(e.model.phases, lay.phases, fns):
            V, r = e.V_ocp[pl.ocp_x_off:], e.r_ocp[pl.ocp_x_off:]
            z = [V[b] * x[pl.x_off + b * pl.N:pl.x_off + (b + 1) * pl.N] + r[b] for b in range(pl.n_z)]
            w = []
            kinds = pm.w_kind or [0] * pm.n_s
            idxs = pm.w_idx or list(range(pm.n_s))
            for kind, idx in zip(kinds, idxs):
                if kind == 0:
                    w.append(s_vals[idx])
                else:
                    o = pl.n_z + (idx if kind == 1 else pl.n_q + idx)
                    xo = (pl.q_off + idx) if kind == 1 else (pl.t_off + idx)
                    w.append(V[o] * x[xo] + r[o])
            for fn in fn_list:
                out.append(np.broadcast_to(np.asarray(fn(*z, *w), float), (pl.N,)))
        return np.concatenate(out) if out else np.zeros(0)
