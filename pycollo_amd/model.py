"""Symbolic model compiler: problem description -> per-node and endpoint derivative tapes.

Plays the role of the reference's (dead) hSAD differentiator + numbafy code generators
(pycollo/expression_graph.py, pycollo/numbafy.py, pycollo/numbafy_hessian.py) and of the live
``ca.jacobian`` call (pycollo/backend.py:1674-1679): it produces, for every phase,

* the continuous functions ``F = [y_eqn | p_con | q_fnc]`` in canonical symbols
  (``z0..`` = needed states then needed controls, ``s0..`` = needed static parameters),
* their structurally non-zero first partials w.r.t. ``v = [z | s]`` (dense ``dc_dx`` of
  pycollo/compiled.py:230, kept sparse here),
* the lower-triangular Hessian of the *node Lagrangian*
  ``l = sum_a mf_a f_a + sum_m mp_m p_m + sum_m mg_m g_m`` w.r.t. ``v`` -- the quantity
  ``numbafy_continuous_hessian`` (numbafy_hessian.py:75-165) evaluates per node,

and for the endpoint functions (objective ``J``, endpoint constraints ``b``) value, gradient and
Hessian w.r.t. the point variables (pycollo/backend.py:1439-1446, compiled.py:61-121,381-403,479-482).

Derivatives are taken w.r.t. *unscaled* variables; the scaling chain rule (x = V x~ + r,
pycollo/backend.py:263-280) is applied by the kernels.  Dynamics, path constraints and integrands may depend on
states, controls, static parameters and -- as in the live reference, which substitutes y and u per node and leaves
q, t0, tF, s global (pycollo/backend.py:1526-1539,1565-1570) -- on the phase's integral variables and its free
initial / final time variables.  (The reference's dead explicit Jacobian has no d(zeta,gamma,rho)/d(q,t) blocks besides
the stretch terms, compiled.py:254,263-264,324-328,371-374.)  Everything that is not a node variable is a *parameter*
of the node functions: ``s`` lists them in x order, [used q | used free t | static parameters].
"""
from __future__ import annotations

import hashlib
from dataclasses import dataclass, field

import numpy as np
import sympy as sym

from . import problem as _pb

MAX_AUX_DEPTH = 100  # pycollo/backend.py:67


# ------------------------------------------------------------------------------------------------
@dataclass
class PhaseModel:
    index: int
    name: str
    n_y: int
    n_u: int
    n_q: int
    n_p: int
    n_s: int
    t_free: tuple[bool, bool]
    t_fixed: tuple[float, float]
    z: list[sym.Symbol]
    s: list[sym.Symbol]          # parameters of the node functions: [used q | used free t | static parameters]
    f: list[sym.Expr]
    p: list[sym.Expr]
    g: list[sym.Expr]
    # first partials: (function row in [f|p|g], variable col in [z|s], expression)
    jac: list[tuple[int, int, sym.Expr]]
    # node-Lagrangian multipliers and lower-triangular Hessian in v=[z|s]: (row>=col, col, expr)
    mf: list[sym.Symbol]
    mp: list[sym.Symbol]
    mg: list[sym.Symbol]
    hess: list[tuple[int, int, sym.Expr]]
    # named numeric constants (auxiliary data / eliminated variables) kept symbolic so that SymPy
    # never folds them into the expressions (exp(R_E/h_0) style overflow, evaluation-order drift)
    consts: list[tuple[sym.Symbol, float]] = field(default_factory=list)
    # bounds of the needed variables in x order: y, u, q, t (for scaling)
    x_bounds: list[tuple[float, float]] = field(default_factory=list)
    p_bounds: list[tuple[float, float]] = field(default_factory=list)
    y_t0_bounds: list[tuple[float, float]] = field(default_factory=list)
    y_tF_bounds: list[tuple[float, float]] = field(default_factory=list)
    # what each parameter is: kind 0 = static parameter (idx = its global index), 1 = integral variable (idx = m),
    # 2 = free time variable (idx = position among the phase's free times); empty = all static
    w_kind: list[int] = field(default_factory=list)
    w_idx: list[int] = field(default_factory=list)

    @property
    def n_wq(self) -> int:
        return sum(1 for k in self.w_kind if k == 1)

    @property
    def n_wt(self) -> int:
        return sum(1 for k in self.w_kind if k == 2)

    def param_index(self, kind: int, idx: int) -> int:
        """Position in ``s`` of the parameter (kind, idx), or -1 when the node functions do not depend on it."""
        if not self.w_kind:
            return idx if kind == 0 else -1
        for l, (k, i) in enumerate(zip(self.w_kind, self.w_idx)):
            if k == kind and i == idx:
                return l
        return -1

    def t_strip_mask(self) -> list[bool]:
        """Per node variable b: does the Hessian carry a (t, z_b) strip?  Through the stretch factor when a state
        equation or an integrand depends on z_b; through a second partial when any row depends on (t, z_b)."""
        jm, hm = self.jac_mask(), self.hess_mask()
        out = []
        for b in range(self.n_z):
            on = any(jm[r, b] for r in range(self.n_fn) if not (self.n_y <= r < self.n_y + self.n_p))
            for l, k in enumerate(self.w_kind):
                on = on or (k == 2 and bool(hm[self.n_z + l, b]))
            out.append(bool(on))
        return out

    @property
    def eval_ops(self) -> int:
        """Arithmetic operations in the node functions and their derivatives (before common-subexpression
        elimination): a size measure of the model, used as a launch-shape hint only."""
        cached = self.__dict__.get("_eval_ops")
        if cached is None:
            exprs = list(self.f) + list(self.p) + list(self.g) + [e for _, _, e in self.jac] + [e for _, _, e in self.hess]
            cached = self.__dict__["_eval_ops"] = int(sum(int(sym.count_ops(e)) for e in exprs))
        return cached

    @property
    def n_z(self) -> int:
        return self.n_y + self.n_u

    @property
    def n_t(self) -> int:
        return int(self.t_free[0]) + int(self.t_free[1])

    @property
    def n_fn(self) -> int:
        return self.n_y + self.n_p + self.n_q

    @property
    def n_v(self) -> int:
        return self.n_z + self.n_s

    def jac_mask(self) -> np.ndarray:
        m = np.zeros((self.n_fn, self.n_v), dtype=np.uint8)
        for r, c, _ in self.jac:
            m[r, c] = 1
        return m

    def hess_mask(self) -> np.ndarray:
        m = np.zeros((self.n_v, self.n_v), dtype=np.uint8)
        for r, c, _ in self.hess:
            m[r, c] = 1
        return m


@dataclass
class PointVar:
    symbol: sym.Symbol
    phase: int          # -1 for static parameters
    kind: str           # "y0", "yF", "q", "t0", "tF", "s"
    idx: int            # index within its kind (state index / integral index / parameter index)


@dataclass
class PointModel:
    """Endpoint functions in *unscaled* point variables ``xb`` (ordered as x_point_var,
    pycollo/backend.py:658-661,1264-1269: per phase y0(t0), y0(tF), y1(t0), ... then q, t; then s)."""
    vars: list[PointVar]
    J: sym.Expr
    b: list[sym.Expr]
    J_grad: list[tuple[int, sym.Expr]]
    b_jac: list[tuple[int, int, sym.Expr]]
    sigma: sym.Symbol                      # objective factor (already multiplied by w_J by caller)
    lam: list[sym.Symbol]                  # endpoint multipliers (already multiplied by W_e)
    hess: list[tuple[int, int, sym.Expr]]  # lower triangle in xb index of sigma*J + lam.b
    consts: list[tuple[sym.Symbol, float]] = field(default_factory=list)
    b_bounds: list[tuple[float, float]] = field(default_factory=list)


@dataclass
class Model:
    name: str
    phases: list[PhaseModel]
    point: PointModel
    n_s: int
    s_bounds: list[tuple[float, float]]
    scaling_method: str | None
    quadrature_method: str
    digest: str = ""


# ------------------------------------------------------------------------------------------------
def _resolve_aux(aux: dict, primitives: set[sym.Symbol]) -> dict:
    """Fully substitute auxiliary data so every value is an expression of primitives/numbers."""
    aux = {k: sym.sympify(v) for k, v in aux.items()}
    resolved: dict = {}
    pending = dict(aux)
    for _ in range(MAX_AUX_DEPTH):
        progressed = False
        for k in list(pending):
            e = pending[k].xreplace(resolved)
            if not (e.free_symbols & set(pending)):
                resolved[k] = e
                del pending[k]
                progressed = True
            else:
                pending[k] = e
        if not pending:
            break
        if not progressed:
            raise ValueError(f"auxiliary data has a cyclic definition among {sorted(map(str, pending))}")
    if pending:
        raise ValueError("auxiliary data substitution exceeded the maximum depth")
    return resolved


class _ConstPool:
    """Maps numeric auxiliary data / eliminated variables to canonical constant symbols K<i>.

    0 and +-1 are substituted as numbers (CasADi's SX constructors simplify x*0, x*1 and x+0 the same
    way, so they never reach the structural pattern); every other value stays a named constant."""

    def __init__(self, prefix: str):
        self.prefix = prefix
        self.table: list[tuple[sym.Symbol, float]] = []
        self._by_key: dict = {}

    def get(self, name: str, value: float):
        value = float(value)
        if value in (0.0, 1.0, -1.0):
            return sym.Integer(int(value))
        key = (name, value)
        k = self._by_key.get(key)
        if k is None:
            k = sym.Symbol(f"{self.prefix}{len(self.table)}", real=True)
            self.table.append((k, value))
            self._by_key[key] = k
        return k


def _split_aux(aux: dict, pool: _ConstPool) -> tuple[dict, dict]:
    """(expression aux, constant map) -- numeric leaves become pooled constant symbols."""
    expr_aux, const_map = {}, {}
    for k, v in aux.items():
        v = sym.sympify(v)
        if v.is_number:
            const_map[k] = pool.get(str(k), float(v))
        else:
            expr_aux[k] = v
    return expr_aux, const_map


def _check_free(expr: sym.Expr, allowed: set, what: str):
    extra = expr.free_symbols - allowed
    if extra:
        raise ValueError(f"{what} contains symbols that are neither variables nor auxiliary data: "
                         f"{sorted(map(str, extra))}")


def _nz(e: sym.Expr) -> bool:
    return e != 0


_MODEL_CACHE: dict = {}    # per process: everything below depends on the equations and bounds, never on the mesh


def _problem_key(prob: _pb.ProblemSpec):
    """What ``compile_model`` reads of a problem, cheaply: structural hashes of the expressions (SymPy caches them per
    object) and the text of the bounds.  A mesh iteration re-creates the engine for the same equations on a new mesh;
    differentiating, hashing and counting a heavy model again each time was most of its host time (space station:
    4 s of symbolic work per mesh iteration, twice, for a 0.2 s NLP solve)."""
    def ex(seq):
        return tuple(hash(sym.sympify(e)) for e in (seq or ()))

    def aux(d):
        return tuple(sorted((str(k), hash(sym.sympify(v))) for k, v in (d or {}).items()))

    def bnd(b):
        return tuple((k, repr(v)) for k, v in sorted(vars(b).items()))

    key = [prob.name, prob.scaling_method, prob.quadrature_method, ex(prob.parameter_variables), aux(prob.auxiliary_data),
           hash(sym.sympify(prob.objective_function)), ex(prob.endpoint_constraints), bnd(prob.bounds)]
    for ph in prob.phases:
        key += [ph.name, ph.i, ex(ph.state_variables), ex(ph.control_variables), ex(ph.state_equations), ex(ph.path_constraints),
                ex(ph.integrand_functions), aux(ph.auxiliary_data), bnd(ph.bounds)]
    return tuple(key)


def compile_model(prob: _pb.ProblemSpec) -> Model:
    if not prob.phases:
        raise ValueError("a problem needs at least one phase")
    if prob.objective_function is None:
        raise ValueError("objective_function is required")
    try:
        key = _problem_key(prob)
    except Exception:           # an unhashable oddity in a user's problem: compile without the cache
        key = None
    if key is not None and key in _MODEL_CACHE:
        return _MODEL_CACHE[key]
    model = _compile_model(prob)
    if key is not None:
        _MODEL_CACHE[key] = model
    return model


def _compile_model(prob: _pb.ProblemSpec) -> Model:

    # ---- static parameters --------------------------------------------------------------------
    s_user = list(prob.parameter_variables)
    s_bnds_all = _pb._bounds_for(s_user, prob.bounds.parameter_variables, "parameter variable")
    s_need = _pb.needed(s_bnds_all)
    s_const = {s: 0.5 * (lo + hi) for s, (lo, hi), nd in zip(s_user, s_bnds_all, s_need) if not nd}
    s_used = [s for s, nd in zip(s_user, s_need) if nd]
    s_bounds = [b for b, nd in zip(s_bnds_all, s_need) if nd]
    s_canon = [sym.Symbol(f"s{i}", real=True) for i in range(len(s_used))]
    s_map = dict(zip(s_used, s_canon))

    phases: list[PhaseModel] = []
    point_vars: list[PointVar] = []
    point_subs: dict = {}          # user point symbol -> canonical point symbol or constant

    for ph in prob.phases:
        y_b, u_b, q_b = _pb.phase_variable_bounds(ph)
        (t0_b, tF_b) = _pb.phase_time_bounds(ph)
        y_need, u_need, q_need = _pb.needed(y_b), _pb.needed(u_b), _pb.needed(q_b)
        t_need = _pb.needed([t0_b, tF_b])
        if len(ph.state_equations) != len(ph.state_variables):
            raise ValueError(f"phase {ph.name}: one state equation per state variable is required")

        y_used = [y for y, nd in zip(ph._y, y_need) if nd]
        u_used = [u for u, nd in zip(ph._u, u_need) if nd]
        pool = _ConstPool("K")
        consts = {s: pool.get(str(s), val) for s, val in s_const.items()}
        consts.update({y: pool.get(str(y), 0.5 * (lo + hi)) for y, (lo, hi), nd in zip(ph._y, y_b, y_need) if not nd})
        consts.update({u: pool.get(str(u), 0.5 * (lo + hi)) for u, (lo, hi), nd in zip(ph._u, u_b, u_need) if not nd})
        z_user = y_used + u_used
        z_canon = [sym.Symbol(f"z{i}", real=True) for i in range(len(z_user))]
        var_map = dict(zip(z_user, z_canon))
        var_map.update(s_map)
        # q, t0, tF inside f / p / g: needed ones become parameters, eliminated ones constants
        q_user = list(ph.integral_variables)
        t_user = [ph.initial_time_variable, ph.final_time_variable]
        consts.update({q_: pool.get(str(q_), 0.5 * (lo + hi)) for q_, (lo, hi), nd in zip(q_user, q_b, q_need) if not nd})
        consts.update({t_: pool.get(str(t_), 0.5 * (lo + hi)) for t_, (lo, hi), nd in zip(t_user, (t0_b, tF_b), t_need) if not nd})
        q_kept = [q_ for q_, nd in zip(q_user, q_need) if nd]
        t_kept = [t_ for t_, nd in zip(t_user, t_need) if nd]
        qt_canon = {q_: sym.Symbol(f"wq{m}", real=True) for m, q_ in enumerate(q_kept)}
        qt_canon.update({t_: sym.Symbol(f"wt{j}", real=True) for j, t_ in enumerate(t_kept)})
        var_map.update(qt_canon)
        primitives = set(ph._y) | set(ph._u) | set(s_user) | set(q_user) | set(t_user)

        aux = dict(prob.auxiliary_data)
        aux.update(ph.auxiliary_data)        # phase data overrides problem data
        clash = set(aux) & primitives
        if clash:
            raise ValueError(f"auxiliary data redefines variables {sorted(map(str, clash))}")
        aux_expr, const_map = _split_aux(aux, pool)
        aux_res = _resolve_aux(aux_expr, primitives)

        def lower(e, what):
            e = sym.sympify(e).xreplace(aux_res).xreplace(const_map).xreplace(consts)
            _check_free(e, primitives | {k for k, _ in pool.table}, what)
            return e.xreplace(var_map)

        # equations of eliminated states are dropped together with the state (backend keeps
        # y_eqn for needed states only)
        f = [lower(e, f"state equation {i}") for i, (e, nd) in enumerate(zip(ph.state_equations, y_need)) if nd]
        p = [lower(e, f"path constraint {i}") for i, e in enumerate(ph.path_constraints)]
        g_all = [lower(e, f"integrand {i}") for i, e in enumerate(ph.integrand_functions)]
        g = [e for e, nd in zip(g_all, q_need) if nd]
        F = f + p + g
        used = set().union(*[e.free_symbols for e in F]) if F else set()
        w_syms, w_kind, w_idx = [], [], []
        for m, q_ in enumerate(q_kept):
            if qt_canon[q_] in used:
                w_syms.append(qt_canon[q_]); w_kind.append(1); w_idx.append(m)
        for j, t_ in enumerate(t_kept):
            if qt_canon[t_] in used:
                w_syms.append(qt_canon[t_]); w_kind.append(2); w_idx.append(j)
        if w_syms:
            w_kind += [0] * len(s_canon); w_idx += list(range(len(s_canon)))
        w_syms += s_canon
        n_y, n_u, n_q, n_p, n_s = len(f), len(u_used), len(g), len(p), len(w_syms)
        v = z_canon + w_syms

        jac = []
        for r, e in enumerate(F):
            fs = e.free_symbols
            for c, var in enumerate(v):
                if var in fs:
                    d = sym.diff(e, var)
                    if _nz(d):
                        jac.append((r, c, d))

        mf = [sym.Symbol(f"mf{i}", real=True) for i in range(n_y)]
        mp = [sym.Symbol(f"mp{i}", real=True) for i in range(n_p)]
        mg = [sym.Symbol(f"mg{i}", real=True) for i in range(n_q)]
        mult = mf + mp + mg
        # second partials per function, combined symbolically with the multipliers
        first = {(r, c): d for r, c, d in jac}
        hess_acc: dict[tuple[int, int], sym.Expr] = {}
        for (r, c), d in first.items():
            fs = d.free_symbols
            for c2 in range(c + 1):                      # lower triangle: row c >= col c2
                var2 = v[c2]
                if var2 in fs:
                    d2 = sym.diff(d, var2)
                    if _nz(d2):
                        key = (c, c2)
                        hess_acc[key] = hess_acc.get(key, 0) + mult[r] * d2
        hess = [(r, c, e) for (r, c), e in sorted(hess_acc.items()) if _nz(e)]

        x_bounds = ([b for b, nd in zip(y_b, y_need) if nd] + [b for b, nd in zip(u_b, u_need) if nd]
                    + [b for b, nd in zip(q_b, q_need) if nd]
                    + [b for b, nd in zip((t0_b, tF_b), t_need) if nd])
        p_bounds = _pb._bounds_for(list(range(n_p)), ph.bounds.path_constraints, "path constraint") if n_p else []

        def endpoint_bounds(spec, default):
            if spec is None:
                return list(default)
            if isinstance(spec, dict):
                return [_pb._pair(spec[y]) if y in spec else d for y, d in zip(ph._y, default)]
            return [_pb._pair(b) for b in spec]

        y0_b = endpoint_bounds(ph.bounds.initial_state_constraints, y_b)
        yF_b = endpoint_bounds(ph.bounds.final_state_constraints, y_b)

        pm = PhaseModel(index=ph.i, name=ph.name, n_y=n_y, n_u=n_u, n_q=n_q, n_p=n_p, n_s=n_s,
                        t_free=(bool(t_need[0]), bool(t_need[1])),
                        t_fixed=(0.5 * sum(t0_b), 0.5 * sum(tF_b)),
                        z=z_canon, s=w_syms, f=f, p=p, g=g, jac=jac, mf=mf, mp=mp, mg=mg, hess=hess,
                        w_kind=w_kind, w_idx=w_idx,
                        consts=list(pool.table), x_bounds=x_bounds, p_bounds=p_bounds,
                        y_t0_bounds=[b for b, nd in zip(y0_b, y_need) if nd],
                        y_tF_bounds=[b for b, nd in zip(yF_b, y_need) if nd])
        phases.append(pm)

        # ---- point variables of this phase (x_point_var order) --------------------------------
        k = 0
        for i, (y, nd) in enumerate(zip(ph._y, y_need)):
            y0, yF = ph.initial_state_variables[i], ph.final_state_variables[i]
            if nd:
                for kind, usym in (("y0", y0), ("yF", yF)):
                    cs = sym.Symbol(f"xb{len(point_vars)}", real=True)
                    point_vars.append(PointVar(cs, ph.i, kind, k))
                    point_subs[usym] = cs
                k += 1
            else:
                point_subs[y0] = point_subs[yF] = 0.5 * sum(y_b[i])
        k = 0
        for i, nd in enumerate(q_need):
            usym = ph.integral_variables[i]
            if nd:
                cs = sym.Symbol(f"xb{len(point_vars)}", real=True)
                point_vars.append(PointVar(cs, ph.i, "q", k))
                point_subs[usym] = cs
                k += 1
            else:
                point_subs[usym] = 0.5 * sum(q_b[i])
        for kind, usym, nd, bnd in (("t0", ph.initial_time_variable, t_need[0], t0_b),
                                    ("tF", ph.final_time_variable, t_need[1], tF_b)):
            if nd:
                cs = sym.Symbol(f"xb{len(point_vars)}", real=True)
                point_vars.append(PointVar(cs, ph.i, kind, 0))
                point_subs[usym] = cs
            else:
                point_subs[usym] = 0.5 * sum(bnd)

    for i, (s, cs) in enumerate(zip(s_used, s_canon)):
        pv = sym.Symbol(f"xb{len(point_vars)}", real=True)
        point_vars.append(PointVar(pv, -1, "s", i))
        point_subs[s] = pv
    point_subs.update(s_const)

    # ---- endpoint functions ---------------------------------------------------------------------
    ppool = _ConstPool("K")
    point_subs = {k: (ppool.get(str(k), v) if isinstance(v, (int, float)) else v) for k, v in point_subs.items()}
    aux_pt_expr, pt_const_map = _split_aux(dict(prob.auxiliary_data), ppool)
    aux_pt = _resolve_aux(aux_pt_expr, set(point_subs))
    xb = [pv.symbol for pv in point_vars]
    xb_set = set(xb)

    def lower_pt(e, what):
        e = sym.sympify(e).xreplace(aux_pt).xreplace(pt_const_map).xreplace(point_subs)
        _check_free(e, xb_set | {k for k, _ in ppool.table}, what)
        return e

    J = lower_pt(prob.objective_function, "objective function")
    b = [lower_pt(e, f"endpoint constraint {i}") for i, e in enumerate(prob.endpoint_constraints)]
    for i, e in enumerate(b):
        if e in xb_set:
            # pycollo/backend.py:764-770
            raise ValueError(f"endpoint constraint {i} is a bare point variable; use state endpoint "
                             f"bounds instead")
    J_grad = [(c, sym.diff(J, x)) for c, x in enumerate(xb) if x in J.free_symbols]
    J_grad = [(c, d) for c, d in J_grad if _nz(d)]
    b_jac = []
    for r, e in enumerate(b):
        for c, x in enumerate(xb):
            if x in e.free_symbols:
                d = sym.diff(e, x)
                if _nz(d):
                    b_jac.append((r, c, d))
    sigma = sym.Symbol("sigma", real=True)
    lam = [sym.Symbol(f"lb{i}", real=True) for i in range(len(b))]
    hacc: dict[tuple[int, int], sym.Expr] = {}

    def add_h(weight, grads):
        for c, d in grads:
            for c2 in range(c + 1):
                if xb[c2] in d.free_symbols:
                    d2 = sym.diff(d, xb[c2])
                    if _nz(d2):
                        hacc[(c, c2)] = hacc.get((c, c2), 0) + weight * d2

    add_h(sigma, J_grad)
    for r in range(len(b)):
        add_h(lam[r], [(c, d) for rr, c, d in b_jac if rr == r])
    pt_hess = [(r, c, e) for (r, c), e in sorted(hacc.items()) if _nz(e)]
    b_bounds = _pb._bounds_for(list(range(len(b))), prob.bounds.endpoint_constraints, "endpoint constraint") if b else []

    point = PointModel(vars=point_vars, J=J, b=b, J_grad=J_grad, b_jac=b_jac, sigma=sigma, lam=lam,
                       hess=pt_hess, consts=list(ppool.table), b_bounds=b_bounds)
    model = Model(name=prob.name, phases=phases, point=point, n_s=len(s_canon), s_bounds=s_bounds,
                  scaling_method=prob.scaling_method, quadrature_method=prob.quadrature_method)
    model.digest = model_digest(model)
    return model


def model_digest(model: Model) -> str:
    """Stable hash of everything the generated kernels depend on (cache key)."""
    h = hashlib.sha256()

    def put(*items):
        for it in items:
            h.update(repr(it).encode())
            h.update(b"|")

    for pm in model.phases:
        put("phase", pm.n_y, pm.n_u, pm.n_q, pm.n_p, pm.n_s, pm.t_free, [(str(k), v) for k, v in pm.consts])
        if pm.w_kind:
            put("params", pm.w_kind, pm.w_idx)
        put([sym.srepr(e) for e in pm.f + pm.p + pm.g])
        put([(r, c, sym.srepr(e)) for r, c, e in pm.jac])
        put([(r, c, sym.srepr(e)) for r, c, e in pm.hess])
    pt = model.point
    put("point", [(str(k), v) for k, v in pt.consts], [(v.phase, v.kind, v.idx) for v in pt.vars], sym.srepr(pt.J), [sym.srepr(e) for e in pt.b])
    put([(c, sym.srepr(e)) for c, e in pt.J_grad], [(r, c, sym.srepr(e)) for r, c, e in pt.b_jac])
    put([(r, c, sym.srepr(e)) for r, c, e in pt.hess])
    return h.hexdigest()[:16]
