"""Minimal problem description consumed by the callback engine.

This is *not* pycollo's front end (``OptimalControlProblem`` / ``Phase`` / ``Bounds`` validation is
out of scope).  It holds exactly what the hot path needs from the front end's
output: ordered variable symbols, per-phase equation lists, auxiliary data, the bounds that drive
"bounds" scaling (pycollo/scaling.py:87-92) and constant-variable elimination
(pycollo/bounds.py:456-480), and the mesh description.  Attribute names follow the reference so the
benchmark problem definitions under ``pycollo_amd/problems`` read like the reference's examples.
"""
from __future__ import annotations

from typing import Iterable, Sequence

import numpy as np
import sympy as sym

INF = float("inf")
BOUND_CLASH_TOLERANCE = 1e-6  # pycollo/settings.py DEFAULT_BOUND_CLASH_{ABSOLUTE,RELATIVE}_TOLERANCE


def _as_list(x) -> list:
    if x is None:
        return []
    if isinstance(x, (sym.Basic, int, float)):
        return [x]
    return list(x)


def _pair(b) -> tuple[float, float]:
    """Normalise one bound to (lower, upper); a scalar means an equality bound."""
    if isinstance(b, (int, float, np.integer, np.floating, sym.Basic)):
        v = float(b)
        return (v, v)
    b = list(b)
    if len(b) == 1:
        return (float(b[0]), float(b[0]))
    lo, hi = b
    return (float(lo), float(hi))  # float() accepts "inf" / "-inf" strings


def _bounds_for(symbols: Sequence[sym.Symbol], spec, what: str) -> list[tuple[float, float]]:
    """Accept a dict keyed by symbol or a sequence ordered like ``symbols``."""
    if not symbols:
        return []
    if spec is None:
        raise ValueError(f"bounds for {what} are required")
    if isinstance(spec, dict):
        missing = [s for s in symbols if s not in spec]
        if missing:
            raise ValueError(f"missing {what} bounds for {missing}")
        return [_pair(spec[s]) for s in symbols]
    spec = list(spec)
    if len(symbols) == 1 and len(spec) == 2 and not isinstance(spec[0], (list, tuple, np.ndarray)):
        return [_pair(spec)]
    if len(spec) != len(symbols):
        raise ValueError(f"expected {len(symbols)} {what} bounds, got {len(spec)}")
    return [_pair(b) for b in spec]


class PhaseBounds:
    def __init__(self):
        self.initial_time = None
        self.final_time = None
        self.state_variables = None
        self.control_variables = None
        self.integral_variables = None
        self.path_constraints = None
        self.initial_state_constraints = None
        self.final_state_constraints = None


class PhaseGuess:
    """User guess of one phase (pycollo/guess.py:80-107): values at the time points ``time``."""

    def __init__(self):
        self.time = None
        self.state_variables = None
        self.control_variables = None
        self.integral_variables = None


class EndpointGuess:
    def __init__(self):
        self.parameter_variables = None


class PhaseMeshSpec:
    """User mesh description (pycollo/mesh.py:10-107): K, section fractions, nodes per section."""

    def __init__(self):
        self.number_mesh_sections = 10
        self.mesh_section_sizes = None
        self.number_mesh_section_nodes = 4

    def resolved(self) -> tuple[np.ndarray, np.ndarray]:
        K = int(self.number_mesh_sections)
        sizes = self.mesh_section_sizes
        if sizes is None:
            sizes = np.ones(K) / K
        sizes = np.asarray(sizes, dtype=float)
        sizes = sizes / sizes.sum()
        nodes = self.number_mesh_section_nodes
        if np.ndim(nodes) == 0:
            nodes = np.full(K, int(nodes), dtype=np.int64)
        nodes = np.asarray(nodes, dtype=np.int64)
        if sizes.shape[0] != K or nodes.shape[0] != K:
            raise ValueError("mesh description is inconsistent with number_mesh_sections")
        return sizes, nodes


class Phase:
    """One phase: variables, equations, auxiliary data, bounds, mesh."""

    def __init__(self, problem: "ProblemSpec", name: str, index: int):
        self.problem = problem
        self.name = name
        self.i = index
        self._y: list[sym.Symbol] = []
        self._u: list[sym.Symbol] = []
        self._y_eqn: list[sym.Expr] = []
        self.path_constraints: list[sym.Expr] = []
        self._q_fnc: list[sym.Expr] = []
        self.auxiliary_data: dict = {}
        self.bounds = PhaseBounds()
        self.guess = PhaseGuess()
        self.mesh = PhaseMeshSpec()
        self.initial_state_variables: tuple = ()
        self.final_state_variables: tuple = ()
        self.integral_variables: tuple = ()
        self.initial_time_variable = sym.Symbol(f"_t0_P{index}")
        self.final_time_variable = sym.Symbol(f"_tF_P{index}")

    # -- variables -------------------------------------------------------------------------
    @property
    def state_variables(self):
        return tuple(self._y)

    @state_variables.setter
    def state_variables(self, ys):
        self._y = _as_list(ys)
        self.initial_state_variables = tuple(sym.Symbol(f"{y.name}_P{self.i}(t0)") for y in self._y)
        self.final_state_variables = tuple(sym.Symbol(f"{y.name}_P{self.i}(tF)") for y in self._y)

    @property
    def control_variables(self):
        return tuple(self._u)

    @control_variables.setter
    def control_variables(self, us):
        self._u = _as_list(us)

    @property
    def state_equations(self):
        return tuple(self._y_eqn)

    @state_equations.setter
    def state_equations(self, eqns):
        if isinstance(eqns, dict):
            eqns = [eqns[y] for y in self._y]
        self._y_eqn = [sym.sympify(e) for e in _as_list(eqns)]

    @property
    def integrand_functions(self):
        return tuple(self._q_fnc)

    @integrand_functions.setter
    def integrand_functions(self, fncs):
        self._q_fnc = [sym.sympify(e) for e in _as_list(fncs)]
        self.integral_variables = tuple(sym.Symbol(f"_q{i}_P{self.i}") for i in range(len(self._q_fnc)))


class EndpointBounds:
    def __init__(self):
        self.parameter_variables = None
        self.endpoint_constraints = None


class ProblemSpec:
    """Problem-level container: phases, static parameters, objective, endpoint constraints."""

    def __init__(self, name: str = "ocp"):
        self.name = name
        self.phases: list[Phase] = []
        self._s: list[sym.Symbol] = []
        self.objective_function: sym.Expr | None = None
        self.endpoint_constraints: list[sym.Expr] = []
        self.auxiliary_data: dict = {}
        self.bounds = EndpointBounds()
        self.guess = EndpointGuess()
        self.scaling_method: str | None = "bounds"
        self.quadrature_method: str = "lobatto"

    def new_phase(self, name: str) -> Phase:
        ph = Phase(self, name, len(self.phases))
        self.phases.append(ph)
        return ph

    @property
    def parameter_variables(self):
        return tuple(self._s)

    @parameter_variables.setter
    def parameter_variables(self, ss):
        self._s = _as_list(ss)


# ---- resolved numeric views used by the model compiler and the layout ---------------------------

def phase_time_bounds(ph: Phase) -> tuple[tuple[float, float], tuple[float, float]]:
    if ph.bounds.initial_time is None or ph.bounds.final_time is None:
        raise ValueError(f"phase {ph.name}: initial_time and final_time bounds are required")
    return _pair(ph.bounds.initial_time), _pair(ph.bounds.final_time)


def phase_variable_bounds(ph: Phase):
    y = _bounds_for(ph._y, ph.bounds.state_variables, "state variable")
    u = _bounds_for(ph._u, ph.bounds.control_variables, "control variable")
    q = _bounds_for(list(ph.integral_variables), ph.bounds.integral_variables, "integral variable")
    return y, u, q


def needed(bounds: Iterable[tuple[float, float]]) -> list[bool]:
    """A variable whose bounds coincide is a constant and leaves the NLP (bounds.py:456-480)."""
    return [not np.isclose(lo, hi, rtol=BOUND_CLASH_TOLERANCE, atol=BOUND_CLASH_TOLERANCE) for lo, hi in bounds]
