"""NLP variable / constraint layout for one mesh iteration (host bookkeeping, integers only).

Restates ``Iteration.create_variable_constraint_counts_slices`` (pycollo/iteration.py:196-314) and
the variable order of ``create_iteration_specific_variable_symbols`` (pycollo/backend.py:1433-1457):

    x~ = (+)_p [ y_0(tau_0..tau_{N-1}), ..., y_{ny-1}(.), u_0(.), ..., q, t(free) ]  (+)  s
    c~ = (+)_p [ defect(state 0) (N-1 rows), ..., path(0) (N rows), ..., integral ]  (+)  endpoint

plus the "bounds" variable scaling ``V = u - l``, ``r = u - (u - l)/2`` (pycollo/scaling.py:87-92) and
the base constraint scaling (scaling.py:106-115).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .mesh import PhaseMesh
from .model import Model


@dataclass
class PhaseLayout:
    N: int
    K: int
    n_y: int
    n_u: int
    n_q: int
    n_p: int
    n_t: int
    x_off: int          # first x index of the phase (= y block)
    q_off: int
    t_off: int
    ocp_x_off: int      # offset into the per-OCP-variable vectors (V_ocp, r_ocp)
    c_off: int          # first c row of the phase (= defect block)
    c_path_off: int
    c_int_off: int
    ocp_c_off: int      # offset into the per-OCP-constraint vector (W_ocp)

    @property
    def n_z(self):
        return self.n_y + self.n_u

    @property
    def num_x(self):
        return self.n_z * self.N + self.n_q + self.n_t

    @property
    def num_c(self):
        return self.n_y * (self.N - 1) + self.n_p * self.N + self.n_q


class NlpLayout:
    def __init__(self, model: Model, meshes: list[PhaseMesh]):
        if len(meshes) != len(model.phases):
            raise ValueError("one mesh per phase is required")
        self.model = model
        self.meshes = meshes
        self.phases: list[PhaseLayout] = []
        x = c = ox = oc = 0
        for pm, mesh in zip(model.phases, meshes):
            N = mesh.N
            pl = PhaseLayout(N=N, K=mesh.K, n_y=pm.n_y, n_u=pm.n_u, n_q=pm.n_q, n_p=pm.n_p, n_t=pm.n_t,
                             x_off=x, q_off=x + pm.n_z * N, t_off=x + pm.n_z * N + pm.n_q, ocp_x_off=ox,
                             c_off=c, c_path_off=c + pm.n_y * (N - 1),
                             c_int_off=c + pm.n_y * (N - 1) + pm.n_p * N, ocp_c_off=oc)
            self.phases.append(pl)
            x += pl.num_x
            c += pl.num_c
            ox += pm.n_z + pm.n_q + pm.n_t
            oc += pm.n_y + pm.n_p + pm.n_q
        self.n_s = model.n_s
        self.s_off = x
        self.ocp_s_off = ox
        self.num_x = x + self.n_s
        self.n_b = len(model.point.b)
        self.c_end_off = c
        self.ocp_c_end_off = oc
        self.num_c = c + self.n_b
        self.num_ocp_x = ox + self.n_s
        self.num_ocp_c = oc + self.n_b

    # ---- point variables -> x index (backend.py:1439-1446: y(t0)=y[0], y(tF)=y[N-1]) ------------
    def point_x_index(self) -> np.ndarray:
        idx = []
        for pv in self.model.point.vars:
            if pv.kind == "s":
                idx.append(self.s_off + pv.idx)
                continue
            pl = self.phases[pv.phase]
            pm = self.model.phases[pv.phase]
            if pv.kind == "y0":
                idx.append(pl.x_off + pv.idx * pl.N)
            elif pv.kind == "yF":
                idx.append(pl.x_off + pv.idx * pl.N + pl.N - 1)
            elif pv.kind == "q":
                idx.append(pl.q_off + pv.idx)
            elif pv.kind == "t0":
                idx.append(pl.t_off)
            elif pv.kind == "tF":
                idx.append(pl.t_off + (1 if pm.t_free[0] else 0))
        return np.asarray(idx, dtype=np.int64)

    def point_ocp_index(self) -> np.ndarray:
        """Index of every point variable into the per-OCP-variable vectors (V_ocp, r_ocp)."""
        idx = []
        for pv in self.model.point.vars:
            if pv.kind == "s":
                idx.append(self.ocp_s_off + pv.idx)
                continue
            pl = self.phases[pv.phase]
            pm = self.model.phases[pv.phase]
            if pv.kind in ("y0", "yF"):
                idx.append(pl.ocp_x_off + pv.idx)
            elif pv.kind == "q":
                idx.append(pl.ocp_x_off + pm.n_z + pv.idx)
            elif pv.kind == "t0":
                idx.append(pl.ocp_x_off + pm.n_z + pm.n_q)
            elif pv.kind == "tF":
                idx.append(pl.ocp_x_off + pm.n_z + pm.n_q + (1 if pm.t_free[0] else 0))
        return np.asarray(idx, dtype=np.int64)

    # ---- scaling ----------------------------------------------------------------------------------
    def ocp_bounds(self) -> np.ndarray:
        rows = []
        for pm in self.model.phases:
            rows.extend(pm.x_bounds)
        rows.extend(self.model.s_bounds)
        return np.asarray(rows, dtype=np.float64).reshape(-1, 2)

    def base_variable_scaling(self) -> tuple[np.ndarray, np.ndarray]:
        """(V_ocp, r_ocp) per OCP variable: scaling.py:87-101."""
        if self.model.scaling_method is None or self.model.scaling_method == "none":
            return np.ones(self.num_ocp_x), np.zeros(self.num_ocp_x)
        if self.model.scaling_method != "bounds":
            raise NotImplementedError(f"scaling method {self.model.scaling_method!r}")  # scaling.py:94-104
        b = self.ocp_bounds()
        lo, hi = b[:, 0], b[:, 1]
        if not np.all(np.isfinite(b)):
            raise ValueError("'bounds' scaling needs finite bounds on every variable")
        return hi - lo, hi - (hi - lo) / 2

    def base_constraint_scaling(self, V_ocp: np.ndarray) -> np.ndarray:
        """W_ocp with defect = 1/V_y, integral = 1/V_q, path/endpoint = 1 (scaling.py:421-426)."""
        W = np.ones(self.num_ocp_c)
        for pl, pm in zip(self.phases, self.model.phases):
            W[pl.ocp_c_off:pl.ocp_c_off + pm.n_y] = 1.0 / V_ocp[pl.ocp_x_off:pl.ocp_x_off + pm.n_y]
            qo = pl.ocp_x_off + pm.n_z
            W[pl.ocp_c_off + pm.n_y + pm.n_p:pl.ocp_c_off + pm.n_y + pm.n_p + pm.n_q] = 1.0 / V_ocp[qo:qo + pm.n_q]
        return W

    def expand_x(self, base: np.ndarray) -> np.ndarray:
        """Per-OCP-variable vector -> per-NLP-variable vector (scaling.py:212-241)."""
        out = np.empty(self.num_x)
        for pl, pm in zip(self.phases, self.model.phases):
            o = pl.ocp_x_off
            out[pl.x_off:pl.q_off] = np.repeat(base[o:o + pm.n_z], pl.N)
            out[pl.q_off:pl.q_off + pm.n_q + pm.n_t] = base[o + pm.n_z:o + pm.n_z + pm.n_q + pm.n_t]
        out[self.s_off:] = base[self.ocp_s_off:]
        return out

    def expand_c(self, base: np.ndarray) -> np.ndarray:
        """Per-OCP-constraint vector -> per-NLP-row vector (scaling.py:243-269)."""
        out = np.empty(self.num_c)
        for pl, pm in zip(self.phases, self.model.phases):
            o = pl.ocp_c_off
            out[pl.c_off:pl.c_path_off] = np.repeat(base[o:o + pm.n_y], pl.N - 1)
            out[pl.c_path_off:pl.c_int_off] = np.repeat(base[o + pm.n_y:o + pm.n_y + pm.n_p], pl.N)
            out[pl.c_int_off:pl.c_int_off + pm.n_q] = base[o + pm.n_y + pm.n_p:o + pm.n_y + pm.n_p + pm.n_q]
        out[self.c_end_off:] = base[self.ocp_c_end_off:]
        return out
