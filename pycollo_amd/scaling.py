"""Objective / constraint scaling generation on top of the engine (the second caller of the hot path).

Restates ``IterationScaling._calculate_objective_scaling`` / ``_calculate_constraint_scaling``
(pycollo/scaling.py:346-430) without the dense ``np.array(G)`` of scaling.py:394: the row norms are
reduced on the GPU from the CSR values (``pc_row_norms_jac``).
"""
from __future__ import annotations

import numpy as np


def objective_scaling(engine, x_guess) -> float:
    """w = 1 / ||grad J||_2 evaluated with w_J = 1 (scaling.py:346-368)."""
    if engine.model.scaling_method is None:
        return 1.0
    V, r, W = engine.V_ocp, engine.r_ocp, engine.W_ocp
    old = engine.w_J
    engine.set_scaling(V, r, W, 1.0)
    g = engine.evaluate_g(x_guess)
    engine.set_scaling(V, r, W, old)
    norm = float(np.sqrt(np.sum(g * g)))
    return 1.0 if np.isclose(norm, 0.0) else 1.0 / norm


def constraint_scaling(engine, x_guess) -> np.ndarray:
    """W_ocp: defect = 1/V_y, integral = 1/V_q, path = 1/mean row norm, endpoint = 1/row norm, all from
    G evaluated with W = 1 at the guess (scaling.py:370-430)."""
    lay = engine.layout
    ones = np.ones(lay.num_ocp_c)
    if engine.model.scaling_method is None:
        return ones
    V, r, W_old, w_old = engine.V_ocp, engine.r_ocp, engine.W_ocp, engine.w_J
    engine.set_scaling(V, r, ones, w_old)
    norms = engine.G_row_norms(x_guess)
    engine.set_scaling(V, r, W_old, w_old)
    W = np.empty(lay.num_ocp_c)
    for pl, pm in zip(lay.phases, engine.model.phases):
        o = pl.ocp_c_off
        W[o:o + pm.n_y] = 1.0 / V[pl.ocp_x_off:pl.ocp_x_off + pm.n_y]
        path = norms[pl.c_path_off:pl.c_int_off].reshape(pm.n_p, pl.N)
        W[o + pm.n_y:o + pm.n_y + pm.n_p] = 1.0 / np.mean(path, axis=1)
        qo = pl.ocp_x_off + pm.n_z
        W[o + pm.n_y + pm.n_p:o + pm.n_y + pm.n_p + pm.n_q] = 1.0 / V[qo:qo + pm.n_q]
    W[lay.ocp_c_end_off:] = 1.0 / norms[lay.c_end_off:]
    return W
