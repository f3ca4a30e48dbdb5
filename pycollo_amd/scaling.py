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


def constraint_scaling(engine, x_guess, V_rows=None) -> np.ndarray:
    """W_ocp: defect = 1/V_y, integral = 1/V_q, path = 1/mean row norm, endpoint = 1/row norm, all from
    G evaluated with W = 1 at the guess (scaling.py:370-430).  ``V_rows``: the variable stretches the defect and
    integral rows are scaled by when they differ from the ones the NLP functions carry (``update_scaling``)."""
    lay = engine.layout
    ones = np.ones(lay.num_ocp_c)
    if engine.model.scaling_method is None:
        return ones
    V, r, W_old, w_old = engine.V_ocp, engine.r_ocp, engine.W_ocp, engine.w_J
    engine.set_scaling(V, r, ones, w_old)
    norms = engine.G_row_norms(x_guess)
    engine.set_scaling(V, r, W_old, w_old)
    W = np.empty(lay.num_ocp_c)
    if V_rows is not None:
        V = np.asarray(V_rows, dtype=np.float64)
    for pl, pm in zip(lay.phases, engine.model.phases):
        o = pl.ocp_c_off
        W[o:o + pm.n_y] = 1.0 / V[pl.ocp_x_off:pl.ocp_x_off + pm.n_y]
        path = norms[pl.c_path_off:pl.c_int_off].reshape(pm.n_p, pl.N)
        W[o + pm.n_y:o + pm.n_y + pm.n_p] = 1.0 / np.mean(path, axis=1)
        qo = pl.ocp_x_off + pm.n_z
        W[o + pm.n_y + pm.n_p:o + pm.n_y + pm.n_p + pm.n_q] = 1.0 / V[qo:qo + pm.n_q]
    W[lay.ocp_c_end_off:] = 1.0 / norms[lay.c_end_off:]
    return W


def history_weights(n: int, alpha: float) -> np.ndarray:
    """Weights of ``n`` mesh iterations' scalings, oldest first, the current one last (scaling.py:289-293):
    ``alpha (1 - alpha)^i`` counted back from the current iteration, the oldest weight divided by alpha so that the
    weights sum to one."""
    w = np.array([alpha * (1.0 - alpha) ** i for i in range(n)])[::-1].copy()
    w[0] /= alpha
    return w


def scaling_from_previous(layout, model, guess_x_tilde, V_ocp, r_ocp, w_now, history, alpha, constraint_scaling_fn):
    """``IterationScaling._generate_from_previous`` (pycollo/scaling.py:283-344; settings ``update_scaling=True``,
    ``scaling_weight=alpha``): objective, variable and constraint scalings of this mesh iteration as exponentially
    weighted averages over the previous iterations' and this one's.

    ``history``: [(w, V_ocp, r_ocp, W_ocp), ...] of the earlier mesh iterations, oldest first.  ``V_ocp`` / ``r_ocp``:
    this iteration's base scaling (scaling.py:160-166).  ``w_now``: ``objective_scaling`` at the scaled guess.
    ``constraint_scaling_fn(V_ocp)``: ``constraint_scaling`` at the scaled guess given the variable stretches.
    Returns ``(w, V_ocp, r_ocp, W_ocp)``.

    Restated literally, including two things a reader will stumble over:
    * the "current" variable scales are taken from the *scaled* guess (the guess has been through ``scale_x`` when this
      runs, pycollo/iteration.py:69-79,360-373), and a state's samples are gathered with ``var.reshape(N, -1)`` on a
      variable-major slice (scaling.py:299-300), i.e. column j of that view strides through the slice with step
      n_vars, not through variable j;
    * the reference bakes V and r into its NLP callables *before* this update runs (backend.py:1459-1463 inside
      ``generate_nlp_function_callables``, then ``generate_scaling``; iteration.py:375-394), so the updated V / r reach
      the scaled bounds and the unscaling of the solution, not the NLP functions (see ``MeshIteration``)."""
    V_ocp = np.array(V_ocp, dtype=np.float64, copy=True)
    r_ocp = np.array(r_ocp, dtype=np.float64, copy=True)
    x = np.asarray(guess_x_tilde, dtype=np.float64)
    weights = history_weights(len(history) + 1, alpha)
    w = float(np.average(np.array([h[0] for h in history] + [w_now], dtype=np.float64), weights=weights))

    def set_scales_shifts(o0, o1, x0, x1, N=None):
        var = x[x0:x1]
        if var.size == 0:
            return
        if N:
            v2 = var.reshape(N, -1)                                    # scaling.py:300 (see above)
            vmin, vmax = v2.min(axis=0), v2.max(axis=0)
            amp = vmax - vmin
            V_ocp[o0:o1] = amp
            r_ocp[o0:o1] = vmax - 0.5 * amp
        else:
            V_last, r_last = V_ocp[o0:o1].copy(), r_ocp[o0:o1].copy()
            V_next = np.abs(var)
            V_ocp[o0:o1] = V_next
            r_ocp[o0:o1] = (V_next / V_last) * r_last

    for pl, pm in zip(layout.phases, model.phases):
        N, o = pl.N, pl.ocp_x_off
        set_scales_shifts(o, o + pm.n_y, pl.x_off, pl.x_off + pm.n_y * N, N)
        set_scales_shifts(o + pm.n_y, o + pm.n_z, pl.x_off + pm.n_y * N, pl.x_off + pm.n_z * N, N)
        set_scales_shifts(o + pm.n_z, o + pm.n_z + pm.n_q, pl.q_off, pl.q_off + pm.n_q)
        set_scales_shifts(o + pm.n_z + pm.n_q, o + pm.n_z + pm.n_q + pl.n_t, pl.t_off, pl.t_off + pl.n_t)
    set_scales_shifts(layout.ocp_s_off, layout.ocp_s_off + layout.n_s, layout.s_off, layout.s_off + layout.n_s)
    V_new = np.average(np.vstack([h[1] for h in history] + [V_ocp]), axis=0, weights=weights)
    r_new = np.average(np.vstack([h[2] for h in history] + [r_ocp]), axis=0, weights=weights)
    W_now = constraint_scaling_fn(V_new)
    W_new = np.average(np.vstack([h[3] for h in history] + [W_now]), axis=0, weights=weights)
    return w, V_new, r_new, W_new
