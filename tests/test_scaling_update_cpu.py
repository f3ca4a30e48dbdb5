"""``update_scaling`` (pycollo/scaling.py:283-344, settings.py:272-296): the averaging over mesh iterations, restated
in pycollo_amd/scaling.py::scaling_from_previous, against the reference's formulas worked by hand."""
import numpy as np

from pycollo_amd import problems
from pycollo_amd.layout import NlpLayout
from pycollo_amd.mesh import build_phase_mesh
from pycollo_amd.model import compile_model
from pycollo_amd.quadrature import QuadratureTables
from pycollo_amd.scaling import history_weights, scaling_from_previous


def test_history_weights_follow_the_reference():
    """scaling.py:289-293 with alpha = 0.8: [alpha (1-alpha)^i] flipped, the oldest divided by alpha."""
    np.testing.assert_allclose(history_weights(1, 0.8), [1.0])
    np.testing.assert_allclose(history_weights(2, 0.8), [0.2, 0.8])
    np.testing.assert_allclose(history_weights(3, 0.8), [0.04, 0.16, 0.8])
    for n in range(1, 6):
        assert abs(history_weights(n, 0.8).sum() - 1.0) < 1e-15


def test_scaling_from_previous_double_pendulum():
    """4 states, 2 controls, 1 integral, tF free, 2 parameters: every branch of set_scales_shifts."""
    prob = problems.double_pendulum()
    model = compile_model(prob)
    quad = QuadratureTables("lobatto")
    meshes = [build_phase_mesh(quad, *ph.mesh.resolved()) for ph in prob.phases]
    lay = NlpLayout(model, meshes)
    V, r = lay.base_variable_scaling()
    rng = np.random.default_rng(3)
    x = rng.uniform(-0.5, 0.5, lay.num_x)
    hist = [(0.5, V * 1.5, r + 0.1, np.full(lay.num_ocp_c, 2.0)), (0.25, V * 0.5, r - 0.2, np.full(lay.num_ocp_c, 4.0))]
    seen = {}

    def cs(V_rows):
        seen["V"] = V_rows.copy()
        return np.full(lay.num_ocp_c, 8.0)
    w, Vn, rn, Wn = scaling_from_previous(lay, model, x, V, r, 0.125, hist, 0.8, cs)
    wt = np.array([0.04, 0.16, 0.8])
    assert abs(w - (0.5 * 0.04 + 0.25 * 0.16 + 0.125 * 0.8)) < 1e-15
    np.testing.assert_allclose(Wn, 2.0 * 0.04 + 4.0 * 0.16 + 8.0 * 0.8)
    # "current" scales, by the reference's formulas on the scaled guess
    pl, pm = lay.phases[0], model.phases[0]
    N = pl.N
    Vc, rc = V.copy(), r.copy()
    ys = x[pl.x_off:pl.x_off + pm.n_y * N].reshape(N, -1)      # scaling.py:300, literally
    us = x[pl.x_off + pm.n_y * N:pl.x_off + pm.n_z * N].reshape(N, -1)
    for o, blk in ((0, ys), (pm.n_y, us)):
        amp = blk.max(axis=0) - blk.min(axis=0)
        Vc[pl.ocp_x_off + o:pl.ocp_x_off + o + blk.shape[1]] = amp
        rc[pl.ocp_x_off + o:pl.ocp_x_off + o + blk.shape[1]] = blk.max(axis=0) - 0.5 * amp
    for o0, xs in ((pl.ocp_x_off + pm.n_z, x[pl.q_off:pl.q_off + pm.n_q + pl.n_t]), (lay.ocp_s_off, x[lay.s_off:])):
        o1 = o0 + len(xs)
        Vc[o0:o1] = np.abs(xs)
        rc[o0:o1] = np.abs(xs) / V[o0:o1] * r[o0:o1]
    np.testing.assert_allclose(Vn, wt[0] * hist[0][1] + wt[1] * hist[1][1] + wt[2] * Vc, rtol=1e-14)
    np.testing.assert_allclose(rn, wt[0] * hist[0][2] + wt[1] * hist[1][2] + wt[2] * rc, rtol=1e-14, atol=1e-15)
    np.testing.assert_array_equal(seen["V"], Vn)              # the defect / integral rows see the averaged stretches
