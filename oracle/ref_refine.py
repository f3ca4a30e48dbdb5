"""CPU oracle for the ph mesh-error estimate (TEST INFRASTRUCTURE ONLY).

NumPy restatement of ``PattersonRaoMeshRefinement.mesh_error`` / ``phase_mesh_error``
(pycollo/mesh_refinement.py:63-240) together with the per-section polynomial fits it consumes
(pycollo/solution/solution_abc.py:60-107, Lobatto branch):

1. f at the solution nodes (``dy_iter_callable``, pycollo/solution/casadi_solution.py:71);
2. per section and state a degree n_k-1 Legendre fit of (T/2) f, integrated from the section start with
   constant y[start]; per control a degree n_k-1 polynomial fit (solution_abc.py:70-100);
3. the "ph mesh": same sections, n_k + 1 nodes each (mesh_refinement.py:76-88); section boundary values are
   the solution's, interior values come from the fits (mesh_refinement.py:160-196);
4. f on the ph mesh, ``stretch * I_ph f`` added to the section start value, compared with the fitted states;
   relative to 1 + (1 + max|Y_k|) (sic, mesh_refinement.py:211,221-223); section maximum over states/nodes.
"""
from __future__ import annotations

import numpy as np

from .ref_numpy import OracleMesh, OracleNlp


def mesh_error(ora: OracleNlp, xt):
    """Returns per phase (absolute errors [K][n_y][max m_k], max relative error [K])."""
    xt = np.asarray(xt, float)
    out = []
    for P in ora.P:
        mesh = P.mesh
        z, q, stretch, _, w = ora._unpack(P, xt)
        N, K = mesh.N, mesh.K
        y, u = z[:P.n_y], z[P.n_y:]
        a = ora._args(P, z, w)
        dy = np.array([P.F_fn[i](*a) for i in range(P.n_y)])
        tau = mesh.tau
        ph = OracleMesh(ora.tables, mesh.h / mesh.h.sum(), mesh.nodes + 1)
        y_ph = np.zeros((P.n_y, ph.N))
        u_ph = np.zeros((P.n_u, ph.N))
        y_ph[:, ph.bnd] = y[:, mesh.bnd]
        u_ph[:, ph.bnd] = u[:, mesh.bnd]
        for k in range(K):
            i0, i1 = mesh.bnd[k], mesh.bnd[k + 1]
            t_k = tau[i0:i1 + 1]
            sl = slice(ph.bnd[k] + 1, ph.bnd[k + 1])
            for iy in range(P.n_y):
                dpoly = np.polynomial.Legendre.fit(t_k, dy[iy, i0:i1 + 1] * stretch, deg=mesh.nodes[k] - 1, window=[0, 1])
                y_ph[iy, sl] = dpoly.integ(k=y[iy, i0])(ph.tau[sl])
            for iu in range(P.n_u):
                upoly = np.polynomial.Polynomial.fit(t_k, u[iu, i0:i1 + 1], deg=mesh.nodes[k] - 1, window=[0, 1])
                u_ph[iu, sl] = upoly(ph.tau[sl])
        zp = np.vstack([y_ph, u_ph])
        ap = [zp[i] for i in range(P.n_z)] + [np.full(ph.N, w[i]) for i in range(P.n_w)]
        dy_ph = np.array([P.F_fn[i](*ap) for i in range(P.n_y)])           # [n_y][N_ph]
        I_dy = stretch * (ph.I_mat @ dy_ph.T)                               # [N_ph - 1][n_y]
        mmax = int(ph.nodes.max()) - 1
        abs_err = np.zeros((K, P.n_y, mmax))
        max_rel = np.zeros(K)
        for k in range(K):
            i0, m = ph.bnd[k], ph.nodes[k] - 1
            Y_ph = (y_ph[:, i0] + I_dy[i0:i0 + m]).T                         # [n_y][m]
            Y = y_ph[:, i0 + 1:i0 + 1 + m]
            err = np.abs(Y_ph - Y)
            abs_err[k, :, :m] = err
            scale = np.max(np.abs(Y), axis=1) + 1
            max_rel[k] = np.max(err / (1 + scale)[:, None])
        out.append((abs_err, max_rel))
    return out
