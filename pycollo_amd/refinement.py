"""ph mesh refinement: error estimate on the GPU, next mesh on the host.

Restates ``PattersonRaoMeshRefinement`` (pycollo/mesh_refinement.py:61-392).  The error estimate -- f on
the mesh, per-section interpolants, f again on a mesh with one more node per section, a section-local
integration and a max-reduction -- re-uses the model tape and runs as one kernel per phase
(``pc_mesh_error``); the reference builds a second CasADi function with one ``ca.substitute`` per node for it
(mesh_refinement.py:90-158) and fits ``K * (n_y + n_u)`` NumPy polynomials (solution_abc.py:60-107).  The
merge / subdivide decision per section is O(K) scalar logic and stays on the host.
"""
from __future__ import annotations

import numpy as np
from numpy.polynomial import legendre as _leg

from .quadrature import QuadratureTables

MESH_TOLERANCE = 1e-7           # pycollo/settings.py default (mesh_refinement.py:329)
COLLOCATION_POINTS_MIN = 4      # pycollo/quadrature.py:36-37
COLLOCATION_POINTS_MAX = 10


def ph_tables(quad: QuadratureTables, n: int):
    """(B, E, A_ph) for section order n: the interpolant through the n section nodes, integrated up to /
    evaluated at the n-1 interior nodes of the order-(n+1) rule, and the order-(n+1) integration matrix.

    B and E are what ``Legendre.fit(...).integ(k=y0)`` and ``Polynomial.fit(...)`` of
    solution_abc.py:70-100 evaluate to at the ph nodes (a degree n-1 fit through n points interpolates)."""
    x = quad.points(n)                      # solution nodes on [-1, 1]
    xp = quad.points(n + 1)[1:-1]           # interior ph nodes
    V = _leg.legvander(x, n - 1)            # V[i, k] = P_k(x_i)
    Vp = _leg.legvander(xp, n)              # up to P_n for the integrals
    coef = np.linalg.inv(V)                 # coef[k, i]: Legendre coefficients of the Lagrange basis l_i
    E = Vp[:, :n] @ coef
    # int_{-1}^{x} P_k = (P_{k+1}(x) - P_{k-1}(x)) / (2k + 1), k >= 1;  x + 1 for k = 0
    I = np.empty((xp.size, n))
    I[:, 0] = xp + 1.0
    for k in range(1, n):
        I[:, k] = (Vp[:, k + 1] - Vp[:, k - 1]) / (2 * k + 1)
    B = 0.5 * (I @ coef)                    # section variable c in [0, 1]: dx = 2 dc
    return B, E, quad.A(n + 1)


def mesh_error(engine, x_tilde):
    """Per phase: (max relative error per section [K], max absolute error per section and state [K][n_y])."""
    out = []
    for ip, mesh in enumerate(engine.meshes):
        orders = sorted({int(n) for n in np.unique(mesh.n)})
        tabs = [ph_tables(engine.quad, n) for n in orders]
        B = np.concatenate([t[0].ravel() for t in tabs])
        E = np.concatenate([t[1].ravel() for t in tabs])
        A = np.concatenate([t[2].ravel() for t in tabs])
        out.append(engine.mesh_error(ip, x_tilde, orders, B, E, A))
    return out


def next_phase_mesh(sizes, nodes, max_rel_err, *, mesh_tol=MESH_TOLERANCE, n_min=COLLOCATION_POINTS_MIN,
                    n_max=COLLOCATION_POINTS_MAX):
    """Section sizes (fractions) and node counts of the next mesh (mesh_refinement.py:250-392).

    For every section the number of extra nodes is P = ceil(log(e / tol) / log(n)) (with the reference's
    correction for P <= 0); a section whose predicted order reaches ``n_max`` is subdivided into
    ceil(predicted / n_min) sections of ``n_min`` nodes, otherwise its order becomes the predicted one
    (never below ``n_min``).  The reference's merge branch is disabled by its MERGE_TOLERANCE_FACTOR = 0
    (mesh_refinement.py:344-347) and is therefore not restated."""
    sizes = np.asarray(sizes, dtype=float)
    nodes = np.asarray(nodes, dtype=np.int64)
    err = np.asarray(max_rel_err, dtype=float)
    if not np.max(err) > mesh_tol:
        return sizes / sizes.sum(), nodes.copy(), True
    ratio = err / mesh_tol
    P = np.ceil(np.log(ratio) / np.log(nodes))
    neg = P <= 0
    P[neg] = P[neg] + np.ceil(np.log(-P[neg] + 1))
    predicted = P + nodes
    log_tol = np.log(mesh_tol / err)
    with np.errstate(divide="ignore"):
        red = 1 + np.reciprocal(log_tol)
    red[red < 0] = 0
    subdivide = predicted >= n_max
    new_sizes, new_nodes = [], []
    for k in range(len(nodes)):
        if subdivide[k]:
            parts = int(np.ceil(predicted[k] / n_min))
            new_sizes += [sizes[k] / parts] * parts
            new_nodes += [n_min] * parts
        else:
            pn = P[k] + nodes[k]
            if P[k] <= 0:
                pn = np.ceil(P[k] * red[k]) + nodes[k]
            new_sizes.append(sizes[k])
            new_nodes.append(int(max(pn, n_min)))
    new_sizes = np.asarray(new_sizes)
    return new_sizes / new_sizes.sum(), np.asarray(new_nodes, dtype=np.int64), False
