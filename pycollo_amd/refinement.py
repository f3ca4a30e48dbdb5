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


def _merge_sections(group, n_min):
    """``merge_sections`` of mesh_refinement.py:252-285: a run of neighbouring sections that are resolved far better
    than the tolerance asks is replaced by ``ceil(sum p / (n_min - P))`` sections of ``n_min`` nodes, their knots
    placed so that the sections' "required reduction" is spread evenly (the reference's density interpolation,
    restated with the same scipy call)."""
    from scipy import interpolate
    g = np.asarray(group, dtype=float)
    P_q, h_q, p_q = g[:, 0], g[:, 1], g[:, 2]
    T = np.sum(h_q)
    merge_ratio = p_q / (n_min - P_q)
    needed = int(np.ceil(np.sum(merge_ratio)))
    if needed == 1:
        secs = np.array([T])
    else:
        required_reduction = np.divide(h_q, merge_ratio)
        reduction_factor = np.reciprocal(np.sum(required_reduction)) * required_reduction
        knot_locations = np.cumsum(h_q) / T
        current_density = np.cumsum(reduction_factor)
        density_func = interpolate.interp1d(knot_locations, current_density, bounds_error=False, fill_value="extrapolate")
        new_density = np.linspace(1 / needed, 1, needed)
        new_knots = np.concatenate([np.array([0]), density_func(new_density)])
        secs = T * np.diff(new_knots)
    return secs.tolist(), [n_min] * needed


def _subdivide_sections(group, n_min):
    """``subdivide_sections`` of mesh_refinement.py:287-313: per section either ``k`` equal parts of ``n_min`` nodes
    or the predicted order (never below ``n_min``)."""
    g = np.asarray(group, dtype=float)
    sub_req, factor, red_tol, P_q, h_q, p_q = g[:, 0].astype(bool), g[:, 1].astype(int), g[:, 2], g[:, 3], g[:, 4], g[:, 5]
    is_red = P_q <= 0
    predicted = P_q + p_q
    predicted[is_red] = np.ceil(P_q[is_red] * red_tol[is_red]) + p_q[is_red]
    nxt = np.ones_like(predicted, dtype=int) * n_min
    nxt[~sub_req] = predicted[~sub_req]
    nxt[nxt < n_min] = n_min
    sizes, nodes = [], []
    for h, k, n in zip(h_q, factor, nxt):
        sizes.extend([h / k] * k)
        nodes.extend([int(n)] * k)
    return sizes, nodes


def next_phase_mesh(sizes, nodes, max_rel_err, *, mesh_tol=MESH_TOLERANCE, n_min=COLLOCATION_POINTS_MIN,
                    n_max=COLLOCATION_POINTS_MAX):
    """Section sizes (fractions) and node counts of the next mesh (``next_iteration_phase_mesh``,
    mesh_refinement.py:250-392).

    For every section the number of extra nodes is P = ceil(log(e / tol) / log(n)) (with the reference's
    correction for P <= 0).  Sections are walked in order and grouped into alternating runs:
    * *merge* runs -- sections whose predicted order ``P + n`` is negative, i.e. resolved several orders better than
      the tolerance (``merge_required = predicted < MERGE_TOLERANCE_FACTOR / log(tol / e)`` with the factor 0,
      mesh_refinement.py:344-347: the factor only zeroes the threshold, the branch is live) -- are coalesced into
      fewer ``n_min``-node sections (:252-285);
    * the other runs -- a section whose predicted order reaches ``n_max`` is subdivided into
      ceil(predicted / n_min) sections of ``n_min`` nodes, otherwise its order becomes the predicted one, never
      below ``n_min`` (:287-313)."""
    sizes = np.asarray(sizes, dtype=float)
    nodes = np.asarray(nodes, dtype=np.int64)
    err = np.asarray(max_rel_err, dtype=float)
    if not np.max(err) > mesh_tol:
        return sizes / sizes.sum(), nodes.copy(), True
    with np.errstate(divide="ignore", invalid="ignore"):
        P = np.ceil(np.log(err / mesh_tol) / np.log(nodes))
        neg = P <= 0
        P[neg] = P[neg] + np.ceil(np.log(-P[neg] + 1))
        predicted = P + nodes
        log_tol = np.log(np.divide(mesh_tol, err))
        merge_required = predicted < (0 / log_tol)                      # MERGE_TOLERANCE_FACTOR = 0
        red = 1 + np.reciprocal(log_tol)
    red[red < 0] = 0
    subdivide = predicted >= n_max
    level = np.ones_like(predicted)
    level[subdivide] = np.ceil(predicted[subdivide] / n_min)
    merge_group, sub_group, new_sizes, new_nodes = [], [], [], []

    def flush_merge():
        nonlocal merge_group
        if merge_group:
            s_, n_ = _merge_sections(merge_group, n_min)
            new_sizes.extend(s_)
            new_nodes.extend(n_)
            merge_group = []

    def flush_sub():
        nonlocal sub_group
        if sub_group:
            s_, n_ = _subdivide_sections(sub_group, n_min)
            new_sizes.extend(s_)
            new_nodes.extend(n_)
            sub_group = []

    for k in range(len(nodes)):
        if merge_required[k]:
            flush_sub()
            merge_group.append([P[k], sizes[k], nodes[k]])
        else:
            flush_merge()
            sub_group.append([subdivide[k], level[k], red[k], P[k], sizes[k], nodes[k]])
    flush_merge()
    flush_sub()
    new_sizes = np.asarray(new_sizes)
    return new_sizes / new_sizes.sum(), np.asarray(new_nodes, dtype=np.int64), False


def synthetic_refined_mesh(target_nodes: int, seed: int = 7, *, K0: int = 10, n0: int = COLLOCATION_POINTS_MIN,
                           mesh_tol: float = MESH_TOLERANCE, features: int = 6, max_iterations: int = 12):
    """(sizes, nodes) of a mesh as the ph rule above leaves it, without a solve: ``next_phase_mesh`` iterated from a
    coarse uniform mesh on a synthetic error field.  A section of n nodes and width h at tau is given the error
    (h / ell(tau))^n -- what a solution with local time scale ell(tau) produces -- where ell is smooth with a few sharp
    dips (``features`` of them, seeded), and its overall level is scanned for the converged mesh closest to
    ``target_nodes`` nodes.  Workload generator for bench.py / the parity tests: the mesh has what refinement produces
    (runs of n_min-node sections where sections were subdivided or merged, individual orders elsewhere,
    mesh_refinement.py:252-321), which a random assignment of orders to sections has not."""
    rng = np.random.default_rng(seed)
    centres = rng.uniform(-0.9, 0.9, features)
    widths = rng.uniform(0.02, 0.15, features)
    depth = rng.uniform(0.6, 2.0, features)     # decades

    def log10_ell(tau):
        v = 0.35 * np.sin(2.1 * tau + rng_phase)
        for c, w, d in zip(centres, widths, depth):
            v = v - d * np.exp(-0.5 * ((tau - c) / w) ** 2)
        return v
    rng_phase = rng.uniform(0, 2 * np.pi)

    def run(level):
        sizes, nodes = np.full(K0, 1.0 / K0), np.full(K0, n0, dtype=np.int64)
        for _ in range(max_iterations):
            edges = -1.0 + 2.0 * np.concatenate([[0.0], np.cumsum(sizes)])
            mid, h = 0.5 * (edges[1:] + edges[:-1]), np.diff(edges)
            err = (h / 10.0 ** (level + log10_ell(mid))) ** nodes
            sizes, nodes, met = next_phase_mesh(sizes, nodes, err, mesh_tol=mesh_tol)
            if met or int(np.sum(nodes - 1)) + 1 > 4 * target_nodes:
                break
        return sizes, nodes, met

    # the level of ell (log10) is scanned: the node count of the converged mesh falls with it, in jumps
    best = None
    for level in np.linspace(-3.5, 2.5, 145):
        sizes, nodes, met = run(level)
        N = int(np.sum(nodes - 1)) + 1
        if met and (best is None or abs(N - target_nodes) < abs(best[2] - target_nodes)):
            best = (sizes, nodes, N)
    if best is None:
        raise RuntimeError("no level of the synthetic error field gives a converged mesh")
    return best[0], best[1]
