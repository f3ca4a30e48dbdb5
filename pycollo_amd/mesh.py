"""Per-phase temporal mesh tables (host side, once per mesh iteration).

Restates what ``Mesh.generate_single_phase`` (pycollo/mesh.py:236-356) produces, but only the
pieces the callback engine indexes by -- no sparse matrices are built here:

* ``tau``    node abscissae on [-1, 1]; each section contributes its points minus its last one
             (mesh.py:255-265)
* ``s``      ``mesh_index_boundaries``: first-node index of every section, plus N-1 (mesh.py:270-271)
* ``n``      nodes per section, both ends included (``N_K``)
* ``h``      section widths in tau (``h_K``, mesh.py:272); sum = 2
* ``w``      per-node integral weights (``W_matrix``, mesh.py:325-326): shared nodes accumulate
             the contribution of both neighbouring sections

The section-local integration block is ``h_k * A(n_k)`` and the difference block is ``[1 | -I]``
(mesh.py:297-335); the kernels apply them section by section instead of through a CSR matrix.
Naming trap: the live reference stores the integration matrix as ``sI_matrix`` and
the +-1 difference matrix as ``sA_matrix``.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .quadrature import QuadratureTables

TAU_0 = -1.0
TAU_F = 1.0


@dataclass
class PhaseMesh:
    """Mesh tables of one phase."""

    sizes: np.ndarray   # [K] section fractions of the period (sum 1)
    n: np.ndarray       # [K] int32 nodes per section
    tau: np.ndarray     # [N]
    s: np.ndarray       # [K+1] int64 first-node index of each section, s[K] = N-1
    h: np.ndarray       # [K]
    w: np.ndarray       # [N]
    method: str

    @property
    def K(self) -> int:
        return int(self.n.shape[0])

    @property
    def N(self) -> int:
        return int(self.tau.shape[0])

    @property
    def num_defect_rows(self) -> int:
        """``num_c_defect_per_y`` (mesh.py:276)."""
        return self.N - 1


def build_phase_mesh(quad: QuadratureTables, sizes, nodes) -> PhaseMesh:
    sizes = np.asarray(sizes, dtype=np.float64).reshape(-1)
    n = np.asarray(nodes, dtype=np.int64).reshape(-1)
    if n.shape[0] == 1 and sizes.shape[0] > 1:
        n = np.full(sizes.shape[0], int(n[0]), dtype=np.int64)
    if sizes.shape != n.shape:
        raise ValueError("mesh_section_sizes and number_mesh_section_nodes must have equal length")
    if np.any(n < 2):
        raise ValueError("every mesh section needs at least two nodes")
    K = n.shape[0]
    # section boundaries accumulate left to right exactly like mesh.py:248-252
    edges = np.empty(K + 1)
    edges[0] = TAU_0
    period = TAU_F - TAU_0
    for k in range(K):
        edges[k + 1] = edges[k] + period * sizes[k]
    s = np.concatenate(([0], np.cumsum(n - 1))).astype(np.int64)
    N = int(s[-1]) + 1
    tau = np.empty(N)
    uniform_orders = np.unique(n)
    for order in uniform_orders:
        ks = np.nonzero(n == order)[0]
        x = quad.points(int(order))[:-1]
        half = 0.5 * (edges[ks + 1] - edges[ks])
        mid = 0.5 * (edges[ks] + edges[ks + 1])
        idx = s[ks][:, None] + np.arange(order - 1)[None, :]
        tau[idx] = half[:, None] * x[None, :] + mid[:, None]
    tau[-1] = TAU_F
    h = np.diff(tau[s])
    w = np.zeros(N)
    for order in uniform_orders:
        ks = np.nonzero(n == order)[0]
        wq = quad.weights(int(order))
        idx = s[ks][:, None] + np.arange(order)[None, :]
        # np.add.at keeps the "+=" accumulation of mesh.py:325-326 at shared nodes
        np.add.at(w, idx, wq[None, :] * h[ks][:, None])
    return PhaseMesh(sizes=sizes, n=n.astype(np.int32), tau=tau, s=s, h=h, w=w, method=quad.method)


def uniform_phase_mesh(quad: QuadratureTables, K: int, order: int) -> PhaseMesh:
    return build_phase_mesh(quad, np.full(K, 1.0 / K), np.full(K, order, dtype=np.int64))
