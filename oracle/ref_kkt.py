"""NumPy execution of the GPU KKT solver's index tables (TEST INFRASTRUCTURE -- never imported by the product).

``pycollo_amd/kkt.py`` lays the interior-point KKT matrix out as leaves / chain nodes / border and
``pycollo_amd/csrc/pc_kkt.hip`` eliminates them on the GPU.  This module runs the same tables and the same
elimination order with dense NumPy blocks, so that (i) the tables can be checked against a general sparse solver on a
CPU-only machine and (ii) the GPU kernels have a step-by-step reference.  The matrix it stands for is the one IPOPT
would hand to MUMPS (pycollo/backend.py:1703-1711, ``linear_solver``)."""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

from pycollo_amd.kkt import SRC_G, SRC_H, KktTables


def _ldl_inplace(M, m):
    """L D L^T of the leading m x m block of M = [A | C] without pivoting (lower triangle read), applied to the
    trailing columns too: on return the strict lower triangle holds L, the diagonal D, and the trailing columns
    Z = L^-1 C.  Returns (number of positive, number of negative) pivots."""
    npos = nneg = 0
    for j in range(m):
        d = M[j, j]
        npos += d > 0
        nneg += d < 0
        col = M[j + 1:m, j].copy()
        l = col / d
        M[j + 1:m, j] = l
        for i in range(j + 1, m):                      # trailing update of the lower triangle
            M[i, j + 1:i + 1] -= l[i - j - 1] * col[:i - j]
        M[j + 1:m, m:] -= np.outer(l, M[j, m:])
    return npos, nneg


def _solve_ldl(M, m, R):
    """A^-1 R for the factor stored in M (R: m x k)."""
    R = R.copy()
    for j in range(m):
        R[j + 1:] -= np.outer(M[j + 1:m, j], R[j])
    R /= np.diag(M[:m, :m])[:, None]
    for j in range(m - 1, -1, -1):
        R[j] -= M[j + 1:m, j] @ R[j + 1:]
    return R


def _source_values(kind, idx, G, H, use_H):
    out = np.ones(len(kind))
    g, h = kind == SRC_G, kind == SRC_H
    out[g] = np.asarray(G, float)[idx[g]]
    out[h] = np.asarray(H, float)[idx[h]] if use_H else 0.0
    return out


class RefKkt:
    def __init__(self, T: KktTables, G=None, H=None):
        self.T = T
        self.G, self.H = G, H          # the values the partial entry points read (a rank's own, NaN elsewhere)
        self.m_l = np.diff(T.leaf_ptr)
        self.nzb = np.diff(T.chain_ptr)
        last = np.zeros(T.n_chain, bool)
        last[T.chain_phase_ptr[1:] - 1] = True
        self.last = last
        self.nzb_next = np.where(last, 0, np.concatenate([self.nzb[1:], [0]]))
        self.base_chain = int(T.leaf_ptr[-1])
        self.base_border = self.base_chain + int(T.chain_ptr[-1])

    def _values(self, G, H, use_H):
        T = self.T
        src = _source_values(T.src_kind, T.src_idx, G, H, use_H) * T.src_coef
        vals = np.zeros(T.total_vals)
        vals[T.dst] = np.add.reduceat(src, T.run_ptr[:-1]) if len(T.dst) else 0.0
        return vals

    def assemble(self, G, H, dvec, use_H=True):
        """The whole matrix as scipy CSR over natural unknowns (for checks against a general solver)."""
        T = self.T
        val = _source_values(T.mv_kind, T.mv_idx, G, H, use_H) * T.mv_coef
        K = sp.csr_matrix((val, T.mv_col, T.mv_ptr), shape=(T.nu, T.nu))
        K.sum_duplicates()
        diag = np.where(T.fixed.astype(bool), 1.0, dvec)
        return K + sp.diags(diag)

    def matvec(self, G, H, dvec, x, use_H=True):
        return self.assemble(G, H, dvec, use_H) @ x

    def factor(self, G, H, dvec, use_H=True):
        T = self.T
        v = self._values(G, H, use_H)
        fixed = T.fixed.astype(bool)
        v[T.diag_pos] = np.where(fixed, 1.0, v[T.diag_pos] + dvec)
        nb = T.nb
        npos = nneg = 0
        self.leafM, self.leafS = [], []
        for l in range(T.n_leaf):
            m, left = int(self.m_l[l]), int(T.leaf_left[l])
            w = int(self.nzb[left] + self.nzb[left + 1] + nb)
            M = v[T.leafA_off[l]:T.leafA_off[l] + m * (m + w)].reshape(m, m + w).copy()
            p, q = _ldl_inplace(M, m)
            npos += p; nneg += q
            Z = M[:, m:].copy()
            d = np.diag(M[:m, :m])
            S = -(Z.T @ (Z / d[:, None])) if m else np.zeros((w, w))
            X = Z / d[:, None] if m else Z
            for j in range(m - 1, -1, -1):
                X[j] -= M[j + 1:m, j] @ X[j + 1:]
            M[:, m:] = X
            self.leafM.append(M); self.leafS.append(S)
        self.chainM = [None] * T.n_chain
        Bd = v[T.border_off:T.border_off + nb * nb].reshape(nb, nb).copy()
        Bd = np.tril(Bd) + np.tril(Bd, -1).T
        for S in self.leafS:                                        # border part of every leaf's Schur block
            Bd += S[S.shape[0] - nb:, S.shape[0] - nb:]
        for ip in range(T.n_phase):
            carry = None                                            # Schur block of the previous chain node
            for c in range(int(T.chain_phase_ptr[ip]), int(T.chain_phase_ptr[ip + 1])):
                nz, nx = int(self.nzb[c]), int(self.nzb_next[c])
                wc = nx + nb
                M = v[T.chainD_off[c]:T.chainD_off[c] + nz * (nz + wc)].reshape(nz, nz + wc).copy()
                M[:, :nz] = np.tril(M[:, :nz]) + np.tril(M[:, :nz], -1).T
                k = c - int(T.chain_phase_ptr[ip])
                if k > 0:                                           # leaf on the left: its R / border rows
                    S = self.leafS[int(np.nonzero(T.leaf_left == c - 1)[0][0])]
                    nl = int(self.nzb[c - 1])
                    M[:, :nz] += S[nl:nl + nz, nl:nl + nz]
                    M[:, nz + nx:] += S[nl:nl + nz, nl + nz:]
                    M[:, :nz] += carry[:nz, :nz]
                    M[:, nz + nx:] += carry[:nz, nz:]
                if not self.last[c]:                                # leaf on the right: its L rows
                    S = self.leafS[int(np.nonzero(T.leaf_left == c)[0][0])]
                    M[:, :nz] += S[:nz, :nz]
                    M[:, nz:nz + nx] += S[:nz, nz:nz + nx]
                    M[:, nz + nx:] += S[:nz, nz + nx:]
                p, q = _ldl_inplace(M, nz)
                npos += p; nneg += q
                Z = M[:, nz:].copy()
                d = np.diag(M[:nz, :nz])
                carry = -(Z.T @ (Z / d[:, None]))
                X = Z / d[:, None]
                for j in range(nz - 1, -1, -1):
                    X[j] -= M[j + 1:nz, j] @ X[j + 1:]
                M[:, nz:] = X
                self.chainM[c] = M
                Bd += carry[nx:, nx:]
        self.B_unfactored = Bd.copy()
        self.partial_counts = (int(npos), int(nneg))
        if getattr(self, "_partial", False):        # (factor_partial: the border is factorised elsewhere)
            return self.partial_counts
        self.Bd = Bd.copy()
        p, q = _ldl_inplace(self.Bd, nb)
        return int(npos + p), int(nneg + q)

    # ---- a rank's part of a factorisation cut across ranks (pycollo_amd/kkt_sharded.py; pc_kkt_*_partial) -----------
    def factor_partial(self, dvec, use_H=True):
        """(border block with every Schur complement added, not factorised; pivots of leaves and chain)."""
        self._partial = True
        try:
            self.factor(self.G, self.H, dvec, use_H)
        finally:
            self._partial = False
        return self.B_unfactored, *self.partial_counts

    def border_load_factor(self, B):
        B = np.asarray(B, float)
        self.Bd = np.tril(B) + np.tril(B, -1).T
        return _ldl_inplace(self.Bd, self.T.nb)

    def forward_partial(self, rhs):
        self._stop_at_border = True
        try:
            return self.solve(rhs)
        finally:
            self._stop_at_border = False

    def backward_partial(self, xb):
        self._xb_given = np.asarray(xb, float)
        try:
            x = self.solve(self._rhs_kept)
        finally:
            self._xb_given = None
        return x

    def solve(self, rhs):
        T = self.T
        nb = T.nb
        r = np.asarray(rhs, float)[T.perm].copy()
        fixed = T.fixed.astype(bool)[T.perm]
        r[fixed] = 0.0
        x = np.zeros(T.nu)
        rb = r[self.base_border:].copy()
        rc = [r[self.base_chain + T.chain_ptr[c]:self.base_chain + T.chain_ptr[c + 1]].copy() for c in range(T.n_chain)]
        tl = []
        for l in range(T.n_leaf):
            m, left = int(self.m_l[l]), int(T.leaf_left[l])
            M = self.leafM[l]
            rl = r[T.leaf_ptr[l]:T.leaf_ptr[l + 1]]
            g = M[:, m:].T @ rl                                     # X_C^T r_l
            nl, nr = int(self.nzb[left]), int(self.nzb[left + 1])
            rc[left] -= g[:nl]
            rc[left + 1] -= g[nl:nl + nr]
            rb -= g[nl + nr:]
            tl.append(_solve_ldl(M, m, rl[:, None])[:, 0] if m else rl)
        tc = [None] * T.n_chain
        for c in range(T.n_chain):
            nz, nx = int(self.nzb[c]), int(self.nzb_next[c])
            M = self.chainM[c]
            g = M[:, nz:].T @ rc[c]
            if nx:
                rc[c + 1] -= g[:nx]
            rb -= g[nx:]
            tc[c] = _solve_ldl(M, nz, rc[c][:, None])[:, 0]
        if getattr(self, "_stop_at_border", False):
            self._rhs_kept = np.asarray(rhs, float).copy()
            return rb
        if getattr(self, "_xb_given", None) is not None:
            xb = self._xb_given
        else:
            xb = _solve_ldl(self.Bd, nb, rb[:, None])[:, 0] if nb else rb
        xc = [None] * T.n_chain
        for c in range(T.n_chain - 1, -1, -1):
            nz, nx = int(self.nzb[c]), int(self.nzb_next[c])
            M = self.chainM[c]
            sep = np.concatenate([xc[c + 1] if nx else np.zeros(0), xb])
            xc[c] = tc[c] - M[:, nz:] @ sep
        out = np.zeros(T.nu)
        for l in range(T.n_leaf):
            m, left = int(self.m_l[l]), int(T.leaf_left[l])
            sep = np.concatenate([xc[left], xc[left + 1], xb])
            out[T.leaf_ptr[l]:T.leaf_ptr[l + 1]] = tl[l] - self.leafM[l][:, m:] @ sep
        for c in range(T.n_chain):
            out[self.base_chain + T.chain_ptr[c]:self.base_chain + T.chain_ptr[c + 1]] = xc[c]
        out[self.base_border:] = xb
        x[T.perm] = out
        return x
