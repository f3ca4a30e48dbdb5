"""Symbolic side of the GPU KKT solve.

What it replaces: the sparse symmetric-indefinite factorisation inside IPOPT -- MUMPS by default -- that the reference
selects by name only (``linear_solver``, pycollo/backend.py:1703-1711, pycollo/settings.py:49-59).  One interior-point
iteration solves

    [ W + Sigma + dw I    J^T   ] [dv  ]   [r1]
    [ J                 -dc I   ] [dlam] = [r2]

with W = the Lagrangian Hessian H~, J = the constraint Jacobian G~ (plus -1 columns of the slacks of inequality rows).
The matrix is *quasi-definite* whenever the (1,1) block is positive definite, so L D L^T exists for every symmetric
ordering with 1 x 1 pivots, and the signs of D are the inertia IPOPT's regularisation is driven by.  That freedom is
spent on the collocation structure (no fill-reducing heuristic, no pivot search):

  leaf (p, k)     the nodes strictly inside a run of g consecutive mesh sections of phase p: their z, path slacks and
                  path multipliers, and the defect multipliers of the rows that end on those nodes.  Leaves touch each
                  other only through separators, so all leaves are eliminated at once (one workgroup each, dense).
  chain node      the section boundary node between two leaves (its z, path slacks / multipliers) and the defect
  (p, k)          multipliers of the rows that end on it.  After the leaves are gone these nodes form a
                  block-tridiagonal chain per phase, eliminated by cyclic reduction (log2 of its length levels, tables
                  in csrc/pc_kkt_cr.hpp; node by node in one workgroup per phase where a block is too large for the
                  level kernels) -- the only sequential part, which is why g > 1: g sections per leaf divide its length.
  border          everything global: integrals q, free times, static parameters, integral and endpoint multipliers,
                  endpoint slacks, and any endpoint variable an endpoint Hessian term couples across nodes.  Dense,
                  factorised last.

Every defect multiplier sits with the node its row ends on, i.e. next to the -W V entry that pairs it with that node's
state, so no block's multiplier part rests on the -dc I regularisation alone.

This module only builds index tables (NumPy); the numeric work is in ``csrc/pc_kkt.hip`` (``pc_kkt_*`` in
include/pycollo_amd.h).  ``oracle/ref_kkt.py`` holds a NumPy execution of the same tables for the tests.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

LEAF, CHAIN, BORDER = 0, 1, 2
SRC_G, SRC_H, SRC_ONE = 0, 1, 2


@dataclass
class KktTables:
    """Everything ``pc_kkt_create`` needs, as flat arrays (see ``pc_kkt_desc`` in include/pycollo_amd.h)."""
    nu: int                      # unknowns: nv primal (x, slacks) then m multipliers, natural order
    nv: int
    n_leaf: int
    n_chain: int
    n_phase: int
    nb: int                      # border size
    n_primal: int                # expected positive pivots (unit pivots of fixed unknowns included)
    n_dual: int                  # expected negative pivots
    # unknown -> block
    perm: np.ndarray             # [nu] natural index of every unknown in block order: leaves, chain nodes, border
    leaf_ptr: np.ndarray         # [n_leaf + 1] into perm
    chain_ptr: np.ndarray        # [n_chain + 1] into perm (offset by leaf_ptr[-1])
    chain_phase_ptr: np.ndarray  # [n_phase + 1] chain nodes of every phase (consecutive)
    leaf_left: np.ndarray        # [n_leaf] chain node on the left of a leaf; the right one is left + 1
    # value buffer layout (doubles)
    leafA_off: np.ndarray        # [n_leaf] A | C of a leaf, row stride m + w
    leafS_off: np.ndarray        # [n_leaf] Schur block of a leaf, w x w
    chainD_off: np.ndarray       # [n_chain] D | E | F of a chain node, row stride nzb + nzb_next + nb
    chainS_off: np.ndarray       # [n_chain] Schur block of a chain node, (nzb_next + nb)^2
    border_off: int
    total_vals: int
    # scatter recipe: destination runs
    dst: np.ndarray              # [n_dst] int64 position in the value buffer
    run_ptr: np.ndarray          # [n_dst + 1]
    src_kind: np.ndarray         # [n_src] int32 SRC_*
    src_idx: np.ndarray          # [n_src] int32 index into G~ / H~ values
    src_coef: np.ndarray         # [n_src] double
    diag_pos: np.ndarray         # [nu] position of every unknown's diagonal entry (natural order)
    fixed: np.ndarray            # [nu] uint8
    # full symmetric matrix as CSR over natural unknowns, for products K x (refinement, J^T lambda)
    mv_ptr: np.ndarray
    mv_col: np.ndarray
    mv_kind: np.ndarray
    mv_idx: np.ndarray
    mv_coef: np.ndarray
    chain_export: np.ndarray | None = None   # [n_chain] uint8: chain nodes that are not eliminated (kkt_sharded.py), or None


def _node_maps(engine, group, cuts=None):
    """Per phase: leaf boundaries (every ``group``-th section boundary, restarting at every cut), node -> (is boundary,
    leaf index), and the chain numbering.  ``cuts[ip]``: interior section-boundary nodes at which the phase's chain is cut
    (the sharded factorisation, kkt_sharded.py): a cut node's unknowns go to the border, and the node stands in the
    chain twice with no unknowns -- as the last node of the segment on its left and the first of the one on its right --
    so every segment is a chain of its own, exactly like a phase.

    Returns per phase (s, N, is_b, sec, is_cut, chain_id, leaf_left, n_chain, seg_ptr): ``chain_id[i]`` the chain node of
    boundary i (the left copy of a cut), ``leaf_left[j]`` the chain node on the left of leaf j, both counted from the
    phase's first chain node; ``seg_ptr`` the first chain node of every segment and one past the last."""
    out = []
    for ip, (mesh, g) in enumerate(zip(engine.meshes, group)):
        s_all = np.asarray(mesh.s, dtype=np.int64)
        cut = np.zeros(0, np.int64) if cuts is None else np.unique(np.asarray(cuts[ip], dtype=np.int64))
        kc = np.searchsorted(s_all, cut)
        if len(cut) and (np.any(kc >= len(s_all)) or np.any(s_all[np.minimum(kc, len(s_all) - 1)] != cut)
                         or cut[0] <= 0 or cut[-1] >= s_all[-1]):
            raise ValueError(f"phase {ip}: a cut must be an interior section boundary")
        edges = np.concatenate([[0], kc, [len(s_all) - 1]]).astype(np.int64)
        s = np.unique(np.concatenate([s_all[a:b:g] for a, b in zip(edges[:-1], edges[1:])] + [s_all[-1:]]))
        N = int(s[-1]) + 1
        is_b = np.zeros(N, bool)
        is_b[s] = True
        sec = np.searchsorted(s, np.arange(N), side="right") - 1      # section whose start <= node
        is_cut = np.isin(s, cut)
        upto = np.cumsum(is_cut)
        before = upto - is_cut
        chain_id = np.arange(len(s), dtype=np.int64) + before
        leaf_left = (np.arange(len(s) - 1, dtype=np.int64) + upto[:-1]).astype(np.int64)
        n_chain = len(s) + int(is_cut.sum())
        seg_ptr = np.concatenate([[0], chain_id[is_cut] + 1, [n_chain]]).astype(np.int64)
        out.append((s, N, is_b, sec, is_cut, chain_id, leaf_left, n_chain, seg_ptr))
    return out


LEAF_TARGET = 72   # unknowns a leaf should hold: its dense [A | C] then fits the workgroup's 64 KB of LDS


def default_group(engine, ineq_rows) -> list[int]:
    """Sections per leaf, per phase.  The chain of separators is eliminated node after node by one workgroup per
    phase, ~10 us a node; a leaf costs its size cubed.  Merging g sections into a leaf divides the chain by g: g is the
    largest that keeps a typical leaf near LEAF_TARGET unknowns."""
    ineq = set(int(r) for r in np.asarray(ineq_rows).reshape(-1))
    out = []
    for pl, pm, mesh in zip(engine.layout.phases, engine.model.phases, engine.meshes):
        n_mean = float(np.mean(mesh.n))
        slack = sum(1 for mm in range(pm.n_p) if (pl.c_path_off + mm * pl.N) in ineq)
        per_node = pm.n_z + pm.n_p + slack + pm.n_y
        per_section = per_node * (n_mean - 1)
        out.append(int(max(1, min(64, LEAF_TARGET // max(1.0, per_section)))))
    return out


class _Plan(C.Structure):
    _fields_ = [("nu", C.c_int64), ("nb", C.c_int64), ("border_off", C.c_int64), ("cls", C.POINTER(C.c_int8)),
                ("blk", C.POINTER(C.c_int64)), ("local", C.POINTER(C.c_int64)),
                ("leafA_off", C.POINTER(C.c_int64)), ("m_l", C.POINTER(C.c_int64)), ("w_l", C.POINTER(C.c_int64)),
                ("leaf_left", C.POINTER(C.c_int64)),
                ("chainD_off", C.POINTER(C.c_int64)), ("nzb", C.POINTER(C.c_int64)), ("nzb_next", C.POINTER(C.c_int64)),
                ("wc", C.POINTER(C.c_int64)), ("last_of_phase", C.POINTER(C.c_uint8))]


def build_tables(engine, ineq_rows, fixed_v, row_scale, group=None, positions: str = "library", cuts=None,
                 _parts: dict | None = None, _layout_only: bool = False) -> KktTables:
    """``ineq_rows``: constraint rows with a slack (in order); ``fixed_v`` [n + ns]: primal unknowns held fixed;
    ``row_scale`` [m]: the solver's constraint-row scaling (multiplies G~ row-wise); ``group``: mesh sections per leaf
    (int or one per phase; default ``default_group``); ``cuts``: per phase the nodes at which the chain is cut
    (``_node_maps``; ``n_phase`` of the result then counts chain segments).  ``_parts``: filled with the classification
    and the entry list, for ``kkt_sharded``."""
    lay, model = engine.layout, engine.model
    if group is None:
        group = default_group(engine, ineq_rows)
    elif np.isscalar(group):
        group = [int(group)] * len(lay.phases)
    n, m = engine.num_x, engine.num_c
    ineq_rows = np.asarray(ineq_rows, dtype=np.int64)
    ns = len(ineq_rows)
    nv, nu = n + ns, n + ns + m
    fixed = np.zeros(nu, bool)
    fixed[:nv] = np.asarray(fixed_v, bool)
    maps = _node_maps(engine, group, cuts)
    chain_base = np.concatenate([[0], np.cumsum([mp[7] for mp in maps])]).astype(np.int64)   # first chain node of a phase
    leaf_phase_ptr = np.concatenate([[0], np.cumsum([len(mp[0]) - 1 for mp in maps])]).astype(np.int64)
    n_chain, n_leaf = int(chain_base[-1]), int(leaf_phase_ptr[-1])
    # the chain's independent pieces: a phase, or with cuts a segment of one (what the tables call a phase of the chain)
    chain_phase_ptr = np.concatenate([chain_base[ip] + mp[8][:-1] for ip, mp in enumerate(maps)] + [chain_base[-1:]]).astype(np.int64)
    n_phase = len(chain_phase_ptr) - 1

    cls = np.full(nu, BORDER, np.int8)
    blk = np.zeros(nu, np.int64)
    key_node = np.zeros(nu, np.int64)      # ordering inside a block: primal before dual, then node, kind, index
    key_kind = np.zeros(nu, np.int64)
    key_idx = np.arange(nu, dtype=np.int64)
    dual = np.zeros(nu, bool)
    dual[nv:] = True

    u_phase = np.full(nu, -1, np.int64)     # the phase and node an unknown sits on (-1: none)
    u_node = np.full(nu, -1, np.int64)

    def place_nodes(u, ip, nodes):
        u_phase[u], u_node[u] = ip, nodes
        s, N, is_b, sec, is_cut, chain_id = maps[ip][:6]
        b = is_b[nodes]
        k = sec[nodes]
        cls[u] = np.where(b, np.where(is_cut[k], BORDER, CHAIN), LEAF)
        blk[u] = np.where(b, chain_base[ip] + chain_id[k], leaf_phase_ptr[ip] + k)
        key_node[u] = nodes

    row_slack = np.full(m, -1, np.int64)
    row_slack[ineq_rows] = np.arange(ns)
    for ip, (pl, pm) in enumerate(zip(lay.phases, model.phases)):
        N = maps[ip][1]
        nz = pm.n_z
        u = pl.x_off + np.arange(nz * N, dtype=np.int64)
        place_nodes(u, ip, (u - pl.x_off) % N)
        key_kind[u] = 0
        # defect rows: the row of node i >= 1 goes with node i
        for a in range(pm.n_y):
            rows = pl.c_off + a * (N - 1) + np.arange(N - 1, dtype=np.int64)
            place_nodes(nv + rows, ip, np.arange(1, N, dtype=np.int64))
            key_kind[nv + rows] = 1
        for mm in range(pm.n_p):
            rows = pl.c_path_off + mm * N + np.arange(N, dtype=np.int64)
            place_nodes(nv + rows, ip, np.arange(N, dtype=np.int64))
            key_kind[nv + rows] = 0
            sl = row_slack[rows]
            has = sl >= 0
            if has.any():
                place_nodes(n + sl[has], ip, np.arange(N, dtype=np.int64)[has])
                key_kind[n + sl[has]] = 1
    # endpoint Hessian terms that couple two different nodes: both variables move to the border
    hr, hc = (np.asarray(a, np.int64) for a in engine.evaluate_H_structure())
    both_node = (cls[hr] != BORDER) & (cls[hc] != BORDER)
    cross = both_node & ((cls[hr] != cls[hc]) | (blk[hr] != blk[hc]))
    promoted = np.unique(np.concatenate([hr[cross], hc[cross]]))
    cls[promoted] = BORDER
    blk[cls == BORDER] = 0
    key_node[cls == BORDER] = 0
    leaf_left = np.concatenate([chain_base[ip] + mp[6] for ip, mp in enumerate(maps)]).astype(np.int64) \
        if n_leaf else np.zeros(0, np.int64)
    jr, jc = (np.asarray(a, np.int64) for a in engine.evaluate_G_structure())
    if _parts is not None:
        _parts.update(cls=cls.copy(), blk=blk.copy(), key_node=key_node.copy(), key_kind=key_kind.copy(), dual=dual.copy(),
                      fixed=fixed.copy(), n=n, m=m, ns=ns, nv=nv, nu=nu, n_leaf=n_leaf, n_chain=n_chain,
                      chain_phase_ptr=chain_phase_ptr.copy(), leaf_left=leaf_left.copy(), maps=maps, chain_base=chain_base,
                      leaf_phase_ptr=leaf_phase_ptr, hr=hr, hc=hc, jr=jr, jc=jc, group=list(group), u_phase=u_phase, u_node=u_node)
    if _layout_only:     # (block order and value-buffer layout without the entry tables: what kkt_sharded needs of the whole plan)
        e0 = (np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0, np.int32), np.zeros(0, np.int64), np.zeros(0))
        return _finish("positions", n, nv, nu, ns, m, cls, blk, key_node, key_kind, key_idx, dual, fixed, n_leaf, n_chain,
                       chain_phase_ptr, leaf_left, hr, hc, jr, jc, row_scale, ineq_rows, entries=e0)
    return _finish(positions, n, nv, nu, ns, m, cls, blk, key_node, key_kind, key_idx, dual, fixed, n_leaf, n_chain,
                   chain_phase_ptr, leaf_left, hr, hc, jr, jc, row_scale, ineq_rows)


def natural_entries(n, nv, hr, hc, jr, jc, row_scale, ineq_rows):
    """Lower-triangle entries of K over natural unknowns as (row, column, source kind, source index, coefficient):
    H~, the row-scaled G~, the -1 of every slack."""
    ns = len(ineq_rows)
    eu = np.concatenate([hr, nv + jr, nv + ineq_rows])
    ev = np.concatenate([hc, jc, n + np.arange(ns, dtype=np.int64)])
    ekind = np.concatenate([np.full(len(hr), SRC_H), np.full(len(jr), SRC_G), np.full(ns, SRC_ONE)]).astype(np.int32)
    eidx = np.concatenate([np.arange(len(hr)), np.arange(len(jr)), np.zeros(ns, np.int64)]).astype(np.int64)
    ecoef = np.concatenate([np.ones(len(hr)), np.asarray(row_scale, float)[jr], -np.ones(ns)])
    return eu, ev, ekind, eidx, ecoef


def _finish(positions, n, nv, nu, ns, m, cls, blk, key_node, key_kind, key_idx, dual, fixed, n_leaf, n_chain,
            chain_phase_ptr, leaf_left, hr, hc, jr, jc, row_scale, ineq_rows, entries=None, n_primal=None, n_dual=None,
            chain_export=None):
    """Block order, value-buffer layout and entry tables of a classified system.  ``entries``: the lower-triangle entries
    (``natural_entries`` form, fixed unknowns already dropped) when the system is not a whole NLP's (a rank's part of a
    sharded factorisation: ``positions`` must then be "positions" or "numpy")."""
    n_phase = len(chain_phase_ptr) - 1
    n_primal = nv if n_primal is None else n_primal
    n_dual = m if n_dual is None else n_dual
    # block order
    order = np.lexsort((key_idx, key_kind, key_node, dual, blk, cls))
    perm = order.astype(np.int64)
    counts_leaf = np.bincount(blk[cls == LEAF], minlength=n_leaf) if n_leaf else np.zeros(0, np.int64)
    counts_chain = np.bincount(blk[cls == CHAIN], minlength=n_chain)
    leaf_ptr = np.concatenate([[0], np.cumsum(counts_leaf)]).astype(np.int64)
    chain_ptr = np.concatenate([[0], np.cumsum(counts_chain)]).astype(np.int64)
    nb = int(np.sum(cls == BORDER))
    local = np.empty(nu, np.int64)          # index inside its block
    pos = np.empty(nu, np.int64)
    pos[perm] = np.arange(nu)
    base_leaf, base_chain = 0, int(leaf_ptr[-1])
    base_border = base_chain + int(chain_ptr[-1])
    is_leaf, is_chain, is_border = cls == LEAF, cls == CHAIN, cls == BORDER
    local[is_leaf] = pos[is_leaf] - base_leaf - leaf_ptr[blk[is_leaf]]
    local[is_chain] = pos[is_chain] - base_chain - chain_ptr[blk[is_chain]]
    local[is_border] = pos[is_border] - base_border

    nzb = counts_chain.astype(np.int64)
    last_of_phase = np.zeros(n_chain, bool)
    last_of_phase[chain_phase_ptr[1:] - 1] = True
    nzb_next = np.where(last_of_phase, 0, np.concatenate([nzb[1:], [0]]))
    m_l = counts_leaf.astype(np.int64)
    w_l = (nzb[leaf_left] + nzb[leaf_left + 1] + nb) if n_leaf else np.zeros(0, np.int64)
    # value buffer layout
    sizeA = m_l * (m_l + w_l)
    sizeS = w_l * w_l
    leafA_off = np.concatenate([[0], np.cumsum(sizeA)])[:-1] if n_leaf else np.zeros(0, np.int64)
    o = int(np.sum(sizeA))
    leafS_off = o + (np.concatenate([[0], np.cumsum(sizeS)])[:-1] if n_leaf else np.zeros(0, np.int64))
    o += int(np.sum(sizeS))
    wc = nzb_next + nb
    sizeD = nzb * (nzb + wc)
    chainD_off = o + np.concatenate([[0], np.cumsum(sizeD)])[:-1]
    o += int(np.sum(sizeD))
    chainS_off = o + np.concatenate([[0], np.cumsum(wc * wc)])[:-1]
    o += int(np.sum(wc * wc))
    border_off = o
    total = o + nb * nb

    # ---- matrix entries (natural unknown pairs), lower triangle of K ------------------------------------------
    def make_plan():
        """The elimination plan as the C structure ``pc_kkt_plan`` (the arrays are kept alive by the caller)."""
        from .engine import load_library
        lib = load_library()
        lib.pc_kkt_plan_positions.argtypes = [C.POINTER(_Plan), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.pc_kkt_plan_entries.argtypes = [C.POINTER(_Plan), C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p] + [C.c_void_p] * 11
        lib.pc_kkt_last_error.restype = C.c_char_p
        keep = [np.ascontiguousarray(a, dtype=t) for a, t in (
            (cls, np.int8), (blk, np.int64), (local, np.int64), (leafA_off, np.int64), (m_l, np.int64), (w_l, np.int64),
            (leaf_left, np.int64), (chainD_off, np.int64), (nzb, np.int64), (nzb_next, np.int64), (wc, np.int64),
            (last_of_phase, np.uint8))]
        keep = [a if a.size else np.zeros(1, a.dtype) for a in keep]
        P = _Plan()
        P.nu, P.nb, P.border_off = int(nu), int(nb), int(border_off)
        for name, a in zip(("cls", "blk", "local", "leafA_off", "m_l", "w_l", "leaf_left", "chainD_off", "nzb", "nzb_next",
                            "wc", "last_of_phase"), keep):
            ctype = {np.dtype(np.int8): C.c_int8, np.dtype(np.int64): C.c_int64, np.dtype(np.uint8): C.c_uint8}[a.dtype]
            setattr(P, name, a.ctypes.data_as(C.POINTER(ctype)))
        return lib, P, keep

    if positions == "library":
        if entries is not None:
            raise ValueError("explicit entries need positions='positions' or 'numpy'")
        # the entry tables in one pass of host C++ (pc_kkt_plan_entries): the NumPy statement below builds them from a
        # dozen entry-sized temporaries, sorts twice and gathers nine times -- 150 ms for 15 k nodes, and several times
        # that whenever the allocator has to fault the temporaries in afresh, which inside a solve is every time
        lib, P, _keepalive = make_plan()
        hr_c, hc_c, jr_c, jc_c = (np.ascontiguousarray(a, dtype=np.int64) for a in (hr, hc, jr, jc))
        rs_c = np.ascontiguousarray(row_scale, dtype=np.float64)
        iq_c = np.ascontiguousarray(ineq_rows, dtype=np.int64)
        fx_c = np.ascontiguousarray(fixed, dtype=np.uint8)
        counts = np.zeros(3, np.int64)

        def call(*outs):
            ok = lib.pc_kkt_plan_entries(C.byref(P), int(n), int(nv), len(hr_c), hr_c.ctypes.data, hc_c.ctypes.data, len(jr_c),
                                         jr_c.ctypes.data, jc_c.ctypes.data, rs_c.ctypes.data, int(ns), iq_c.ctypes.data,
                                         fx_c.ctypes.data, counts.ctypes.data, *[o.ctypes.data if o is not None else None for o in outs])
            if not ok:
                raise RuntimeError(lib.pc_kkt_last_error().decode())
        # one call with outputs sized for the most there can be (np.empty touches no page), trimmed afterwards
        cap = len(hr_c) + len(jr_c) + int(ns)
        dst, run_ptr = np.empty(cap, np.int64), np.empty(cap + 1, np.int64)
        src_kind, src_idx, src_coef = np.empty(cap, np.int32), np.empty(cap, np.int32), np.empty(cap, np.float64)
        mv_ptr, mv_col = np.empty(nu + 1, np.int64), np.empty(2 * cap, np.int32)
        mv_kind, mv_idx, mv_coef = np.empty(2 * cap, np.int32), np.empty(2 * cap, np.int32), np.empty(2 * cap, np.float64)
        call(dst, run_ptr, src_kind, src_idx, src_coef, mv_ptr, mv_col, mv_kind, mv_idx, mv_coef)
        n_src, n_dst, n_mv = (int(c) for c in counts)
        dst, run_ptr = dst[:n_dst], run_ptr[:n_dst + 1]
        src_kind, src_idx, src_coef = src_kind[:n_src], src_idx[:n_src], src_coef[:n_src]
        mv_col, mv_kind, mv_idx, mv_coef = mv_col[:n_mv], mv_kind[:n_mv], mv_idx[:n_mv], mv_coef[:n_mv]
        ar = np.arange(nu, dtype=np.int64)
        diag_pos = np.empty(nu, np.int64)
        if not lib.pc_kkt_plan_positions(C.byref(P), nu, ar.ctypes.data, ar.ctypes.data, diag_pos.ctypes.data):
            raise RuntimeError(lib.pc_kkt_last_error().decode())
        return KktTables(
            nu=nu, nv=nv, n_leaf=n_leaf, n_chain=n_chain, n_phase=n_phase, nb=nb, n_primal=n_primal, n_dual=n_dual,
            perm=perm, leaf_ptr=leaf_ptr, chain_ptr=chain_ptr, chain_phase_ptr=chain_phase_ptr, leaf_left=leaf_left,
            leafA_off=np.asarray(leafA_off, np.int64), leafS_off=np.asarray(leafS_off, np.int64),
            chainD_off=np.asarray(chainD_off, np.int64), chainS_off=np.asarray(chainS_off, np.int64),
            border_off=int(border_off), total_vals=int(total),
            dst=dst, run_ptr=run_ptr, src_kind=src_kind, src_idx=src_idx, src_coef=src_coef, diag_pos=diag_pos,
            fixed=fixed.astype(np.uint8), mv_ptr=mv_ptr, mv_col=mv_col, mv_kind=mv_kind, mv_idx=mv_idx, mv_coef=mv_coef, chain_export=chain_export)

    # ---- the same in NumPy: the statement of the rule (positions = "numpy" / "positions"; the CPU tests hold the library
    #      against it).  "positions" takes only the position rule from the library, as round 3's first version did.
    if entries is None:
        eu, ev, ekind, eidx, ecoef = natural_entries(n, nv, hr, hc, jr, jc, row_scale, ineq_rows)
        keep = ~(fixed[eu] | fixed[ev])
        eu, ev, ekind, eidx, ecoef = eu[keep], ev[keep], ekind[keep], eidx[keep], ecoef[keep]
    else:
        eu, ev, ekind, eidx, ecoef = entries

    def dest_library(u, v):
        """The same rule as ``dest_numpy`` below in one pass of host C++ (``pc_kkt_plan_positions``): the vectorised form
        allocates ~100 temporaries of the entry count each, and their first-touch page faults were most of a table
        build inside a solve (95 of 110 ms at config 2)."""
        lib, P, _keepalive = make_plan()
        u = np.ascontiguousarray(u, dtype=np.int64)
        v = np.ascontiguousarray(v, dtype=np.int64)
        out = np.empty(len(u), np.int64)
        if not lib.pc_kkt_plan_positions(C.byref(P), len(u), u.ctypes.data, v.ctypes.data, out.ctypes.data):
            raise RuntimeError(lib.pc_kkt_last_error().decode())
        return out

    def dest_numpy(u, v):
        """Position in the value buffer of K[u, v] (u, v natural), vectorised; -1 where the pair has no place.  (The
        statement of the rule; ``positions="numpy"`` selects it, the CPU tests hold the library against it.)"""
        cu, cv = cls[u], cls[v]
        # order the pair so that `a` is the one eliminated first: leaf < chain < border; inside a class lower block first
        swap = (cu > cv) | ((cu == cv) & (blk[u] > blk[v])) | ((cu == cv) & (blk[u] == blk[v]) & (local[u] < local[v]))
        a, b = np.where(swap, v, u), np.where(swap, u, v)
        ca, cb, ba, bb, la, lb = cls[a], cls[b], blk[a], blk[b], local[a], local[b]
        out = np.full(len(u), -1, np.int64)
        # leaf x leaf (same leaf): lower triangle of A (a has the larger local index after the swap rule above)
        k = (ca == LEAF) & (cb == LEAF) & (ba == bb)
        out[k] = leafA_off[ba[k]] + la[k] * (m_l[ba[k]] + w_l[ba[k]]) + lb[k]
        # leaf x chain
        k = (ca == LEAF) & (cb == CHAIN)
        left = leaf_left[ba[k]]
        col = np.where(bb[k] == left, lb[k], np.where(bb[k] == left + 1, nzb[left] + lb[k], -1))
        ok = col >= 0
        tmp = np.full(int(k.sum()), -1, np.int64)
        tmp[ok] = leafA_off[ba[k]][ok] + la[k][ok] * (m_l[ba[k]] + w_l[ba[k]])[ok] + m_l[ba[k]][ok] + col[ok]
        out[k] = tmp
        # leaf x border
        k = (ca == LEAF) & (cb == BORDER)
        left = leaf_left[ba[k]]
        out[k] = leafA_off[ba[k]] + la[k] * (m_l[ba[k]] + w_l[ba[k]]) + m_l[ba[k]] + nzb[left] + nzb[left + 1] + lb[k]
        # chain x chain
        k = (ca == CHAIN) & (cb == CHAIN) & (ba == bb)
        out[k] = chainD_off[ba[k]] + la[k] * (nzb[ba[k]] + wc[ba[k]]) + lb[k]
        k = (ca == CHAIN) & (cb == CHAIN) & (bb == ba + 1) & ~last_of_phase[ba]
        out[k] = chainD_off[ba[k]] + la[k] * (nzb[ba[k]] + wc[ba[k]]) + nzb[ba[k]] + lb[k]
        # chain x border
        k = (ca == CHAIN) & (cb == BORDER)
        out[k] = chainD_off[ba[k]] + la[k] * (nzb[ba[k]] + wc[ba[k]]) + nzb[ba[k]] + nzb_next[ba[k]] + lb[k]
        # border x border, lower
        k = (ca == BORDER) & (cb == BORDER)
        out[k] = border_off + la[k] * nb + lb[k]
        return out

    dest = {"positions": dest_library, "numpy": dest_numpy}[positions]
    d = dest(eu, ev)
    if np.any(d < 0):
        bad = np.nonzero(d < 0)[0][0]
        raise RuntimeError(f"KKT entry ({eu[bad]}, {ev[bad]}) couples two blocks the elimination order keeps apart")
    so = np.argsort(d, kind="stable")
    d_sorted = d[so]
    first = np.concatenate([[True], d_sorted[1:] != d_sorted[:-1]]) if len(d) else np.zeros(0, bool)
    dst = d_sorted[first]
    run_ptr = np.concatenate([np.nonzero(first)[0], [len(d_sorted)]]).astype(np.int64)
    diag_pos = dest(np.arange(nu, dtype=np.int64), np.arange(nu, dtype=np.int64))

    # ---- full symmetric CSR over natural unknowns (products) ----------------------------------------------------
    off = eu != ev
    ru = np.concatenate([eu, ev[off]])
    rv = np.concatenate([ev, eu[off]])
    rk = np.concatenate([ekind, ekind[off]])
    ri = np.concatenate([eidx, eidx[off]])
    rc = np.concatenate([ecoef, ecoef[off]])
    # row-major order of the entries, columns ascending inside a row: one sort of the combined key (no pair occurs
    # twice -- checked -- so the order is unique; np.lexsort over the two keys took 60-90 of the 150 ms of a 15 k-node
    # build, a merge sort of the one int64 key -- the entries arrive as a few long sorted runs -- 17)
    key = ru * np.int64(nu) + rv
    o2 = np.argsort(key, kind="stable")
    ks = key[o2]
    if len(ks) > 1 and np.any(ks[1:] == ks[:-1]):
        raise RuntimeError("a KKT entry occurs twice in the symmetric expansion")
    mv_ptr = np.concatenate([[0], np.cumsum(np.bincount(ru, minlength=nu))]).astype(np.int64)

    return KktTables(
        nu=nu, nv=nv, n_leaf=n_leaf, n_chain=n_chain, n_phase=n_phase, nb=nb, n_primal=n_primal, n_dual=n_dual,
        perm=perm, leaf_ptr=leaf_ptr, chain_ptr=chain_ptr, chain_phase_ptr=chain_phase_ptr, leaf_left=leaf_left,
        leafA_off=np.asarray(leafA_off, np.int64), leafS_off=np.asarray(leafS_off, np.int64),
        chainD_off=np.asarray(chainD_off, np.int64), chainS_off=np.asarray(chainS_off, np.int64),
        border_off=int(border_off), total_vals=int(total),
        dst=dst.astype(np.int64), run_ptr=run_ptr, src_kind=ekind[so].astype(np.int32), src_idx=eidx[so].astype(np.int32),
        src_coef=ecoef[so].astype(np.float64), diag_pos=diag_pos.astype(np.int64), fixed=fixed.astype(np.uint8),
        mv_ptr=mv_ptr, mv_col=rv[o2].astype(np.int32), mv_kind=rk[o2].astype(np.int32), mv_idx=ri[o2].astype(np.int32),
        mv_coef=rc[o2].astype(np.float64), chain_export=chain_export)


def export_shapes(T: KktTables):
    """(chain node, its unknowns, unknowns of the exported last node of its segment it is coupled to) for every exported
    chain node, ascending: the layout of ``pc_kkt_export_panels``."""
    if T.chain_export is None:
        return []
    nzb = np.diff(T.chain_ptr)
    out = []
    for c in np.nonzero(T.chain_export)[0]:
        seg = int(np.searchsorted(T.chain_phase_ptr, c, side="right") - 1)
        last = int(T.chain_phase_ptr[seg + 1] - 1)
        nr = int(nzb[last]) if (c == T.chain_phase_ptr[seg] and last != c and T.chain_export[last]) else 0
        out.append((int(c), int(nzb[c]), nr))
    return out


class _Desc(C.Structure):
    _i64p, _i32p, _f64p, _u8p = C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_uint8)
    _fields_ = ([(k, C.c_int64) for k in ("nu", "nv", "n_leaf", "n_chain", "n_phase", "nb", "total_vals", "border_off",
                                          "n_dst", "n_src", "n_mv")]
                + [(k, C.POINTER(C.c_int64)) for k in ("perm", "leaf_ptr", "chain_ptr", "chain_phase_ptr", "leaf_left",
                                                       "leafA_off", "leafS_off", "chainD_off", "chainS_off", "dst", "run_ptr")]
                + [("src_kind", C.POINTER(C.c_int32)), ("src_idx", C.POINTER(C.c_int32)), ("src_coef", C.POINTER(C.c_double)),
                   ("diag_pos", C.POINTER(C.c_int64)), ("fixed", C.POINTER(C.c_uint8)), ("mv_ptr", C.POINTER(C.c_int64)),
                   ("mv_col", C.POINTER(C.c_int32)), ("mv_kind", C.POINTER(C.c_int32)), ("mv_idx", C.POINTER(C.c_int32)),
                   ("mv_coef", C.POINTER(C.c_double)), ("chain_export", C.POINTER(C.c_uint8))])


class GpuKkt:
    """The factorisation object: ``pc_kkt_*`` bound to one engine's device-resident G~ / H~."""

    def __init__(self, engine, ineq_rows, fixed_v, row_scale, group=None, tables=None, d_jac=None, d_hess=None):
        """``tables``: ready-made tables instead of the whole NLP's (a rank's part of a sharded factorisation,
        kkt_sharded.py); ``d_jac`` / ``d_hess``: device addresses of the G~ / H~ values to read instead of the engine's."""
        from .engine import load_library
        if engine.device < 0:
            raise RuntimeError("the KKT solver needs a GPU engine; pycollo_amd has no CPU fallback")
        self.engine = engine
        import time
        t0 = time.perf_counter()
        self.tables = T = tables if tables is not None else build_tables(engine, ineq_rows, fixed_v, row_scale, group)
        self.seconds_tables = time.perf_counter() - t0          # host: the elimination plan as index tables
        self._lib = lib = load_library()
        vp = C.c_void_p
        lib.pc_kkt_last_error.restype = C.c_char_p
        lib.pc_kkt_create.argtypes = [C.POINTER(_Desc), vp, vp, C.c_int, C.POINTER(vp)]
        lib.pc_kkt_destroy.argtypes = [vp]
        lib.pc_kkt_destroy.restype = None
        lib.pc_kkt_factor.argtypes = [vp, C.c_int, vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        lib.pc_kkt_solve.argtypes = [vp, vp, vp]
        lib.pc_kkt_matvec.argtypes = [vp, C.c_int, vp, vp, vp]
        lib.pc_kkt_solve_refined.argtypes = [vp, C.c_int, vp, vp, C.c_int, vp, C.POINTER(C.c_int32)]
        lib.pc_device_results.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
        lib.pc_kkt_factor_partial.argtypes = [vp, C.c_int, vp, vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        lib.pc_kkt_border_load_factor.argtypes = [vp, vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        lib.pc_kkt_forward_partial.argtypes = [vp, vp, vp]
        lib.pc_kkt_backward_partial.argtypes = [vp, vp, vp]
        lib.pc_kkt_export_panels.argtypes = [vp, vp]
        lib.pc_kkt_export_rhs.argtypes = [vp, vp]
        lib.pc_kkt_import_solution.argtypes = [vp, vp]
        d = _Desc()
        self._keep = []
        for k in ("nu", "nv", "n_leaf", "n_chain", "n_phase", "nb", "total_vals", "border_off"):
            setattr(d, k, int(getattr(T, k)))
        d.n_dst, d.n_src, d.n_mv = len(T.dst), len(T.src_kind), len(T.mv_col)
        for k, typ in (("perm", np.int64), ("leaf_ptr", np.int64), ("chain_ptr", np.int64), ("chain_phase_ptr", np.int64),
                       ("leaf_left", np.int64), ("leafA_off", np.int64), ("leafS_off", np.int64), ("chainD_off", np.int64),
                       ("chainS_off", np.int64), ("dst", np.int64), ("run_ptr", np.int64), ("src_kind", np.int32),
                       ("src_idx", np.int32), ("src_coef", np.float64), ("diag_pos", np.int64), ("fixed", np.uint8),
                       ("mv_ptr", np.int64), ("mv_col", np.int32), ("mv_kind", np.int32), ("mv_idx", np.int32),
                       ("mv_coef", np.float64)):
            arr = np.ascontiguousarray(getattr(T, k), dtype=typ)
            if arr.size == 0:
                arr = np.zeros(1, dtype=typ)
            self._keep.append(arr)
            ctype = {np.int64: C.c_int64, np.int32: C.c_int32, np.float64: C.c_double, np.uint8: C.c_uint8}[typ]
            setattr(d, k, arr.ctypes.data_as(C.POINTER(ctype)))
        if T.chain_export is not None and np.any(T.chain_export):
            ce = np.ascontiguousarray(T.chain_export, dtype=np.uint8)
            self._keep.append(ce)
            d.chain_export = ce.ctypes.data_as(C.POINTER(C.c_uint8))
        self._export_shapes = export_shapes(T)
        dg, dj, dh = vp(), vp(), vp()
        if not lib.pc_device_results(engine._h, C.byref(dg), C.byref(dj), C.byref(dh)):
            raise RuntimeError(lib.pc_last_error().decode())
        if d_jac is not None:
            dj = vp(int(d_jac))
        if d_hess is not None:
            dh = vp(int(d_hess))
        self._h = vp()
        if not lib.pc_kkt_create(C.byref(d), dj, dh, int(engine.device), C.byref(self._h)):
            raise RuntimeError("pc_kkt_create failed: " + lib.pc_kkt_last_error().decode())
        self.nu = T.nu
        self.seconds_create = time.perf_counter() - t0 - self.seconds_tables   # descriptor + device allocation / upload

    def _check(self, ok):
        if not ok:
            raise RuntimeError(self._lib.pc_kkt_last_error().decode())

    def factor(self, dvec, use_hess=True):
        """Assemble from the engine's current device G~ / H~ and factorise; returns (n_pos, n_neg) pivots."""
        dvec = np.ascontiguousarray(dvec, dtype=np.float64)
        p, q = C.c_int32(), C.c_int32()
        self._check(self._lib.pc_kkt_factor(self._h, int(bool(use_hess)), dvec.ctypes.data, C.byref(p), C.byref(q)))
        return p.value, q.value

    def solve(self, rhs):
        rhs = np.ascontiguousarray(rhs, dtype=np.float64)
        x = np.empty(self.nu)
        self._check(self._lib.pc_kkt_solve(self._h, rhs.ctypes.data, x.ctypes.data))
        return x

    def matvec(self, dvec, x, use_hess=True):
        dvec = np.ascontiguousarray(dvec, dtype=np.float64)
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty(self.nu)
        self._check(self._lib.pc_kkt_matvec(self._h, int(bool(use_hess)), dvec.ctypes.data, x.ctypes.data, y.ctypes.data))
        return y

    def solve_refined(self, rhs, dvec_true, use_hess=True, max_steps=3):
        """``K^-1 rhs`` with the current factors, iteratively refined on the device against the system with
        ``dvec_true`` on its diagonal (``pc_kkt_solve_refined``); returns (x, back-substitutions performed)."""
        rhs = np.ascontiguousarray(rhs, dtype=np.float64)
        dvec_true = np.ascontiguousarray(dvec_true, dtype=np.float64)
        x = np.empty(self.nu)
        n = C.c_int32()
        self._check(self._lib.pc_kkt_solve_refined(self._h, int(bool(use_hess)), dvec_true.ctypes.data, rhs.ctypes.data,
                                                   int(max_steps), x.ctypes.data, C.byref(n)))
        return x, n.value

    # ---- a rank's part of a factorisation cut across ranks (kkt_sharded.py) ------------------------------------------
    def factor_partial(self, dvec, use_hess=True):
        """Assemble, eliminate the leaves and the chain, and return the border block with every Schur complement added
        but *not* factorised ([nb, nb], lower triangle valid), plus the (positive, negative) pivots so far."""
        dvec = np.ascontiguousarray(dvec, dtype=np.float64)
        nb = self.tables.nb
        B = np.zeros((nb, nb))
        p, q = C.c_int32(), C.c_int32()
        self._check(self._lib.pc_kkt_factor_partial(self._h, int(bool(use_hess)), dvec.ctypes.data, B.ctypes.data,
                                                    C.byref(p), C.byref(q)))
        return B, p.value, q.value

    def border_load_factor(self, B):
        """Take the border block as given (lower triangle read) and factorise it; returns its pivot counts."""
        B = np.ascontiguousarray(B, dtype=np.float64)
        if B.shape != (self.tables.nb, self.tables.nb):
            raise ValueError("border block of the wrong shape")
        p, q = C.c_int32(), C.c_int32()
        self._check(self._lib.pc_kkt_border_load_factor(self._h, B.ctypes.data, C.byref(p), C.byref(q)))
        return p.value, q.value

    def forward_partial(self, rhs):
        """Forward elimination through leaves and chain; returns the border's right-hand side minus what they owe it."""
        rhs = np.ascontiguousarray(rhs, dtype=np.float64)
        rb = np.empty(self.tables.nb)
        self._check(self._lib.pc_kkt_forward_partial(self._h, rhs.ctypes.data, rb.ctypes.data))
        return rb

    def backward_partial(self, xb):
        """Back-substitution from the given border solution (block order); returns the whole local solution."""
        xb = np.ascontiguousarray(xb, dtype=np.float64)
        if xb.shape != (self.tables.nb,):
            raise ValueError("border solution of the wrong length")
        x = np.empty(self.nu)
        self._check(self._lib.pc_kkt_backward_partial(self._h, xb.ctypes.data, x.ctypes.data))
        return x

    def export_panels(self):
        """After ``factor_partial``: the assembled panels [D | K(node, exported last node) | F] of the exported chain nodes,
        in ascending node order (``export_shapes``)."""
        out = np.zeros(sum(nz * (nz + nr + self.tables.nb) for _, nz, nr in self._export_shapes) or 1)
        self._check(self._lib.pc_kkt_export_panels(self._h, out.ctypes.data))
        panels, o = [], 0
        for _, nz, nr in self._export_shapes:
            n = nz * (nz + nr + self.tables.nb)
            panels.append(out[o:o + n].reshape(nz, nz + nr + self.tables.nb))
            o += n
        return panels

    def export_rhs(self):
        """After ``forward_partial``: the exported nodes' right-hand sides minus what the eliminated blocks owe them."""
        out = np.zeros(sum(nz for _, nz, _ in self._export_shapes) or 1)
        self._check(self._lib.pc_kkt_export_rhs(self._h, out.ctypes.data))
        return out[:sum(nz for _, nz, _ in self._export_shapes)]

    def import_solution(self, x):
        """Before ``backward_partial``: the exported nodes' solution (concatenated in node order)."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        if x.shape != (sum(nz for _, nz, _ in self._export_shapes),):
            raise ValueError("exported solution of the wrong length")
        if x.size:
            self._check(self._lib.pc_kkt_import_solution(self._h, x.ctypes.data))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.pc_kkt_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
