"""The KKT factorisation cut across ranks: the consumer that needs only a rank's own rows.

``sharding.ShardedNlp`` splits the evaluation of c~, G~, H~ by contiguous section ranges, but a solver that lives on one
rank needs all of it gathered (0.47 ms of ring all-gather against a 20 us evaluation at config 4).  The block elimination
of ``kkt.py`` is already local in the mesh: a leaf reads the G~ rows and H~ node blocks of its own sections only.  Here the
chain of every phase is **cut where the evaluation is cut**:

  rank r      eliminates the leaves and the chain segments of its own section ranges from the G~ / H~ entries its own tile
              kernels wrote (plus the handful the tail kernel writes, which every rank has after the exchange of the
              per-tile partial sums -- a few doubles per tile, never the rows);
  cut nodes   the node two ranks share (first node of a rank's range) -- its variables, path slacks / multipliers and the
              defect multipliers of the rows that end on it -- is a chain node of BOTH ranks' segments (the last of the
              one, the first of the other) and is eliminated by NEITHER (``KktTables.chain_export``): the cyclic reduction
              of a segment runs between its two ends (an exported last node is the right separator of every node whose own
              would lie beyond it), and the ends' assembled panels [D | K(first, last) | F] are what the rank adds to the
              reduced system.  A rank's border is therefore the NLP's own (integrals, times, parameters, endpoint rows) and
              its blocks are exactly as wide as the unsharded plan's.  (``ends="border"``: the first version -- the shared
              nodes in the rank's local border, dense in every block; kept for the unchanged kernels and the NumPy oracle.)
  reduced     every rank's terms -- its Schur complement on the border and its exported panels -- are added into the system
  system      of all border and shared unknowns (one all-reduce of nb_red^2 doubles -- (8 ranks x 4 phases x ~20)^2 at the
              largest configuration here), which every rank then factorises redundantly: a dense block of a few hundred.

A solve is the same in three steps: local forward elimination, one all-reduce of the reduced right-hand side (nb_red
doubles), the reduced solve, local back-substitution.  The pivot signs of all local blocks and of the reduced system add up
to the inertia (Sylvester), so the interior-point method's regularisation loop is unchanged.  The reference has no
counterpart: IPOPT hands the whole matrix to MUMPS on one process (pycollo/backend.py:1703-1711).

``ShardedKktPlan`` is pure NumPy (tested on CPU against a general sparse solver through ``oracle/ref_kkt.py``);
``ShardedKkt`` runs one rank on its GPU (``pc_kkt_factor_partial`` / ``_forward_partial`` / ``_backward_partial`` /
``pc_kkt_border_load_factor``) and moves the two small reductions with ``torch.distributed``.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import kkt
from .kkt import BORDER, CHAIN, LEAF, SRC_G, SRC_H, KktTables


@dataclass
class RankTables:
    tables: KktTables        # the rank's system: its leaves, its chain segments, its local border
    univ: np.ndarray         # [nu_local] natural index (whole NLP's KKT numbering) of every local unknown, ascending
    own: np.ndarray          # [nu_local] bool: this rank supplies the unknown's diagonal / right-hand side, reports its solution
    border_red: np.ndarray   # [nb_local] position in the reduced system of every local border unknown (block order)
    export_red: list = None         # per exported chain node (ascending): reduced positions of its unknowns (block order)
    export_anchor_red: list = None  # ... and of the exported last node of its segment it is coupled to (empty: none)


def shard_cuts(engine, shard_plan):
    """Per phase: the nodes at which ``shard_plan`` cuts the mesh, and the rank of every segment between them."""
    cuts, seg_rank = [], []
    for ip, mesh in enumerate(engine.meshes):
        s = np.asarray(mesh.s, dtype=np.int64)
        k0s = np.asarray(shard_plan.tiles[ip][0], dtype=np.int64)
        c, sr = [], []
        for r in range(shard_plan.world):
            tb, te = shard_plan.tile_ranges[r][ip]
            if te <= tb:
                continue
            if sr:
                c.append(int(s[k0s[tb]]))
            sr.append(r)
        cuts.append(c)
        seg_rank.append(sr)
    return cuts, seg_rank


class ShardedKktPlan:
    """Local tables of every rank and the reduced system, from one classification of the whole NLP's unknowns."""

    def __init__(self, engine, ineq_rows, fixed_v, row_scale, shard_plan, group=None, positions="positions", only=None,
                 ends="chain", whole_entries=False):
        """``only``: the ranks whose local tables are built (default all; a process needs its own -- ``ranks[r]`` is None
        for the others; the checks that every entry is assembled exactly once still cover every rank).

        ``ends``: where a rank keeps the nodes it shares with its neighbours.  "chain" (default): as the first / last node
        of its chain segment, *not eliminated* (``KktTables.chain_export``): the cyclic reduction runs between them and
        their assembled panels are the rank's term of the reduced system -- the rank's border is the NLP's own.
        "border": in the rank's local border (every block of the rank then carries their columns: the first version,
        kept because it runs on the unchanged elimination kernels and in the NumPy oracle).

        ``whole_entries``: also build the entry tables of ``self.whole`` (the cut plan as ONE system: the tests' cross-check);
        by default only its block order and layout are computed -- the entry tables of the whole NLP are exactly what a
        rank of a sharded solve should not have to build."""
        if ends not in ("chain", "border"):
            raise ValueError("ends must be 'chain' or 'border'")
        self.ends = ends
        self.world = W = shard_plan.world
        ineq_rows = np.asarray(ineq_rows, dtype=np.int64)
        self.cuts, self.seg_rank = shard_cuts(engine, shard_plan)
        P: dict = {}
        # the cut plan as one system (a single rank can run it: the reference of the tests, and the border's order)
        self.whole = Tg = kkt.build_tables(engine, ineq_rows, fixed_v, row_scale, group, cuts=self.cuts, _parts=P,
                                           _layout_only=not whole_entries)
        cls, blk, nu, nv, n = P["cls"], P["blk"], P["nu"], P["nv"], P["n"]
        self.nu = nu
        base_border = int(Tg.leaf_ptr[-1]) + int(Tg.chain_ptr[-1])
        self.border = Bg = Tg.perm[base_border:]                 # border unknowns in the reduced system's order
        self.nb_red = len(Bg)
        red_pos = np.full(nu, -1, np.int64)
        red_pos[Bg] = np.arange(len(Bg))
        # ---- rank of every chain node, leaf and unknown ---------------------------------------------------------------
        seg_rank_flat = np.asarray([r for sr in self.seg_rank for r in sr], dtype=np.int64)
        assert len(seg_rank_flat) == Tg.n_phase
        chain_rank = np.repeat(seg_rank_flat, np.diff(Tg.chain_phase_ptr))
        leaf_rank = chain_rank[Tg.leaf_left] if Tg.n_leaf else np.zeros(0, np.int64)
        urank = np.full(nu, -1, np.int64)                         # leaf / chain unknowns: the rank that eliminates them
        urank[cls == LEAF] = leaf_rank[blk[cls == LEAF]]
        urank[cls == CHAIN] = chain_rank[blk[cls == CHAIN]]
        # border unknowns: owner (supplies diagonal and right-hand side) and the ranks that hold it in their local border
        owner = np.zeros(nu, np.int64)
        member = np.zeros((W, nu), bool)
        is_border = cls == BORDER
        member[:, is_border] = True
        remap = [[] for _ in range(W)]      # per rank: (unknowns of a cut node, chain node that stands for it in the rank's segment)
        for ip, (c, sr) in enumerate(zip(self.cuts, self.seg_rank)):
            mp = P["maps"][ip]
            cut_boundaries = np.nonzero(mp[4])[0]
            for j, node in enumerate(c):
                u = np.nonzero(is_border & (P["u_phase"] == ip) & (P["u_node"] == node))[0]
                owner[u] = sr[j + 1]
                member[:, u] = False
                member[sr[j], u] = True
                member[sr[j + 1], u] = True
                g_end = int(P["chain_base"][ip] + mp[5][cut_boundaries[j]])   # last node of the segment on its left ...
                remap[sr[j]].append((u, g_end))
                remap[sr[j + 1]].append((u, g_end + 1))                        # ... first node of the one on its right
        for r in range(W):
            member[r, urank == r] = True
        # ---- entries and the rank whose kernels write each entry's source ----------------------------------------------
        eu, ev, ekind, eidx, ecoef = kkt.natural_entries(n, nv, P["hr"], P["hc"], P["jr"], P["jc"], row_scale, ineq_rows)
        fixed = P["fixed"]
        keep = ~(fixed[eu] | fixed[ev])
        eu, ev, ekind, eidx, ecoef = eu[keep], ev[keep], ekind[keep], eidx[keep], ecoef[keep]
        oG, oH = shard_plan.num_c, shard_plan.num_c + shard_plan.nnz_G
        pos_owner = np.full(oH + shard_plan.nnz_H, -1, np.int64)      # -1: written by the tail kernel (every rank has it)
        for r in range(W):
            ix = shard_plan.index[r]
            pos_owner[ix[ix < len(pos_owner)]] = r
        self.pos_owner = pos_owner
        src_pos = np.where(ekind == SRC_H, oH + eidx, np.where(ekind == SRC_G, oG + eidx, eu - nv))
        eowner = pos_owner[src_pos]
        eowner = np.where(eowner < 0, 0, eowner)
        a_in, b_in = ~is_border[eu], ~is_border[ev]
        # the rank that assembles an entry: the one that eliminates an end of it, else the one whose tiles write its source
        # (one pass over the entries, not one per rank: 10 M entries at 60 k shuttle nodes)
        erank = eowner.copy()
        erank[b_in] = urank[ev[b_in]]
        erank[a_in] = urank[eu[a_in]]
        both = a_in & b_in
        if np.any(urank[eu[both]] != urank[ev[both]]):
            raise RuntimeError("a KKT entry is assembled by no rank or by two")
        if not (member[erank, eu].all() and member[erank, ev].all()):
            bad = int(erank[np.nonzero(~(member[erank, eu] & member[erank, ev]))[0][0]])
            raise RuntimeError(f"rank {bad}: a KKT entry couples its blocks with an unknown outside its local border")
        so = pos_owner[src_pos]                    # a rank must find every source it reads among its own tile outputs or the tail's
        if np.any((so >= 0) & (so != erank)):
            bad = int(erank[np.nonzero((so >= 0) & (so != erank))[0][0]])
            raise RuntimeError(f"rank {bad}: a KKT entry of its blocks is written by another rank's tiles")
        self.ranks: list[RankTables] = []
        for r in range(W):
            if only is not None and r not in only:
                self.ranks.append(None)
                continue
            sel = erank == r
            self.ranks.append(self._local(r, P, Tg, member[r], urank, owner, red_pos, chain_rank, leaf_rank,
                                          (eu[sel], ev[sel], ekind[sel], eidx[sel], ecoef[sel]), positions,
                                          remap[r] if ends == "chain" else []))
        # ---- the reduced system: all border unknowns, dense, its entries arrive as the ranks' Schur complements --------
        nr = self.nb_red
        z = np.zeros(nr, np.int64)
        e0 = (np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0, np.int32), np.zeros(0, np.int64), np.zeros(0))
        self.reduced = kkt._finish(positions, 0, 0, nr, 0, 0, np.full(nr, BORDER, np.int8), z.copy(), z.copy(), z.copy(),
                                   np.arange(nr, dtype=np.int64), np.zeros(nr, bool), np.zeros(nr, bool), 0, 0,
                                   np.zeros(1, np.int64), np.zeros(0, np.int64), None, None, None, None, None, None,
                                   entries=e0, n_primal=int(np.sum(Bg < nv)), n_dual=int(np.sum(Bg >= nv)))

    def _local(self, r, P, Tg, member, urank, owner, red_pos, chain_rank, leaf_rank, entries, positions, remap) -> RankTables:
        cls, blk = P["cls"], P["blk"]
        if remap:                                   # the rank's shared nodes: chain nodes of its own segments
            cls, blk = cls.copy(), blk.copy()
            for u, gid in remap:
                cls[u], blk[u] = CHAIN, gid
        univ = np.nonzero(member)[0].astype(np.int64)
        g2l = np.full(len(member), -1, np.int64)
        g2l[univ] = np.arange(len(univ))
        # own chain segments and leaves, renumbered consecutively in the whole plan's order
        my_chain = np.nonzero(chain_rank == r)[0]
        chain_l = np.full(Tg.n_chain, -1, np.int64)
        chain_l[my_chain] = np.arange(len(my_chain))
        my_leaf = np.nonzero(leaf_rank == r)[0]
        leaf_l = np.full(Tg.n_leaf, -1, np.int64)
        leaf_l[my_leaf] = np.arange(len(my_leaf))
        seg = [(int(a), int(b)) for a, b in zip(Tg.chain_phase_ptr[:-1], Tg.chain_phase_ptr[1:]) if chain_rank[a] == r]
        seg_ptr = np.concatenate([[0], np.cumsum([b - a for a, b in seg])]).astype(np.int64)
        leaf_left = chain_l[Tg.leaf_left[my_leaf]] if len(my_leaf) else np.zeros(0, np.int64)
        c_l = cls[univ]
        b_l = np.zeros(len(univ), np.int64)
        b_l[c_l == LEAF] = leaf_l[blk[univ][c_l == LEAF]]
        b_l[c_l == CHAIN] = chain_l[blk[univ][c_l == CHAIN]]
        assert np.all(b_l >= 0) and np.all(leaf_left >= 0)
        shared = P["cls"][univ] == BORDER           # (by the whole plan's classes: the NLP's border and the cut nodes)
        own = (urank[univ] == r) | (shared & (owner[univ] == r))
        chain_export = None
        if remap:
            chain_export = np.zeros(len(my_chain), np.uint8)
            chain_export[chain_l[[gid for _, gid in remap]]] = 1
        eu, ev, ekind, eidx, ecoef = entries
        nul = len(univ)
        T = kkt._finish(positions, 0, 0, nul, 0, 0, c_l.astype(np.int8), b_l, P["key_node"][univ], P["key_kind"][univ],
                        np.arange(nul, dtype=np.int64), P["dual"][univ], P["fixed"][univ] & own, len(my_leaf), len(my_chain),
                        seg_ptr, leaf_left, None, None, None, None, None, None,
                        entries=(g2l[eu], g2l[ev], ekind, eidx, ecoef),
                        n_primal=int(np.sum(own & ~P["dual"][univ])), n_dual=int(np.sum(own & P["dual"][univ])),
                        chain_export=chain_export)
        base_chain = int(T.leaf_ptr[-1])
        base_border = base_chain + int(T.chain_ptr[-1])
        border_red = red_pos[univ[T.perm[base_border:]]]
        assert np.all(border_red >= 0) and np.all(np.diff(border_red) > 0)   # a sub-sequence of the reduced order

        def node_red(c):
            return red_pos[univ[T.perm[base_chain + int(T.chain_ptr[c]):base_chain + int(T.chain_ptr[c + 1])]]]
        export_red, export_anchor_red = [], []
        for c, nz, nr in kkt.export_shapes(T):
            seg = int(np.searchsorted(T.chain_phase_ptr, c, side="right") - 1)
            export_red.append(node_red(c))
            export_anchor_red.append(node_red(int(T.chain_phase_ptr[seg + 1] - 1)) if nr else np.zeros(0, np.int64))
            assert np.all(export_red[-1] >= 0) and len(export_red[-1]) == nz and len(export_anchor_red[-1]) == nr
        return RankTables(T, univ, own, border_red, export_red, export_anchor_red)

    # ---- what a rank contributes and takes, as index operations on whole-NLP vectors (shared by every driver) --------
    def local_vector(self, r, v):
        R = self.ranks[r]
        return np.where(R.own, np.asarray(v, float)[R.univ], 0.0)

    def add_border(self, r, B_local, B_red, panels=None):
        """Add a rank's terms into the reduced matrix (kept whole, both triangles): its Schur complement on its local
        border (lower triangle valid) and the assembled panels [D | K | F] of its exported chain nodes."""
        R = self.ranks[r]
        ix = R.border_red
        Bl = np.tril(np.asarray(B_local).reshape(len(ix), len(ix)))
        B_red[np.ix_(ix, ix)] += Bl + np.tril(Bl, -1).T
        for P_, ru, ra in zip(panels or [], R.export_red or [], R.export_anchor_red or []):
            nz, nr = len(ru), len(ra)
            B_red[np.ix_(ru, ru)] += P_[:, :nz]
            if nr:
                B_red[np.ix_(ru, ra)] += P_[:, nz:nz + nr]
                B_red[np.ix_(ra, ru)] += P_[:, nz:nz + nr].T
            B_red[np.ix_(ru, ix)] += P_[:, nz + nr:]
            B_red[np.ix_(ix, ru)] += P_[:, nz + nr:].T

    def footprint(self, r) -> dict:
        """Doubles of matrix storage a rank's factorisation holds, against the unsharded plan's."""
        return {"local_vals": int(self.ranks[r].tables.total_vals), "reduced_vals": int(self.reduced.total_vals),
                "nb_local": int(self.ranks[r].tables.nb), "nb_reduced": int(self.nb_red)}


def factor_ranks(plan: ShardedKktPlan, handles, reduced, dvec, use_hess=True, reduce=None):
    """One factorisation over the rank handles this process holds (``handles``: {rank: object with factor_partial}); the
    others' contributions arrive through ``reduce`` (sum over processes of an array; None: all ranks are here)."""
    B = np.zeros((plan.nb_red, plan.nb_red))
    cnt = np.zeros(2, np.int64)
    for r, h in handles.items():
        Bl, p, q = h.factor_partial(plan.local_vector(r, dvec), use_hess)
        plan.add_border(r, Bl, B, h.export_panels() if plan.ranks[r].export_red else None)
        cnt += (p, q)
    if reduce is not None:
        B = reduce(B)
        cnt = reduce(cnt.astype(np.float64)).astype(np.int64)
    p, q = reduced.border_load_factor(B)
    return int(cnt[0] + p), int(cnt[1] + q)


def solve_ranks(plan: ShardedKktPlan, handles, reduced, rhs, reduce=None):
    """K^-1 rhs with the standing factors; returns the solution entries of the ranks held here (others zero) -- summed
    over processes by ``reduce`` into the whole vector."""
    rb = np.zeros(plan.nb_red)
    for r, h in handles.items():
        R = plan.ranks[r]
        rb[R.border_red] += h.forward_partial(plan.local_vector(r, rhs))
        if R.export_red:
            np.add.at(rb, np.concatenate(R.export_red), h.export_rhs())
    if reduce is not None:
        rb = reduce(rb)
    xb = reduced.solve(rb)
    x = np.zeros(plan.nu)
    for r, h in handles.items():
        R = plan.ranks[r]
        if R.export_red:
            h.import_solution(xb[np.concatenate(R.export_red)])
        xl = h.backward_partial(xb[R.border_red])
        x[R.univ[R.own]] = xl[R.own]
    if reduce is not None:
        x = reduce(x)
    return x


class ShardedKkt:
    """One process's part of the sharded factorisation on its GPU: rank ``rank`` of ``plan`` (or several ranks one after
    the other, ``ranks=[...]``: how a one-GPU box emulates a node), G~ / H~ read from the given device pointers.

    ``group``: a ``torch.distributed`` group for the two small reductions (None with all ranks held here)."""

    def __init__(self, engine, plan: ShardedKktPlan, ranks, d_jac=None, d_hess=None, group=None, distributed=False):
        self.plan = plan
        self.handles = {}
        for i, r in enumerate(ranks):
            dj = d_jac[i] if isinstance(d_jac, (list, tuple)) else d_jac
            dh = d_hess[i] if isinstance(d_hess, (list, tuple)) else d_hess
            self.handles[r] = kkt.GpuKkt(engine, None, None, None, tables=plan.ranks[r].tables, d_jac=dj, d_hess=dh)
        self.reduced = kkt.GpuKkt(engine, None, None, None, tables=plan.reduced)
        self.group = group
        self.distributed = distributed

    def _reduce(self):
        if not self.distributed:
            return None
        import torch
        import torch.distributed as dist

        def red(a):
            t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).copy())
            if dist.get_backend(self.group) == "nccl":
                t = t.cuda()
            dist.all_reduce(t, group=self.group)
            return t.cpu().numpy().reshape(np.shape(a))
        return red

    def factor(self, dvec, use_hess=True):
        return factor_ranks(self.plan, self.handles, self.reduced, dvec, use_hess, self._reduce())

    def solve(self, rhs):
        return solve_ranks(self.plan, self.handles, self.reduced, rhs, self._reduce())

    def close(self):
        for h in list(self.handles.values()) + [self.reduced]:
            h.close()
