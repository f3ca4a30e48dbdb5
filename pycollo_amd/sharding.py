"""Multi-GPU evaluation: shard every phase's mesh by contiguous section ranges, one rank per GPU.

SURVEY.md section 8e.  A defect row of section k reads only that section's nodes, path rows and the
Hessian node blocks are per node, so a rank that owns the tile range [tb, te) of a phase produces a
handful of *contiguous CSR runs* of c~, G~ and H~ (one per state / path constraint / variable block /
strip) plus its tiles' partial sums.  One evaluation is

    bulk kernels over the local tiles  ->  pack the local runs  ->  ONE all-gather (RCCL over xGMI)
    ->  unpack the other ranks' runs   ->  tail kernel on every rank over the complete partial sums

so every rank ends with the complete, bit-identical c~, G~, H~ (the integral rows, the (q,t,s) Jacobian
columns, the Hessian sums and the endpoint block are finished redundantly by each rank's tail kernel
from the gathered per-tile partials -- no separate all-reduce, and the summation order does not depend on
the number of ranks).  x~ and lambda are replicated (a few MB at most).

``ShardPlan`` is pure NumPy (testable on a CPU); ``SegmentExchange`` moves tensors with torch ops only
and therefore runs unchanged over gloo on the CPU (tests) and over RCCL on GPUs.
"""
from __future__ import annotations

import numpy as np


def _indptr(rows: np.ndarray, n_rows: int) -> np.ndarray:
    out = np.zeros(n_rows + 1, dtype=np.int64)
    np.add.at(out, rows.astype(np.int64) + 1, 1)
    return np.cumsum(out)


def tile_cuts(sec_s: np.ndarray, tile_k0, world: int) -> list:
    """Tile index at which every rank's contiguous share of one phase starts (``world + 1`` entries): shares balanced
    by the nodes they hold (a rank's part of c~, G~, H~ is proportional to its nodes; on a ph-refined mesh tiles hold
    different numbers of nodes, and a padded exchange is as long as the longest share)."""
    k0s = np.asarray(tile_k0, dtype=np.int64)
    n_tiles = len(k0s) - 1
    tile_nodes = np.diff(sec_s[k0s])
    cum = np.concatenate([[0], np.cumsum(tile_nodes)])
    cuts = [int(np.searchsorted(cum, cum[-1] * r / world, side="left")) for r in range(world + 1)]
    cuts[0], cuts[-1] = 0, n_tiles
    for r in range(1, world + 1):
        cuts[r] = max(cuts[r], cuts[r - 1])
    return cuts


def _range_segments(lay, pm, pl, mesh, gp, hp, h_cols, oG, oH, part_off, nred, ka, kb, tb, te, owns_last=None):
    """(start, stop) positions in the combined buffer [c | G | H | partials ...] that the tiles [tb, te) = sections
    [ka, kb) of one phase produce: the tile kernels' outputs for the nodes they own -- defect rows of every state with
    their Jacobian blocks, path rows, the z entries of the integral rows, the Hessian rows / strips of the owned nodes
    -- and the tiles' partial sums.  Shared by the global plan (ShardPlan) and a rank-local handle (LocalShard), whose
    segment lists therefore correspond one to one."""
    seg = []
    N = pl.N
    jmask, hmask, tz = pm.jac_mask(), pm.hess_mask(), pm.t_strip_mask()
    wk = pm.w_kind or [0] * pm.n_s
    wi = pm.w_idx or list(range(pm.n_s))

    def h_slot(row, col):
        a, b = hp[row], hp[row + 1]
        j = a + np.searchsorted(h_cols[a:b], col)
        assert j < b and h_cols[j] == col
        return int(j)

    n0, n1 = int(mesh.s[ka]), int(mesh.s[kb])
    if owns_last is None:
        owns_last = kb == mesh.K
    n1o = n1 + (1 if owns_last else 0)             # owned nodes [n0, n1o): the phase's last node belongs to its last tile
    for a in range(pm.n_y):                         # defect rows n0 .. n1-1 of every state
        r0 = pl.c_off + a * (N - 1)
        seg.append((r0 + n0, r0 + n1))
        seg.append((oG + gp[r0 + n0], oG + gp[r0 + n1]))
    for m in range(pm.n_p):                         # path rows of the owned nodes
        r0 = pl.c_path_off + m * N
        seg.append((r0 + n0, r0 + n1o))
        seg.append((oG + gp[r0 + n0], oG + gp[r0 + n1o]))
    for m in range(pm.n_q):                         # z entries of the integral rows
        row = pl.c_int_off + m
        rank_b = 0
        for b in range(pm.n_z):
            if jmask[pm.n_y + pm.n_p + m, b]:
                base = oG + gp[row] + rank_b * N
                seg.append((base + n0, base + n1o))
                rank_b += 1
    for b in range(pm.n_z):                         # Hessian rows of the owned nodes
        if hmask[b, :pm.n_z].any():
            r0 = pl.x_off + b * N
            seg.append((oH + hp[r0 + n0], oH + hp[r0 + n1o]))
    for jt in range(pm.n_t):                        # t strips
        for b in range(pm.n_z):
            if tz[b]:
                base = oH + h_slot(pl.t_off + jt, pl.x_off + b * N)
                seg.append((base + n0, base + n1o))
    for l in range(pm.n_s):                         # strips of the parameters (s, q; a time's is its t strip)
        if wk[l] == 2:
            continue
        prow = lay.s_off + wi[l] if wk[l] == 0 else pl.q_off + wi[l]
        for b in range(pm.n_z):
            if hmask[pm.n_z + l, b]:
                base = oH + h_slot(prow, pl.x_off + b * N)
                seg.append((base + n0, base + n1o))
    if nred:
        seg.append((part_off + tb * nred, part_off + te * nred))
    return seg


class ShardPlan:
    """Which positions of the combined buffer [c | G | H | partials_p0 | partials_p1 | ...] each rank produces."""

    def __init__(self, engine, world: int):
        self.world = world
        lay, model = engine.layout, engine.model
        self.num_c, self.nnz_G, self.nnz_H = engine.num_c, engine.nnz_jac, engine.nnz_hess
        g_rows, g_cols = engine.evaluate_G_structure()
        h_rows, h_cols = engine.evaluate_H_structure()
        gp = _indptr(g_rows, engine.num_c)
        hp = _indptr(h_rows, engine.num_x)
        oG, oH = self.num_c, self.num_c + self.nnz_G
        self.part_off = []
        off = oH + self.nnz_H
        self.tiles = []
        for ip in range(len(model.phases)):
            k0, nred = engine.phase_tiles(ip)
            self.tiles.append((k0, nred))
            self.part_off.append(off)
            off += (len(k0) - 1) * nred
        self.total = off
        self.tile_ranges = [[None] * len(model.phases) for _ in range(world)]
        self.segments = [[] for _ in range(world)]   # (start, stop) in the combined buffer

        for ip, (pm, pl, mesh) in enumerate(zip(model.phases, lay.phases, engine.meshes)):
            k0s, nred = self.tiles[ip]
            n_tiles = len(k0s) - 1
            cuts = tile_cuts(np.asarray(mesh.s, dtype=np.int64), k0s, world)
            for r in range(world):
                tb, te = cuts[r], cuts[r + 1]
                self.tile_ranges[r][ip] = (tb, te)
                if te <= tb:
                    continue
                self.segments[r] += _range_segments(lay, pm, pl, mesh, gp, hp, h_cols, oG, oH, self.part_off[ip], nred,
                                                     int(k0s[tb]), int(k0s[te]), tb, te)
        self.segments = [[(int(a), int(b)) for a, b in s if b > a] for s in self.segments]
        self.index = [np.concatenate([np.arange(a, b, dtype=np.int64) for a, b in s]) if s else np.zeros(0, np.int64)
                      for s in self.segments]
        self.lengths = [len(i) for i in self.index]
        self.maxlen = max(self.lengths) if self.lengths else 0
        # share of the gathered buffer that is padding (all_gather_into_tensor moves equal-sized pieces)
        self.padding_fraction = 1.0 - sum(self.lengths) / max(1, world * self.maxlen)

    def split(self):
        """The plan as two exchanges: (c~ and G~ runs) and (H~ runs + per-tile partial sums).  The first can travel
        while the tiles of H~ are still being computed (``ShardedNlp.evaluate_all_device(overlap=True)``)."""
        cut = self.num_c + self.nnz_G
        return (_SubPlan(self.world, [[(a, b) for a, b in seg if a < cut] for seg in self.segments]),
                _SubPlan(self.world, [[(a, b) for a, b in seg if a >= cut] for seg in self.segments]))


class _SubPlan:
    """A subset of a plan's segments with the attributes SegmentExchange reads."""

    def __init__(self, world, segments):
        self.world, self.segments = world, segments
        self.index = [np.concatenate([np.arange(a, b, dtype=np.int64) for a, b in s]) if s else np.zeros(0, np.int64)
                      for s in segments]
        self.lengths = [len(i) for i in self.index]
        self.maxlen = max(self.lengths) if self.lengths else 0
        self.padding_fraction = 1.0 - sum(self.lengths) / max(1, world * self.maxlen)


def _chunk_table(runs, chunk: int) -> np.ndarray:
    """[(src offset, dst offset, length)] of contiguous runs, cut into pieces of at most ``chunk`` elements."""
    out = []
    for so, do, n in runs:
        for o in range(0, n, chunk):
            out.append((so + o, do + o, min(chunk, n - o)))
    return np.asarray(out, dtype=np.int64).reshape(-1, 3)


class SegmentExchange:
    """pack -> all_gather_into_tensor -> unpack on one combined buffer.

    On GPU tensors over RCCL the pack and the unpack are one launch each of the library's run-copy kernel
    (``pc_copy_runs``): a rank's share is a handful of contiguous runs, so no index arrays are read.  On CPU
    tensors (gloo, the tests) and in the several-ranks-on-one-GPU rehearsal the same moves are torch index ops."""

    def __init__(self, plan: ShardPlan, rank: int, device, group=None):
        import torch
        self.torch = torch
        self.plan, self.rank, self.group = plan, rank, group
        self.world = plan.world
        self.maxlen = ml = max(plan.maxlen, 1)
        self.idx_me = torch.from_numpy(plan.index[rank]).to(device)
        self.recv = torch.zeros(self.world * ml, dtype=torch.float64, device=device)
        self.send = torch.zeros(ml, dtype=torch.float64, device=device)
        src, dst = [], []
        for r in range(self.world):
            if r == rank:
                continue
            src.append(np.arange(plan.lengths[r], dtype=np.int64) + r * ml)
            dst.append(plan.index[r])
        self.unpack_src = torch.from_numpy(np.concatenate(src) if src else np.zeros(0, np.int64)).to(device)
        self.unpack_dst = torch.from_numpy(np.concatenate(dst) if dst else np.zeros(0, np.int64)).to(device)
        self.lib = None
        if torch.device(device).type == "cuda":
            from .engine import load_library
            self.lib = load_library()
            chunk = int(self.lib.pc_run_chunk())
            runs, o = [], 0
            for a, b in plan.segments[rank]:
                runs.append((a, o, b - a))
                o += b - a
            self.pack_tab = torch.from_numpy(_chunk_table(runs, chunk)).to(device)
            runs = []
            for r in range(self.world):
                if r == rank:
                    continue
                o = r * ml
                for a, b in plan.segments[r]:
                    runs.append((o, a, b - a))
                    o += b - a
            self.unpack_tab = torch.from_numpy(_chunk_table(runs, chunk)).to(device)

    def _copy_runs(self, src, dst, tab):
        if tab.shape[0] == 0:
            return
        stream = self.torch.cuda.current_stream().cuda_stream
        if not self.lib.pc_copy_runs(src.data_ptr(), dst.data_ptr(), tab.data_ptr(), tab.shape[0], stream):
            raise RuntimeError("pc_copy_runs failed: " + self.lib.pc_last_error().decode())

    def run(self, buf, root=None, unpadded=False):
        """Exchange in place.  ``root=None``: all-gather, every rank ends with the complete buffer.  ``root=r``: gather
        to rank r only (the rank an NLP solver lives on): the other ranks send their share and keep their own.
        ``unpadded``: the all-gather in its list form, every rank's piece at its exact length (an all-gatherv: no padding
        to the longest share; RCCL runs it as a group of broadcasts -- an A/B switch, the padded single-buffer form is
        the default)."""
        import torch.distributed as dist
        torch = self.torch
        n = self.idx_me.numel()
        if self.world > 1 and max(self.plan.lengths) == 0:
            return buf
        if root is not None:
            if buf.is_cuda:
                self._copy_runs(buf, self.send, self.pack_tab)
            elif n:
                torch.index_select(buf, 0, self.idx_me, out=self.send[:n])
            ml = self.maxlen
            if buf.is_cuda and dist.get_backend(self.group) == "gloo":   # rehearsal (ranks sharing one GPU): through the host
                pieces = [torch.empty(ml, dtype=self.recv.dtype) for _ in range(self.world)] if self.rank == root else None
                dist.gather(self.send.cpu(), pieces, dst=root, group=self.group)
                if self.rank == root:
                    self.recv.copy_(torch.cat(pieces))
            else:
                pieces = [self.recv[q * ml:(q + 1) * ml] for q in range(self.world)] if self.rank == root else None
                dist.gather(self.send, pieces, dst=root, group=self.group)
            if self.rank == root:
                if buf.is_cuda:
                    self._copy_runs(self.recv, buf, self.unpack_tab)
                elif self.unpack_dst.numel():
                    buf.index_copy_(0, self.unpack_dst, self.recv.index_select(0, self.unpack_src))
            return buf
        rehearsal = buf.is_cuda and dist.get_backend(self.group) == "gloo"
        if buf.is_cuda:
            self._copy_runs(buf, self.send, self.pack_tab)
        elif n:
            torch.index_select(buf, 0, self.idx_me, out=self.send[:n])
        if rehearsal and unpadded:
            # rehearsal of the all-gatherv: the same per-rank broadcasts at exact lengths, staged through the host
            ml = self.maxlen
            recv = torch.zeros(self.recv.shape, dtype=self.recv.dtype)
            send = self.send.cpu()
            for q in range(self.world):
                nq = self.plan.lengths[q]
                if nq:
                    piece = recv[q * ml:q * ml + nq]
                    if q == self.rank:
                        piece.copy_(send[:nq])
                    dist.broadcast(piece, src=dist.get_global_rank(self.group, q) if self.group is not None else q, group=self.group)
            self.recv.copy_(recv)
        elif rehearsal:
            # rehearsal only (several ranks sharing one GPU cannot form an RCCL group): stage through the host
            recv = torch.empty(self.recv.shape, dtype=self.recv.dtype)
            dist.all_gather_into_tensor(recv, self.send.cpu(), group=self.group)
            self.recv.copy_(recv)
        elif unpadded:
            ml = self.maxlen
            # all-gatherv as one broadcast per rank at its exact length (what the list form of all_gather is on RCCL;
            # gloo's insists on equal sizes)
            for q in range(self.world):
                nq = self.plan.lengths[q]
                if nq:
                    piece = self.recv[q * ml:q * ml + nq]
                    if q == self.rank:
                        piece.copy_(self.send[:nq])
                    dist.broadcast(piece, src=dist.get_global_rank(self.group, q) if self.group is not None else q, group=self.group)
        else:
            dist.all_gather_into_tensor(self.recv, self.send, group=self.group)
        if buf.is_cuda:
            self._copy_runs(self.recv, buf, self.unpack_tab)
        elif self.unpack_dst.numel():
            buf.index_copy_(0, self.unpack_dst, self.recv.index_select(0, self.unpack_src))
        return buf


class ShardedNlp:
    """One rank of a section-sharded NLP evaluation (GPU).  Every rank holds the complete outputs."""

    def __init__(self, problem, device: int = 0, threads_per_block: int = 0, group=None, engine=None):
        """``engine``: an existing GPU engine to shard instead of building one from ``problem`` (a mesh iteration's, with
        its scaling set: ``ipm_sharded.solve_sharded``); it is restricted to this rank's tiles from here on."""
        import torch
        import torch.distributed as dist
        from .engine import NlpEngine
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.engine = eng = engine if engine is not None else NlpEngine(problem, device=device, threads_per_block=threads_per_block)
        self.plan = plan = ShardPlan(eng, self.world)
        dev = torch.device("cuda", device)
        self.buf = torch.zeros(plan.total, dtype=torch.float64, device=dev)
        oG, oH = plan.num_c, plan.num_c + plan.nnz_G
        self.c = self.buf[:oG]
        self.G = self.buf[oG:oH]
        self.H = self.buf[oH:oH + plan.nnz_H]
        for ip, ((k0, nred), off) in enumerate(zip(plan.tiles, plan.part_off)):
            if nred:
                eng.set_partials_buffer(ip, self.buf[off:off + (len(k0) - 1) * nred])
            tb, te = plan.tile_ranges[self.rank][ip]
            eng.set_tile_range(ip, tb, te)
        # (a group of one rank has nothing to exchange; ``always_exchange`` sends its outputs through the collectives
        #  all the same -- how the RCCL path is rehearsed on a one-GPU box, tests/test_gpu_sharded_process.py)
        self.always_exchange = False
        self.exchange = SegmentExchange(plan, self.rank, dev, group)
        cg, hp = plan.split()
        self.exchange_cg = SegmentExchange(cg, self.rank, dev, group)
        self.exchange_h = SegmentExchange(hp, self.rank, dev, group)
        # the per-tile partial sums alone (a few doubles per tile): all a rank needs of the others when its consumer reads
        # only its own rows (evaluate_local_device; kkt_sharded.py)
        p0 = oH + plan.nnz_H
        self.exchange_partials = SegmentExchange(_SubPlan(self.world, [[(a, b) for a, b in seg if a >= p0] for seg in plan.segments]),
                                                 self.rank, dev, group)
        self.cstream = torch.cuda.Stream(device=dev)       # the exchange's own stream when it overlaps the H~ tiles
        self._ev = [torch.cuda.Event() for _ in range(3)]
        # kernels, torch pack/unpack ops and the collective must share ONE non-default stream (the C ABI maps a
        # NULL stream to the handle's private stream)
        self.tstream = torch.cuda.Stream(device=dev)
        self.num_x, self.num_c, self.nnz_jac, self.nnz_hess = eng.num_x, eng.num_c, eng.nnz_jac, eng.nnz_hess
        # algorithmic bytes this rank's kernels move per evaluation: its share of the outputs + the inputs it reads
        self.local_algorithmic_bytes = 8 * (plan.lengths[self.rank] + (eng.num_x + eng.num_c) // self.world)

    def evaluate_all_device(self, d_x, obj_factor, d_lam, stream=None, root=None, overlap=False, unpadded=False):
        """Asynchronous on the torch stream that is current when called (must not be the default stream);
        falls back to this object's own stream.  ``root``: gather to that rank only (it alone finishes the evaluation
        and holds the complete c~, G~, H~); default: every rank does.

        ``overlap``: the tiles are launched twice -- c~ / G~ first, H~ second -- and the exchange of the c~ / G~ runs
        (four fifths of the bytes) travels on a second stream while the H~ tiles run; the H~ runs and the partial sums
        follow, then the tail.  Same bits as the serial form (the tile kernels compute an output the same way whatever
        else the launch computes); the price is a second evaluation of the node functions."""
        import torch
        cur = torch.cuda.current_stream()
        ts = cur if cur.cuda_stream != 0 else self.tstream
        eng = self.engine
        exchanging = self.world > 1 or self.always_exchange
        if overlap and exchanging:
            e1, e2, e3 = self._ev
            with torch.cuda.stream(ts):
                eng.launch_bulk_flags(d_x, d_lam, self.c, self.G, self.H, 1 | 2, ts.cuda_stream)
                e1.record(ts)
                eng.launch_bulk_flags(d_x, d_lam, self.c, self.G, self.H, 4, ts.cuda_stream)
                e2.record(ts)
            with torch.cuda.stream(self.cstream):
                self.cstream.wait_event(e1)
                self.exchange_cg.run(self.buf, root, unpadded)
                self.cstream.wait_event(e2)
                self.exchange_h.run(self.buf, root, unpadded)
                e3.record(self.cstream)
            with torch.cuda.stream(ts):
                ts.wait_event(e3)
                if root is None or self.rank == root:
                    eng.launch_tail_only(d_x, obj_factor, d_lam, self.c, self.G, self.H, ts.cuda_stream)
            return self.c, self.G, self.H
        with torch.cuda.stream(ts):
            eng.launch_bulk_only(d_x, d_lam, self.c, self.G, self.H, ts.cuda_stream)
            if exchanging:
                self.exchange.run(self.buf, root, unpadded)
            if root is None or self.rank == root:
                eng.launch_tail_only(d_x, obj_factor, d_lam, self.c, self.G, self.H, ts.cuda_stream)
        return self.c, self.G, self.H


    def release(self):
        """Give the engine back whole: every phase's full tile range, its own partial-sum buffers (an engine adopted with
        ``engine=`` goes on to serve its mesh iteration: mesh-error estimate, scaling)."""
        import torch
        torch.cuda.synchronize()
        for ip, (k0, nred) in enumerate(self.plan.tiles):
            if nred:
                self.engine.set_partials_buffer(ip, None)
            self.engine.set_tile_range(ip, 0, len(k0) - 1)

    def evaluate_local_device(self, d_x, obj_factor, d_lam, stream=None):
        """The evaluation for a consumer that reads only this rank's rows (the KKT factorisation cut across ranks,
        ``kkt_sharded.ShardedKkt``): the rank's tiles, an all-gather of the **per-tile partial sums only**, the tail on
        every rank.  c~, G~, H~ then hold this rank's runs and the tail's outputs; what other ranks' tiles write is not
        exchanged and stays whatever the buffer held.  Same stream rules as ``evaluate_all_device``."""
        import torch
        cur = torch.cuda.current_stream()
        ts = cur if cur.cuda_stream != 0 else self.tstream
        with torch.cuda.stream(ts):
            self.engine.launch_bulk_only(d_x, d_lam, self.c, self.G, self.H, ts.cuda_stream)
            if self.world > 1 or self.always_exchange:
                self.exchange_partials.run(self.buf)
            self.engine.launch_tail_only(d_x, obj_factor, d_lam, self.c, self.G, self.H, ts.cuda_stream)
        return self.c, self.G, self.H


# ---------------------------------------------------------------------------------------------------------------------
# Rank-local handles: a rank holds its section range only (pattern, tables, buffers), the root holds the whole NLP
# ---------------------------------------------------------------------------------------------------------------------
def global_tile_plan(model, meshes, device: int = -1, threads_per_block: int = 0, quad=None, mixed="auto", orders=None):
    """The tiling the unsharded handle would cut -- per phase (tile_k0, tile_order), plus the build (orders, mixed) and the
    workgroup size -- from a plan-only handle (``pc_problem_desc.plan_only``: O(sections) memory, no patterns)."""
    from .engine import NlpEngine
    plan = NlpEngine(model, meshes, device=None, threads_per_block=threads_per_block, quad=quad, plan_only=True,
                     plan_device=device, mixed=mixed, orders=orders)
    tiles = [(plan.phase_tiles(ip)[0], plan.phase_tile_orders(ip)) for ip in range(len(model.phases))]
    out = {"tiles": tiles, "orders": plan.orders, "mixed": plan.mixed, "threads_per_block": plan.info["threads_per_block"]}
    plan.close()
    return out


class LocalShard:
    """One rank of the section-sharded evaluation with RANK-LOCAL memory ("mesh-refinement
    iterations that blow past one GPU's collocation-node count").

    The rank builds an engine for its own part of every phase's mesh only -- the sections of its tile range between two
    one-section halos (the section before the range: the range's first node closes it, so that node's adjoint weight
    and quadrature weight need it; the section after it: the range's end node opens it and belongs to the next rank, so
    the rank's last tile must not be the mesh's last) -- with the global tiling's tiles of that range as a fixed tile
    table between two halo tiles.  Every owned tile therefore computes bit for bit what it computes in the whole mesh, on a handle whose
    pattern, tables and buffers are the rank's share.  Inputs are the rank's slices of x~ and lambda (``x_index`` /
    ``lam_index`` into the global vectors); outputs are the rank's segments of the global combined buffer
    [c | G | H | partials], packed in the order of the global plan's segment list (``pack``), which the root scatters
    into place (``LocalRoot``) before it runs the tail kernel of the whole NLP.  The only exchange is the gather to the
    root.  The reference has no counterpart (single process)."""

    def __init__(self, problem, rank: int, world: int, device: int | None = 0, meshes=None, quad=None, plan=None,
                 threads_per_block: int = 0):
        from .engine import NlpEngine
        from .layout import NlpLayout
        from .mesh import build_phase_mesh
        from .model import Model, compile_model
        from .quadrature import QuadratureTables
        self.rank, self.world = int(rank), int(world)
        self.model = model = problem if isinstance(problem, Model) else compile_model(problem)
        self.quad = quad or QuadratureTables(model.quadrature_method)
        if meshes is None:
            meshes = [build_phase_mesh(self.quad, *ph.mesh.resolved()) for ph in problem.phases]
        self.global_meshes = meshes
        plan = plan or global_tile_plan(model, meshes, -1 if device is None else int(device), threads_per_block, self.quad)
        self.plan = plan
        glay = NlpLayout(model, meshes)           # offsets only: O(phases)
        self.global_num_x, self.global_num_c = glay.num_x, glay.num_c
        local_meshes, fixed, mixed, self.ranges, self.halo, self.halo_after, self.first_section = [], [], [], [], [], [], []
        for ip, (pm, mesh) in enumerate(zip(model.phases, meshes)):
            k0s, orders = plan["tiles"][ip]
            cuts = tile_cuts(np.asarray(mesh.s, dtype=np.int64), k0s, world)
            tb, te = cuts[rank], cuts[rank + 1]
            empty = te <= tb
            if empty:                              # nothing of this phase: one section, nothing owned
                ka, kb, halo, after = 0, 1, 0, 0
            else:
                ka, kb = int(k0s[tb]), int(k0s[te])
                halo = 1 if ka > 0 else 0
                after = 1 if kb < mesh.K else 0
            k_first, k_last = ka - halo, kb + after
            sizes = np.asarray(mesh.sizes, dtype=np.float64)[k_first:k_last]     # fractions of the WHOLE period: h_k = 2 sizes_k
            nodes = np.asarray(mesh.n, dtype=np.int64)[k_first:k_last]
            lm = build_phase_mesh(self.quad, sizes, nodes)
            # the section widths the kernels read are the WHOLE mesh's, bit for bit (build_phase_mesh pins its last node
            # to tau = +1 and accumulates edges from -1: neither holds for a part of the mesh)
            lm.h = np.asarray(mesh.h, dtype=np.float64)[k_first:k_last].copy()
            local_meshes.append(lm)
            spec = tuple(plan["mixed"][ip]) if plan["mixed"][ip] else ((int(plan["orders"][ip]),) if plan["orders"][ip] else ())
            mixed.append(spec)
            if empty:
                tk, to = np.array([0, 1], dtype=np.int32), np.array([0], dtype=np.int32)
            else:
                body = np.asarray(k0s[tb:te + 1], dtype=np.int64) - k_first
                tk = np.concatenate([[0], body]) if halo else body
                own = np.asarray(orders[tb:te], dtype=np.int32) if plan["mixed"][ip] else np.full(te - tb, plan["orders"][ip], dtype=np.int32)
                nh = int(mesh.n[k_first])
                to = np.concatenate([[nh if nh in spec else 0], own]) if halo else own
                if after:
                    na = int(mesh.n[kb])
                    tk = np.concatenate([tk, [tk[-1] + 1]])
                    to = np.concatenate([to, [na if na in spec else 0]])
            fixed.append((tk.astype(np.int32), to.astype(np.int32)))
            self.ranges.append((tb, te))
            self.halo.append(halo)
            self.halo_after.append(after)
            self.first_section.append(k_first)
        self.engine = eng = NlpEngine(model, local_meshes, device=device, quad=self.quad, orders=tuple(0 for _ in local_meshes),
                                      mixed=tuple(mixed), fixed_tiles=fixed, threads_per_block=plan["threads_per_block"])
        llay = eng.layout
        # ---- which global x~ / lambda entries the local vectors are (the halo and the shared end node included)
        xi = np.zeros(eng.num_x, dtype=np.int64)
        li = np.zeros(eng.num_c, dtype=np.int64)
        for pm, gl, ll, mesh, k_first in zip(model.phases, glay.phases, llay.phases, meshes, self.first_section):
            ns = int(mesh.s[k_first])
            loc = np.arange(ll.N, dtype=np.int64)
            for b in range(pm.n_z):
                xi[ll.x_off + b * ll.N + loc] = gl.x_off + b * gl.N + ns + loc
            for m in range(pm.n_q + pm.n_t):
                xi[ll.q_off + m] = gl.q_off + m
            for a in range(pm.n_y):
                li[ll.c_off + a * (ll.N - 1) + loc[:-1]] = gl.c_off + a * (gl.N - 1) + ns + loc[:-1]
            for m in range(pm.n_p):
                li[ll.c_path_off + m * ll.N + loc] = gl.c_path_off + m * gl.N + ns + loc
            for m in range(pm.n_q):
                li[ll.c_int_off + m] = gl.c_int_off + m
        xi[llay.s_off:llay.s_off + llay.n_s] = glay.s_off + np.arange(llay.n_s)
        li[llay.c_end_off:llay.c_end_off + llay.n_b] = glay.c_end_off + np.arange(llay.n_b)
        self.x_index, self.lam_index = xi, li
        # ---- the rank's segments in its own combined buffer [c | G | H | partials ...], in the global plan's order
        g_rows, _ = eng.evaluate_G_structure()
        h_rows, h_cols = eng.evaluate_H_structure()
        gp, hp = _indptr(g_rows, eng.num_c), _indptr(h_rows, eng.num_x)
        oG, oH = eng.num_c, eng.num_c + eng.nnz_jac
        self.part_off, off, self.segments = [], oH + eng.nnz_hess, []
        for ip, (pm, ll, lmesh) in enumerate(zip(model.phases, llay.phases, local_meshes)):
            k0, nred = eng.phase_tiles(ip)
            self.part_off.append(off)
            off += (len(k0) - 1) * nred
            tb, te = self.ranges[ip]
            if te <= tb:
                continue
            t0, t1 = self.halo[ip], len(k0) - 1 - self.halo_after[ip]      # owned local tiles [t0, t1)
            self.segments += _range_segments(llay, pm, ll, lmesh, gp, hp, h_cols, oG, oH, self.part_off[ip], nred,
                                             int(k0[t0]), int(k0[t1]), t0, t1, owns_last=not self.halo_after[ip])
        self.total = off
        self.segments = [(int(a), int(b)) for a, b in self.segments if b > a]
        self.length = sum(b - a for a, b in self.segments)
        self.buf = None
        if device is not None:
            import torch
            dev = torch.device("cuda", int(device))
            self.buf = torch.zeros(self.total, dtype=torch.float64, device=dev)
            self.c, self.G, self.H = self.buf[:oG], self.buf[oG:oH], self.buf[oH:oH + eng.nnz_hess]
            for ip, off_p in enumerate(self.part_off):
                k0, nred = eng.phase_tiles(ip)
                if nred:
                    eng.set_partials_buffer(ip, self.buf[off_p:off_p + (len(k0) - 1) * nred])
            runs, o = [], 0
            chunk = int(eng._lib.pc_run_chunk())
            for a, b in self.segments:
                runs.append((a, o, b - a))
                o += b - a
            self.pack_tab = torch.from_numpy(_chunk_table(runs, chunk)).to(dev)
            self.send = torch.zeros(max(self.length, 1), dtype=torch.float64, device=dev)
            self.x_index_t = torch.from_numpy(xi).to(dev)
            self.lam_index_t = torch.from_numpy(li).to(dev)

    def device_bytes(self) -> int:
        """Doubles-and-indices footprint of the rank's handle: inputs, the combined output buffer, the send buffer and
        the pattern's index arrays (what scales with the rank's share of the mesh)."""
        e = self.engine
        return 8 * (e.num_x + e.num_c + self.total + self.length) + 8 * (e.nnz_jac + e.nnz_hess)

    def set_scaling(self, V_ocp, r_ocp, W_ocp, w_J=1.0):
        self.engine.set_scaling(V_ocp, r_ocp, W_ocp, w_J)

    def evaluate_packed(self, d_x_global, d_lam_global, stream=None):
        """Tile kernels over the rank's tiles on its slices of the global x~ / lambda (device tensors); returns the send
        buffer: the rank's segments of the global combined buffer, packed in plan order.  Asynchronous on the current
        torch stream."""
        import torch
        st = torch.cuda.current_stream().cuda_stream if stream is None else stream
        x = d_x_global.index_select(0, self.x_index_t)
        lam = d_lam_global.index_select(0, self.lam_index_t)
        self.engine.launch_bulk_only(x, lam, self.c, self.G, self.H, st)
        if self.pack_tab.shape[0]:
            if not self.engine._lib.pc_copy_runs(self.buf.data_ptr(), self.send.data_ptr(), self.pack_tab.data_ptr(),
                                                 self.pack_tab.shape[0], st):
                raise RuntimeError("pc_copy_runs failed: " + self.engine._lib.pc_last_error().decode())
        self._keep = (x, lam)
        return self.send[:self.length]

    def close(self):
        self.engine.close()


class LocalRoot:
    """The root of the rank-local sharded evaluation: it holds the whole NLP (pattern, full buffers -- the solver lives
    here), receives every rank's packed segments and finishes the evaluation with the tail kernel."""

    def __init__(self, engine, world: int):
        self.engine, self.world = engine, int(world)
        self.plan = ShardPlan(engine, world)

    def unpack_index(self, rank: int) -> np.ndarray:
        """Positions of rank ``rank``'s packed values in the combined buffer [c | G | H | partials ...]."""
        return self.plan.index[rank]


class LocalShardedNlp:
    """The rank-local sharded evaluation as processes run it (one per GPU, ``torch.distributed``): every rank holds a
    :class:`LocalShard` of its section range; the root rank additionally holds the whole NLP (``NlpEngine`` -- the NLP
    solver lives there) and finishes every evaluation.  Per evaluation: the root scatters every rank's slices of x~ and
    lambda, the ranks run their tiles and pack, ONE gather brings the packed segments to the root, which scatters them
    into the whole NLP's buffer (``pc_copy_runs``) and runs the tail kernel.  No rank but the root ever holds more than
    its share.  Over gloo (CPU tests; several ranks sharing one GPU) the tensors are staged through the host."""

    def __init__(self, problem, device: int | None = 0, root: int = 0, group=None, threads_per_block: int = 0):
        import torch
        import torch.distributed as dist
        from .engine import NlpEngine
        from .mesh import build_phase_mesh
        from .model import compile_model
        from .quadrature import QuadratureTables
        self.torch, self.dist, self.group, self.root = torch, dist, group, int(root)
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        model = compile_model(problem)
        quad = QuadratureTables(model.quadrature_method)
        meshes = [build_phase_mesh(quad, *ph.mesh.resolved()) for ph in problem.phases]
        plan = global_tile_plan(model, meshes, -1 if device is None else int(device), threads_per_block, quad)
        self.shard = LocalShard(model, self.rank, self.world, device, meshes=meshes, quad=quad, plan=plan)
        self.is_root = self.rank == self.root
        self.on_device = device is not None
        self.dev = torch.device("cuda", int(device)) if self.on_device else torch.device("cpu")
        self._host_staged = (not self.on_device) or (dist.is_initialized() and dist.get_backend(group) == "gloo")
        # every rank's packed length and (on the root) its index maps
        mine = (self.shard.length, self.shard.x_index, self.shard.lam_index)
        if dist.is_initialized() and self.world > 1:
            gathered = [None] * self.world if self.is_root else None
            dist.gather_object(mine, gathered, dst=self._global(self.root), group=group)
            sizes = [None]
            if self.is_root:
                sizes = [([g[0] for g in gathered], max(len(g[1]) for g in gathered), max(len(g[2]) for g in gathered))]
            dist.broadcast_object_list(sizes, src=self._global(self.root), group=group)
            self.lengths, self.x_max, self.lam_max = sizes[0]
        else:
            gathered, self.lengths = [mine], [self.shard.length]
            self.x_max, self.lam_max = len(self.shard.x_index), len(self.shard.lam_index)
        self.maxlen = max(max(self.lengths), 1)
        self.engine = None
        if self.is_root:
            self.engine = eng = NlpEngine(model, meshes, device=device, quad=quad, orders=plan["orders"], mixed=plan["mixed"],
                                          threads_per_block=plan["threads_per_block"])
            self.root_plan = rp = ShardPlan(eng, self.world)
            if [int(l) for l in self.lengths] != [int(l) for l in rp.lengths]:
                raise RuntimeError("the ranks' packed lengths differ from the root's plan")
            self.x_idx = [torch.from_numpy(g[1]).to(self.dev) for g in gathered]
            self.lam_idx = [torch.from_numpy(g[2]).to(self.dev) for g in gathered]
            self.num_x, self.num_c, self.nnz_jac, self.nnz_hess = eng.num_x, eng.num_c, eng.nnz_jac, eng.nnz_hess
            self.buf = torch.zeros(rp.total, dtype=torch.float64, device=self.dev)
            oG, oH = rp.num_c, rp.num_c + rp.nnz_G
            self.c, self.G, self.H = self.buf[:oG], self.buf[oG:oH], self.buf[oH:oH + rp.nnz_H]
            self.recv = torch.zeros(self.world * self.maxlen, dtype=torch.float64, device=self.dev)
            self.unpack_index = [torch.from_numpy(rp.index[r]).to(self.dev) for r in range(self.world)]
            if self.on_device:
                for ip, ((k0, nred), off) in enumerate(zip(rp.tiles, rp.part_off)):
                    if nred:
                        eng.set_partials_buffer(ip, self.buf[off:off + (len(k0) - 1) * nred])
                chunk = int(eng._lib.pc_run_chunk())
                runs = []
                for r in range(self.world):
                    o = r * self.maxlen
                    for a, b in rp.segments[r]:
                        runs.append((o, a, b - a))
                        o += b - a
                self.unpack_tab = torch.from_numpy(_chunk_table(runs, chunk)).to(self.dev)
        self.send = torch.zeros(self.maxlen, dtype=torch.float64, device=self.dev)
        self.x_local = torch.zeros(len(self.shard.x_index), dtype=torch.float64, device=self.dev)
        self.lam_local = torch.zeros(len(self.shard.lam_index), dtype=torch.float64, device=self.dev)

    def _global(self, r):
        return self.dist.get_global_rank(self.group, r) if self.group is not None else r

    def set_scaling(self, V_ocp, r_ocp, W_ocp, w_J=1.0):
        self.shard.set_scaling(V_ocp, r_ocp, W_ocp, w_J)
        if self.is_root and self.on_device:
            self.engine.set_scaling(V_ocp, r_ocp, W_ocp, w_J)

    def distribute(self, d_x=None, d_lam=None):
        """The root's global x~ / lambda -> every rank's slices (``x_local`` / ``lam_local``)."""
        dist, torch = self.dist, self.torch
        if self.world == 1:
            self.x_local.copy_(d_x.index_select(0, self.x_idx[0]))
            self.lam_local.copy_(d_lam.index_select(0, self.lam_idx[0]))
            return
        # (a scatter moves equal-sized pieces: every rank's slice is padded to the longest)
        for mine, src, idx, width in ((self.x_local, d_x, "x_idx", self.x_max), (self.lam_local, d_lam, "lam_idx", self.lam_max)):
            pieces = None
            if self.is_root:
                pieces = []
                for i in getattr(self, idx):
                    p = torch.zeros(width, dtype=src.dtype, device=src.device)
                    p[:i.numel()] = src.index_select(0, i)
                    pieces.append(p.cpu() if self._host_staged else p)
            out = torch.empty(width, dtype=mine.dtype, device="cpu" if self._host_staged else mine.device)
            dist.scatter(out, pieces, src=self._global(self.root), group=self.group)
            mine.copy_(out[:mine.numel()])

    def collect(self, packed):
        """The ranks' packed segments -> the root's whole buffer (gather + scatter into place)."""
        dist, torch = self.dist, self.torch
        n = packed.numel()
        self.send[:n].copy_(packed)
        if self.world == 1:
            self.recv[:self.maxlen].copy_(self.send)
        elif self._host_staged:
            pieces = [torch.empty(self.maxlen, dtype=self.send.dtype) for _ in range(self.world)] if self.is_root else None
            dist.gather(self.send.cpu(), pieces, dst=self._global(self.root), group=self.group)
            if self.is_root:
                self.recv.copy_(torch.cat(pieces))
        else:
            pieces = [self.recv[q * self.maxlen:(q + 1) * self.maxlen] for q in range(self.world)] if self.is_root else None
            dist.gather(self.send, pieces, dst=self._global(self.root), group=self.group)
        if not self.is_root:
            return
        if self.on_device:
            if self.unpack_tab.shape[0]:
                st = torch.cuda.current_stream().cuda_stream
                if not self.engine._lib.pc_copy_runs(self.recv.data_ptr(), self.buf.data_ptr(), self.unpack_tab.data_ptr(),
                                                     self.unpack_tab.shape[0], st):
                    raise RuntimeError("pc_copy_runs failed: " + self.engine._lib.pc_last_error().decode())
        else:
            for r in range(self.world):
                self.buf.index_copy_(0, self.unpack_index[r], self.recv[r * self.maxlen:r * self.maxlen + self.lengths[r]])

    def evaluate_all_device(self, d_x, obj_factor, d_lam, stream=None):
        """One evaluation; ``d_x`` / ``d_lam`` are the global vectors on the root (ignored elsewhere).  Returns the
        root's (c~, G~, H~) views (None on the other ranks).  Runs on the current torch stream."""
        torch = self.torch
        self.distribute(d_x, d_lam)
        st = torch.cuda.current_stream().cuda_stream if stream is None else stream
        sh = self.shard
        sh.engine.launch_bulk_only(self.x_local, self.lam_local, sh.c, sh.G, sh.H, st)
        if sh.pack_tab.shape[0]:
            if not sh.engine._lib.pc_copy_runs(sh.buf.data_ptr(), sh.send.data_ptr(), sh.pack_tab.data_ptr(), sh.pack_tab.shape[0], st):
                raise RuntimeError("pc_copy_runs failed: " + sh.engine._lib.pc_last_error().decode())
        self.collect(sh.send[:sh.length])
        if not self.is_root:
            return None
        self.engine.launch_tail_only(d_x, obj_factor, d_lam, self.c, self.G, self.H, st)
        return self.c, self.G, self.H

    def close(self):
        self.shard.close()
        if self.engine is not None:
            self.engine.close()
