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

        def h_slot(row, col):
            a, b = hp[row], hp[row + 1]
            j = a + np.searchsorted(h_cols[a:b], col)
            assert j < b and h_cols[j] == col
            return int(j)

        for ip, (pm, pl, mesh) in enumerate(zip(model.phases, lay.phases, engine.meshes)):
            k0s, nred = self.tiles[ip]
            n_tiles = len(k0s) - 1
            N = pl.N
            jmask = pm.jac_mask()
            hmask = pm.hess_mask()
            tz = pm.t_strip_mask()
            wk = pm.w_kind or [0] * pm.n_s
            wi = pm.w_idx or list(range(pm.n_s))
            # Ranks get contiguous tile ranges balanced by the nodes they hold (a rank's share of c~, G~, H~ is
            # proportional to its nodes): on a ph-refined mesh tiles hold different numbers of nodes, and the exchange
            # is padded to the longest share.
            tile_nodes = np.diff(np.asarray(mesh.s, dtype=np.int64)[np.asarray(k0s, dtype=np.int64)])
            cum = np.concatenate([[0], np.cumsum(tile_nodes)])
            cuts = [int(np.searchsorted(cum, cum[-1] * r / world, side="left")) for r in range(world + 1)]
            cuts[0], cuts[-1] = 0, n_tiles
            for r in range(1, world + 1):
                cuts[r] = max(cuts[r], cuts[r - 1])
            for r in range(world):
                tb, te = cuts[r], cuts[r + 1]
                self.tile_ranges[r][ip] = (tb, te)
                if te <= tb:
                    continue
                seg = self.segments[r]
                ka, kb = int(k0s[tb]), int(k0s[te])
                n0, n1 = int(mesh.s[ka]), int(mesh.s[kb])
                n1o = n1 + (1 if kb == mesh.K else 0)          # owned nodes [n0, n1o)
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
                    seg.append((self.part_off[ip] + tb * nred, self.part_off[ip] + te * nred))
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

    def __init__(self, problem, device: int = 0, threads_per_block: int = 0, group=None):
        import torch
        import torch.distributed as dist
        from .engine import NlpEngine
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.engine = eng = NlpEngine(problem, device=device, threads_per_block=threads_per_block)
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
