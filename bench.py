"""Headline benchmark: NLP-callback evaluations per second (c + jac_g + hess at one (x, sigma, lambda)).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one fused ``pc_eval_all_device`` over the whole transcribed NLP with x~ and lambda already
resident in HBM and c~, G~, H~ left in HBM.  N = 1: BASELINE.json configs[1] -- hypersensitive problem,
1 phase, 2000 mesh sections x 6 Lobatto nodes = 10 001 collocation nodes.

N > 1 (one rank per GPU, RCCL): the NLP's mesh is sharded by contiguous section ranges; every step contains the
path's exchange -- one all-gather of the ranks' CSR runs and per-tile partial sums, after which every rank holds the
complete c~, G~, H~ (``--gather-root``: gather to rank 0 only) -- pycollo_amd/sharding.py.
``value`` is always the evaluations per second of ONE NLP, never multiplied by the rank count:
* ``--scaling strong`` (default, what BASELINE.json's metric says: the 10 k-node NLP on 1/2/4/8 GPUs): the SAME mesh
  (``--sections`` in total) on every N.  At 10 k nodes an evaluation is 5 us and the exchange tens of us, so this
  series goes DOWN with N; it goes up where the north_star puts sharding, e.g. ``--problem shuttle --sections 20000
  --order 4`` (config 4) or ``--problem delta_iii --sections 3125 --order 5`` (config 5).
* ``--scaling weak``: ``--sections`` per GPU, the mesh grows with N; ``value`` is still evaluations of that (N times
  larger) NLP per second, and ``shard_evals_per_s`` = N x value is the per-shard rate.

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` and `cpu_baseline`.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def cpu_baseline(K_sections: int, order: int, budget_s: float = 21.0):
    """Time the oracle's C restatement on the GPU box's host cores: a thread sweep 1, 2, 4, ... 64 (bounded sample:
    ~3 s per thread count), with the CPU model and the container's CPU quota beside it."""
    try:
        from oracle import cport
        return cport.time_hypersensitive(K_sections, order, budget_s)
    except Exception as exc:  # the baseline is reported, never required for the GPU number
        return {"value": None, "unit": "evals/s", "cores": 0, "kind": "port", "sample": f"unavailable: {exc}"}


def host_leg(eng, x_np, lam_np, n_calls: int):
    """SURVEY 8d leg (ii): the fused callback from HOST pointers -- x~, lambda in host memory, c~, G~, H~ delivered to
    host memory, the call returning only when they are there -- timed call by call (median and p95 over >= 1000 calls
    after 50 warm-up calls).  `inplace`: the caller reads / writes the library's pinned staging blocks
    (pc_host_buffers), so no host memcpy on either side; `copying`: fresh pageable numpy arrays in and out, as the
    reference's cyipopt object hands them over (pycollo/nlp.py:47-63).  Every data-movement mode is measured."""
    import numpy as np
    n_calls = max(1000, int(n_calls))
    hx, hl, _, _, _ = eng.host_buffers()

    def stats(fn):
        for _ in range(50):
            fn()
        ts = np.empty(n_calls)
        for i in range(n_calls):
            t0 = time.perf_counter()
            fn()
            ts[i] = time.perf_counter() - t0
        return float(np.median(ts) * 1e6), float(np.percentile(ts, 95) * 1e6)

    names = {0: "dma_up_dma_down", 1: "kernel_reads_host_dma_down", 2: "dma_up_kernel_writes_host",
             3: "kernel_reads_and_writes_host"}
    by_mode = {}
    for mode in (0, 1, 2, 3):
        eng.set_host_mode(mode)
        hx[:] = x_np
        hl[:] = lam_np
        med, p95 = stats(lambda: eng.evaluate_all_inplace(1.0))
        cmed, cp95 = stats(lambda: eng.evaluate_all(x_np, 1.0, lam_np))
        by_mode[names[mode]] = {"inplace_median_us": round(med, 2), "inplace_p95_us": round(p95, 2),
                                "copying_median_us": round(cmed, 2), "copying_p95_us": round(cp95, 2)}
    eng.set_host_mode(0)
    best = min(by_mode, key=lambda k: by_mode[k]["inplace_median_us"])
    b = by_mode[best]
    return {"call": "pc_eval_all (host pointers in, host pointers out, synchronous)", "calls": n_calls, "mode": best,
            "median_us": b["inplace_median_us"], "p95_us": b["inplace_p95_us"],
            "evals_per_s": round(1e6 / b["inplace_median_us"], 1),
            "copying_median_us": b["copying_median_us"], "copying_p95_us": b["copying_p95_us"],
            "by_mode": by_mode}


def host_by_config(local_rank, no_cpu):
    """what a host-resident (serial) solver sees beyond config 2 -- the fused callback from host
    pointers (pc_eval_all: x~, lambda in host memory, c~, G~, H~ delivered to host memory, synchronous; best data-movement
    mode, in-place pinned blocks) at BASELINE.json configs[2] and configs[3], with the CPU port on one thread and on the
    container's CPU quota beside it.  At these sizes the call is the bus: 6.8 MB / 82 MB of results per evaluation."""
    import numpy as np
    from pycollo_amd import problems
    from pycollo_amd.engine import NlpEngine
    out = []
    for label, name, K, order in (("config 3: cart-pole, 5000 sections x 4 nodes", "cart_pole", 5000, 4),
                                  ("config 4: shuttle, 20000 sections x 4 nodes", "shuttle", 20000, 4)):
        eng = NlpEngine(problems.REGISTRY[name](K=K, order=order), device=local_rank)
        x = np.random.default_rng(1234).uniform(-0.45, 0.45, eng.num_x)
        lam = np.random.default_rng(1235).normal(size=eng.num_c)
        hx, hl, _, _, _ = eng.host_buffers()
        best = None
        for mode in (0, 3):      # DMA both ways / the kernels read and write the pinned host blocks
            eng.set_host_mode(mode)
            hx[:] = x
            hl[:] = lam
            for _ in range(5):
                eng.evaluate_all_inplace(1.0)
            n = 200 if name == "cart_pole" else 40
            ts = np.empty(n)
            for i in range(n):
                t0 = time.perf_counter()
                eng.evaluate_all_inplace(1.0)
                ts[i] = time.perf_counter() - t0
            med = float(np.median(ts) * 1e6)
            if best is None or med < best[0]:
                best = (med, float(np.percentile(ts, 95) * 1e6), mode)
        eng.set_host_mode(0)
        entry = {"workload": label, "nodes": int(eng.layout.phases[0].N), "result_bytes": int(8 * (eng.num_c + eng.nnz_jac + eng.nnz_hess)),
                 "host_pointer_median_us": round(best[0], 1), "host_pointer_p95_us": round(best[1], 1), "host_mode": best[2],
                 "host_pointer_evals_per_s": round(1e6 / best[0], 1), "bus_GBs": round(8 * (eng.num_c + eng.nnz_jac + eng.nnz_hess) / best[0] / 1e3, 1)}
        eng.close()
        if not no_cpu:
            try:
                from oracle import cport
                cb = cport.time_problem(name, K, order, 6.0)
                entry["cpu_port_evals_per_s"] = cb["by_threads"]
                one = cb["by_threads"]["1"]
                bestc = max(cb["by_threads"].values())
                entry["host_pointer_vs_1_thread"] = round(entry["host_pointer_evals_per_s"] / one, 1)
                entry["host_pointer_vs_best_threads"] = round(entry["host_pointer_evals_per_s"] / bestc, 1)
            except Exception as exc:   # noqa: BLE001 -- the baseline is reported, never required
                entry["cpu_port_evals_per_s"] = f"unavailable: {exc}"[:200]
        out.append(entry)
    return out


def sharded_config_lines(world, rank, local_rank, dev, tstream, args):
    """BASELINE.json configs[3] and configs[4] -- the configurations the north_star shards -- on the same N ranks, beside
    the headline line: whole-NLP evaluations per second (strong scaling: the mesh is fixed), the time of the exchange
    alone, the time of this launch's tile kernels on the slowest rank and their HBM fraction.  Every rank runs this (it
    contains collectives); rank 0 reports.  Serial and overlapped exchange, padded all-gather and all-gatherv."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from pycollo_amd import problems
    from pycollo_amd.sharding import ShardedNlp
    lines = []
    for label, name, kw in (("config 4: shuttle, 20000 sections x 4 nodes", "shuttle", dict(K=20000, order=4)),
                            ("config 5: Delta III, 4 phases x 3125 sections x 5 nodes", "delta_iii", dict(K=3125, order=5))):
        sh = ShardedNlp(problems.REGISTRY[name](**kw), device=local_rank)
        dist.barrier()   # every rank has its engine (a rank that had to compile the code object is waited for here)
        lo, hi = (0.05, 0.3) if name == "delta_iii" else (-0.45, 0.45)
        x = torch.from_numpy(np.random.default_rng(1234).uniform(lo, hi, sh.num_x)).to(dev)
        lam = torch.from_numpy(np.random.default_rng(1235).normal(size=sh.num_c)).to(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

        def timed(fn, n):
            for _ in range(5):
                fn()
            torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
            with torch.cuda.stream(tstream):
                e0.record(tstream)
                for _ in range(n):
                    fn()
                e1.record(tstream)
            torch.cuda.synchronize()
            t = torch.tensor([e0.elapsed_time(e1) / n], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item()) * 1e3   # us, slowest rank

        def run(**kk):
            with torch.cuda.stream(tstream):
                sh.evaluate_all_device(x, 1.0, lam, tstream.cuda_stream, **kk)

        def bulk():
            sh.engine.launch_bulk_only(x, lam, sh.c, sh.G, sh.H, tstream.cuda_stream)

        def exch():
            with torch.cuda.stream(tstream):
                sh.exchange.run(sh.buf)

        n = 30
        t_serial = timed(lambda: run(), n)
        t_overlap = timed(lambda: run(overlap=True), n)
        t_unpadded = timed(lambda: run(unpadded=True), n)
        t_root = timed(lambda: run(root=0), n)
        t_bulk = timed(bulk, n)
        t_exch = timed(exch, n)
        lines.append({"workload": label, "nodes": int(sum(pl.N for pl in sh.engine.layout.phases)), "n_gpus": world,
                      "evals_per_s": round(1e6 / min(t_serial, t_overlap), 1),
                      "us_per_eval": {"serial_exchange": round(t_serial, 1), "exchange_overlapping_hessian_tiles": round(t_overlap, 1),
                                      "all_gatherv_unpadded": round(t_unpadded, 1), "gather_to_rank0": round(t_root, 1)},
                      "exchange_us": round(t_exch, 1), "exchange_bytes_per_rank": int(8 * sh.plan.maxlen * world),
                      "rank_tile_kernels_us": round(t_bulk, 2),
                      "rank_hbm_fraction": round(sh.local_algorithmic_bytes / (t_bulk * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                      "rank_algorithmic_bytes": int(sh.local_algorithmic_bytes)})
        if name == "shuttle":
            try:
                lines[-1]["sharded_kkt"] = sharded_kkt_line(sh, x, lam, dev, tstream, rank, world)
            except Exception as exc:   # noqa: BLE001 -- reported, never required (the ranks are symmetric)
                lines[-1]["sharded_kkt"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
        sh.engine.close()
    return lines


def sharded_kkt_line(sh, x, lam, dev, tstream, rank, world):
    """The consumer that needs only a rank's own rows, on the same ranks: the evaluation with only the
    per-tile partial sums exchanged, and one interior-point KKT system factorised and solved with the chain cut across the
    ranks (pycollo_amd/kkt_sharded.py) -- wall time of the slowest rank, reductions included."""
    import time
    import numpy as np
    import torch
    import torch.distributed as dist
    from pycollo_amd.kkt_sharded import ShardedKkt, ShardedKktPlan
    eng = sh.engine
    n, m = eng.num_x, eng.num_c
    lay = eng.layout
    ineq = []
    for pl, pm in zip(lay.phases, eng.model.phases):
        ineq += list(range(pl.c_path_off, pl.c_path_off + pm.n_p * pl.N))
    ineq = np.array(sorted(ineq + list(range(lay.c_end_off, m, 2))), dtype=np.int64)
    ns = len(ineq)
    rng = np.random.default_rng(7)
    fixed = np.zeros(n + ns, bool)
    sc = np.ones(m)
    dvec = np.concatenate([rng.uniform(0.5, 2.0, n + ns) + 50.0, -1e-8 * np.ones(m)])
    rhs = rng.normal(size=n + ns + m)

    def slowest(fn, reps):
        fn()
        torch.cuda.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        t = torch.tensor([(time.perf_counter() - t0) / reps], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def local_eval():
        with torch.cuda.stream(tstream):
            sh.evaluate_local_device(x, 1.0, lam, tstream)
    t_eval = slowest(local_eval, 20)
    t0 = time.perf_counter()
    plan = ShardedKktPlan(eng, ineq, fixed, sc, sh.plan, only=[rank])
    t_plan = time.perf_counter() - t0
    sk = ShardedKkt(eng, plan, [rank], d_jac=sh.G.data_ptr(), d_hess=sh.H.data_ptr(), distributed=True)
    inertia = sk.factor(dvec)
    t_factor = slowest(lambda: sk.factor(dvec), 5)
    t_solve = slowest(lambda: sk.solve(rhs), 5)
    f = plan.footprint(rank)
    out = {"nodes_unknowns": int(plan.nu), "local_eval_us_partial_sums_only": round(t_eval * 1e6, 1),
           "factor_ms": round(t_factor * 1e3, 3), "solve_ms": round(t_solve * 1e3, 3),
           "inertia": list(inertia), "inertia_expected": [int(n + ns), int(m)],
           "reduced_border_unknowns": int(plan.nb_red), "rank0_local_border": f["nb_local"],
           "rank0_matrix_MB": round(8e-6 * f["local_vals"], 1), "plan_build_s": round(t_plan, 2),
           "note": "host vectors in and out of every call; reductions: nb_red^2 doubles per factorisation, nb_red + nu per solve"}
    sk.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--warmup", type=int, default=2000)
    ap.add_argument("--sections", type=int, default=2000, help="mesh sections (in total; per GPU with --scaling weak)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong", help="N > 1: same NLP on every N, or N times the mesh")
    ap.add_argument("--gather-root", action="store_true", help="N > 1: gather the shards to rank 0 only instead of all-gather")
    ap.add_argument("--order", type=int, default=6, help="nodes per section")
    ap.add_argument("--problem", default="hypersensitive")
    ap.add_argument("--tpb", type=int, default=0, help="threads per block (0 = auto)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--build-only", action="store_true",
                    help="compile the code object this command needs and exit without touching the GPU (run this, "
                         "unprofiled, before any rocprofv3 pass: no compiler may be spawned under the profiler)")
    ap.add_argument("--no-host", action="store_true", help="skip the host-pointer (pc_eval_all) timing leg")
    ap.add_argument("--no-host-configs", action="store_true", help="skip the host-pointer lines of configs 3 and 4 (host_by_config)")
    ap.add_argument("--host-calls", type=int, default=2000, help="timed host-pointer calls per variant (>= 1000)")
    ap.add_argument("--no-pin", action="store_true", help="leave the launching thread to the scheduler")
    ap.add_argument("--ragged", action="store_true", help="ph-refined style mesh: random section sizes, orders 4..8")
    ap.add_argument("--refined", type=int, default=0, metavar="NODES",
                    help="a mesh as ph refinement leaves it, about NODES nodes per phase (refinement.synthetic_refined_mesh, "
                         "seeds 7, 8, ...): orders in runs, the mixed build")
    ap.add_argument("--generic", action="store_true", help="use the any-mesh kernels (no order specialisation)")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL); 'gloo' only to rehearse N > 1 on one GPU")
    ap.add_argument("--check", action="store_true", help="N > 1: compare the sharded result with an unsharded one")
    ap.add_argument("--overlap", action="store_true", help="N > 1: exchange the c / G runs while the H tiles still run (two tile launches)")
    ap.add_argument("--unpadded", action="store_true", help="N > 1: all-gatherv (one broadcast per rank at its exact length) instead of the padded all-gather")
    ap.add_argument("--no-sharded-configs", action="store_true", help="N > 1: skip the config-4 / config-5 lines (BASELINE.json configs[3], configs[4])")
    args = ap.parse_args()

    import numpy as np
    if args.build_only:
        import __graft_entry__ as entry
        from pycollo_amd import codegen, problems as _pr
        from pycollo_amd.model import compile_model
        entry.build_library()
        pb = _pr.REGISTRY[args.problem](K=args.sections * max(1, args.gpus), order=args.order)
        orders = tuple(0 for _ in pb.phases) if (args.ragged or args.generic or args.refined) else tuple(args.order for _ in pb.phases)
        mixed = None
        if args.refined and not args.generic:
            from pycollo_amd.engine import choose_spec_orders
            _pr.with_refined_mesh(pb, args.refined)
            mixed = tuple(choose_spec_orders(ph.mesh.resolved()[1]) for ph in pb.phases)
        print(codegen.build_code_object(compile_model(pb), orders, mixed=mixed))
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # one core for the launching thread and the HIP runtime's helpers, before anything initialises HIP
    pinned_cpu, full_mask = (-1, set())
    if not args.no_pin:
        from pycollo_amd.hostpin import pin_launch_thread
        pinned_cpu, full_mask = pin_launch_thread(local_rank, world)
        pinned_mask = os.sched_getaffinity(0)
        if os.environ.get("PYCOLLO_AMD_SPIN_PROBE"):
            from pycollo_amd.hostpin import spin_seconds
            print(f"spin probe: cpu {pinned_cpu} {spin_seconds() * 1e3:.3f} ms", file=sys.stderr, flush=True)

    import torch
    import torch.distributed as dist

    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: pycollo_amd has no CPU fallback")
    if args.backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()   # rehearsal: ranks may share a device
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    from pycollo_amd import problems
    from pycollo_amd.engine import NlpEngine

    K_total = args.sections * (world if args.scaling == "weak" else 1)
    prob = problems.REGISTRY[args.problem](K=K_total, order=args.order)
    if args.ragged:
        rr = np.random.default_rng(7)
        for ph in prob.phases:
            ph.mesh.mesh_section_sizes = rr.uniform(0.5, 1.5, K_total)
            ph.mesh.number_mesh_section_nodes = rr.integers(4, 9, K_total)
    if args.refined:
        problems.with_refined_mesh(prob, args.refined)
    # a dedicated (non-default) stream: the library treats a NULL stream as "use the handle's own stream",
    # and torch events only see the stream they are recorded on
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    assert stream != 0

    if world == 1:
        eng = NlpEngine(prob, device=local_rank, threads_per_block=args.tpb, specialise=not args.generic)
        rng = np.random.default_rng(1234)
        x = torch.from_numpy(rng.uniform(-0.45, 0.45, eng.num_x)).to(dev)
        lam = torch.from_numpy(np.random.default_rng(1235).normal(size=eng.num_c)).to(dev)
        c = torch.empty(eng.num_c, dtype=torch.float64, device=dev)
        G = torch.empty(eng.nnz_jac, dtype=torch.float64, device=dev)
        H = torch.empty(eng.nnz_hess, dtype=torch.float64, device=dev)

        step = eng.bind_device(x, lam, c, G, H, stream)   # evaluate_all_device with the addresses resolved once

        def bulk_only():
            eng.launch_bulk_only(x, lam, c, G, H, stream)

        info = eng.info
        alg_bytes = info["algorithmic_bytes"]
        nph = len(prob.phases)
        workload = (f"{args.problem}, {nph} phase{'s' if nph > 1 else ''}, {K_total} mesh sections x {args.order} Lobatto nodes "
                    f"= {eng.layout.phases[0].N} collocation nodes{' per phase' if nph > 1 else ''}")
        if args.refined or args.ragged:
            kind = "ph-refined (synthetic error field)" if args.refined else "random orders 4..8 per section"
            workload = (f"{args.problem}, {nph} phase{'s' if nph > 1 else ''}, {kind} mesh, "
                        f"{sum(m.K for m in eng.meshes)} sections, {sum(pl.N for pl in eng.layout.phases)} collocation nodes")
        extra = {"num_x": eng.num_x, "num_c": eng.num_c, "nnz_jac": eng.nnz_jac, "nnz_hess": eng.nnz_hess,
                 "tiles": info["n_tiles_total"], "threads_per_block": info["threads_per_block"], "waves_per_tile": info["waves_per_tile"],
                 "launches_per_eval": info["n_launches"], "lds_bytes_per_workgroup": info["lds_bytes_max"],
                 "launch_thread_cpu": pinned_cpu}
        if any(eng.mixed):   # mixed build: which orders have a tile body, and how much of the mesh runs them
            od = [eng.phase_tile_orders(p) for p in range(nph)]
            extra["mixed_build"] = {"specialised_orders": [list(m) for m in eng.mixed],
                                    "order_pure_tiles": int(sum(int((o > 0).sum()) for o in od)),
                                    "any_order_tiles": int(sum(int((o == 0).sum()) for o in od))}
        if os.environ.get("PYCOLLO_AMD_BENCH_ADDR"):   # diagnostic: where the buffers landed
            extra["addr"] = {k: hex(t.data_ptr()) for k, t in (("x", x), ("lam", lam), ("c", c), ("G", G), ("H", H))}
    else:
        from pycollo_amd.sharding import ShardedNlp
        sh = ShardedNlp(prob, device=local_rank, threads_per_block=args.tpb)
        rng = np.random.default_rng(1234)
        x = torch.from_numpy(rng.uniform(-0.45, 0.45, sh.num_x)).to(dev)
        lam = torch.from_numpy(np.random.default_rng(1235).normal(size=sh.num_c)).to(dev)

        root = 0 if args.gather_root else None

        def step():
            sh.evaluate_all_device(x, 1.0, lam, stream, root, overlap=args.overlap, unpadded=args.unpadded)

        def bulk_only():   # this rank's tiles only
            sh.engine.launch_bulk_only(x, lam, sh.c, sh.G, sh.H, stream)

        if args.check:   # the complete, reassembled outputs of every rank equal the unsharded evaluation bit for bit
            ref = NlpEngine(prob, device=local_rank, threads_per_block=sh.engine.info["threads_per_block"])
            rc = torch.empty(ref.num_c, dtype=torch.float64, device=dev)
            rG = torch.empty(ref.nnz_jac, dtype=torch.float64, device=dev)
            rH = torch.empty(ref.nnz_hess, dtype=torch.float64, device=dev)
            ref.evaluate_all_device(x, 1.0, lam, rc, rG, rH, stream)
            c_, G_, H_ = sh.evaluate_all_device(x, 1.0, lam, stream, root, overlap=args.overlap, unpadded=args.unpadded)
            torch.cuda.synchronize()
            def same(a, b):   # bit for bit, NaN == NaN (a random point may leave a model's domain)
                return bool(torch.equal(torch.nan_to_num(a, nan=1.25e300), torch.nan_to_num(b, nan=1.25e300)))
            holds_all = root is None or rank == root     # --gather-root: only the solver's rank ends with every shard
            ok = (same(rc, c_) and same(rG, G_) and same(rH, H_)) if holds_all else True
            if not holds_all:
                print(f"[rank {rank}] not the root of the gather: holds its own shard only", file=sys.stderr, flush=True)
            if not ok:
                for nm, a, b in (("c", rc, c_), ("G", rG, G_), ("H", rH, H_)):
                    bad = torch.nonzero(torch.nan_to_num(a, nan=1.25e300) != torch.nan_to_num(b, nan=1.25e300)).flatten()
                    print(f"[rank {rank}] {nm}: {bad.numel()} differing entries, first {bad[:5].tolist()}, "
                          f"NaNs {int(torch.isnan(a).sum())}/{int(torch.isnan(b).sum())}", file=sys.stderr, flush=True)
            if holds_all:
                print(f"[rank {rank}] sharded == unsharded: {ok}", file=sys.stderr, flush=True)
            if not ok:
                raise SystemExit("sharded evaluation differs from the unsharded one")
            ref.close()
        alg_bytes = sh.local_algorithmic_bytes
        nph = len(prob.phases)
        workload = (f"{args.problem}, {nph} phase{'s' if nph > 1 else ''}, {K_total} mesh sections x {args.order} Lobatto nodes "
                    f"= {sh.engine.layout.phases[0].N} collocation nodes{' per phase' if nph > 1 else ''}, sharded by section over "
                    f"{world} GPUs ({args.scaling} scaling)")
        extra = {"num_x": sh.num_x, "num_c": sh.num_c, "nnz_jac": sh.nnz_jac, "nnz_hess": sh.nnz_hess,
                 "exchange": ("one gather to rank 0 per evaluation" if args.gather_root else "one all_gather_into_tensor per evaluation")
                             + " (CSR runs of c, G, H + per-tile partial sums)",
                 "exchange_padding_fraction": round(sh.plan.padding_fraction, 4),
                 "exchange_overlapped_with_hessian_tiles": bool(args.overlap), "exchange_unpadded": bool(args.unpadded),
                 "backend": dist.get_backend(), "world_size": dist.get_world_size()}
        assert dist.get_world_size() == world == args.gpus

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    colocated = []
    if not args.no_pin:   # the runtime's completion thread next to the launching thread (pycollo_amd/hostpin.py)
        from pycollo_amd.hostpin import colocate_runtime_threads

        def burst():
            for _ in range(max(200, min(3000, args.warmup))):
                step()
            torch.cuda.synchronize()
        burst()
        extra["pin_survived_hip_init"] = os.sched_getaffinity(0) == pinned_mask
        if world > 1 and not extra["pin_survived_hip_init"]:
            os.sched_setaffinity(0, pinned_mask)   # this rank's slice again (HIP's start-up reset it)
        if world == 1 and full_mask:   # one rank: also choose the launching core by measurement
            from pycollo_amd.hostpin import tune_launch_core
            best, timings = tune_launch_core(burst, full_mask)
            if best >= 0:
                extra["launch_thread_cpu"] = best
                extra["launch_core_probe"] = [[c, round(t * 1e3, 3)] for c, t in timings]
            colocated = [0]   # (moved inside tune_launch_core)
        else:
            colocated = colocate_runtime_threads(burst)
    extra["runtime_threads_colocated"] = len(colocated)

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    evals_per_s = args.steps / dt   # evaluations of ONE NLP per second, whatever the number of ranks

    # ---- dominant kernel (bulk): K back-to-back launches between two HIP events on the launch stream
    roofline = None
    dev_step_ms = None
    if bulk_only is not None:
        for _ in range(20):
            bulk_only()
        # The launches are queued behind a blocker (an fp64 GEMM of ~15 ms on the same stream) so that the GPU
        # finds them back to back: at 5 us a kernel the host's launch rate (3.4-4.4 us a launch, bimodal between
        # runs) would otherwise leak into the figure.  e0/e1 are recorded on that stream, after the blocker.
        # its own launch counts, whatever --steps says: 5 batches of 2000 launches each, the median batch is reported
        n_roof, n_batches = 2000, 5
        blk = torch.empty((8192, 8192), dtype=torch.float64, device=dev).normal_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

        def timed_batch(fn):
            torch.cuda.synchronize()
            torch.mm(blk, blk)
            e0.record()
            for _ in range(n_roof):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n_roof

        k_batches = sorted(timed_batch(bulk_only) for _ in range(n_batches))
        k_ms = k_batches[n_batches // 2]
        # the same for whole evaluations (bulk + tail): device time per evaluation with the host out of the picture
        d_batches = sorted(timed_batch(step) for _ in range(n_batches))
        dev_step_ms = d_batches[n_batches // 2]
        del blk
        # One launch per evaluation (resident tail): the dominant kernel IS the evaluation, pc_bulk_p0_r / pc_bulk_all_r,
        # and its launch time is the device time per step; the tile-only kernel is reported beside it.
        one_launch = world == 1 and extra.get("launches_per_eval") == 1
        kname = ("pc_bulk_all" if len(prob.phases) > 1 else "pc_bulk_p0") + ("_r" if one_launch else "")
        if one_launch:   # the per-replica variant pc_create picks when tiles are shared (two-wave build of heavy models)
            from pycollo_amd import codegen as _cg
            wpt = int(extra.get("waves_per_tile", 1))
            if wpt > 1 and all(wpt in _cg._static_w_list(pm, single_phase=len(eng.model.phases) == 1) for pm in eng.model.phases):
                kname += f"_w{wpt}"
        # HBM bytes per launch of that kernel from the PMC passes committed under profiles/ (rocprofv3 cannot run inside
        # this process); only quoted when the workload is the one those passes profiled
        traffic = None
        traffic_source = None
        try:
            import glob
            latest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic.json")))[-1]
            with open(latest) as f:
                pmc = json.load(f)
            for entry in pmc.values():
                if world == 1 and isinstance(entry, dict) and entry.get("workload") == workload:
                    traffic = entry["kernels"].get(kname, {}).get("hbm_bytes_per_launch")
                    if traffic:
                        traffic_source = (f"profiles/{os.path.basename(latest)} (separate rocprofv3 --pmc passes over this "
                                          f"workload; not measured by this run)")
        except (OSError, IndexError):
            pass
        dom_ms = dev_step_ms if one_launch else k_ms
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_source,
                    "algorithmic_bytes_per_launch": int(alg_bytes), "avg_launch_us": round(dom_ms * 1e3, 3),
                    "tiles_only_kernel_us": round(k_ms * 1e3, 3),
                    "launch_us_batches": [round(b * 1e3, 3) for b in (d_batches if one_launch else k_batches)],
                    "method": f"median of {n_batches} batches of {n_roof} launches, each batch queued behind a blocker, "
                              f"between two HIP events on the launch stream"}

    # ---- what a host-side caller (IPOPT) sees: pc_eval_all with host pointers in and out, one call at a time ----
    host = None
    if world == 1 and not args.no_host:
        # (eight timed loops of `calls` evaluations each: bounded to ~32 GB over the bus in total for large meshes; config 2 keeps its 2000)
        per_call = 8.0 * (eng.num_x + 2 * eng.num_c + eng.nnz_jac + eng.nnz_hess)
        calls = int(max(20, min(args.host_calls, 4e9 / per_call)))
        host = host_leg(eng, x.cpu().numpy(), lam.cpu().numpy(), calls)

    # (after every headline figure has been measured; a failure here -- the ranks are symmetric, so it is a failure on
    #  every rank -- is reported in the line instead of costing it)
    sharded_configs = None
    if world > 1 and not args.no_sharded_configs:
        try:
            sharded_configs = sharded_config_lines(world, rank, local_rank, dev, tstream, args)
        except Exception as exc:   # noqa: BLE001
            sharded_configs = [{"error": f"{type(exc).__name__}: {exc}"[:300]}]
    if rank == 0:
        out = {"metric": "NLP-callback evals/sec (g + jac_g + hess)", "value": round(evals_per_s, 2), "unit": "evals/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 6),
               "device_ms_per_step": round(dev_step_ms, 6) if roofline is not None else None,
               "higher_is_better": True, "scaling": args.scaling if world > 1 else "strong", "vs_baseline": None, "dtype": "f64",
               "data": "synthetic",
               "config": {"workload": workload, **extra}}
        if world > 1 and args.scaling == "weak":
            out["shard_evals_per_s"] = round(evals_per_s * world, 2)
        if sharded_configs is not None:
            out["sharded_configs"] = sharded_configs
        if roofline is not None:
            out["roofline"] = roofline
        if host is not None:
            out["host_ms_per_step"] = round(host["median_us"] * 1e-3, 6)
            out["host"] = host
        default_workload = args.problem == "hypersensitive" and args.sections == 2000 and args.order == 6 and not (args.ragged or args.refined or args.generic)
        if world == 1 and not args.no_host and not args.no_host_configs and default_workload:
            if full_mask:
                from pycollo_amd.hostpin import restore_affinity
                restore_affinity(full_mask)
            out["host_by_config"] = host_by_config(local_rank, args.no_cpu)
        if world == 1 and not args.no_cpu:
            if full_mask:
                from pycollo_amd.hostpin import restore_affinity
                restore_affinity(full_mask)   # the CPU leg may use several cores
            cb = cpu_baseline(args.sections, args.order)
            out["cpu_baseline"] = cb
            # the north_star's ">= 50x" read against this run's own CPU figures, for the device-resident rate (`value`)
            # and for the host-pointer call a serial solver makes (`host_ms_per_step`)
            if cb.get("value"):
                ratios = {"device_resident_vs_1_thread": round(evals_per_s / cb["value"], 1),
                          "device_resident_vs_best_threads": round(evals_per_s / cb.get("best_value", cb["value"]), 1)}
                if host is not None:
                    ratios["host_pointer_vs_1_thread"] = round(host["evals_per_s"] / cb["value"], 1)
                    ratios["host_pointer_vs_best_threads"] = round(host["evals_per_s"] / cb.get("best_value", cb["value"]), 1)
                out["speedup_vs_cpu_port"] = ratios
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
