"""The sharded evaluation as the driver launches it -- one process per rank under ``torch.distributed.run`` -- rehearsed
with two ranks on the one GPU of the test box over gloo (RCCL needs a GPU per rank): every exchange form of
``ShardedNlp.evaluate_all_device`` must reassemble, bit for bit, what the unsharded launch writes
(``bench.py --check``; SURVEY.md section 8e, reference partition pycollo/mesh.py:297-335)."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


CASES = [
    ("overlapped split exchange, config-2 model", ["--sections", "400", "--overlap"], {"exchange_overlapped_with_hessian_tiles": True}),
    ("all-gatherv on a ragged multi-phase mesh", ["--problem", "delta_iii", "--sections", "60", "--ragged", "--unpadded", "--overlap"],
     {"exchange_unpadded": True, "exchange_overlapped_with_hessian_tiles": True}),
    ("gather to the solver's rank, serial exchange", ["--problem", "shuttle", "--sections", "300", "--order", "4", "--gather-root"], {}),
]


@pytest.mark.parametrize("label,extra,expect", CASES, ids=[c[0] for c in CASES])
def test_two_ranks_reassemble_the_unsharded_evaluation(built, label, extra, expect):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--check",
           "--steps", "3", "--warmup", "1", "--no-cpu", "--no-host", "--no-sharded-configs", "--no-pin"] + extra
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    out = res.stdout + res.stderr
    assert res.returncode == 0, out[-3000:]
    assert out.count("sharded == unsharded: True") == (1 if "--gather-root" in extra else 2), out[-3000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["world_size"] == 2 and d["config"]["backend"] == "gloo"
    for k, v in expect.items():
        assert d["config"][k] == v, (k, d["config"].get(k))


_RCCL_WORLD_OF_ONE = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from pycollo_amd import problems
from pycollo_amd.engine import NlpEngine
from pycollo_amd.sharding import ShardedNlp
for name, kw in (("delta_iii", dict(K=40, order=4)), ("shuttle", dict(K=300, order=4))):
    prob = problems.REGISTRY[name](**kw)
    sh = ShardedNlp(prob, device=0)
    sh.always_exchange = True
    dev = torch.device("cuda", 0)
    x = torch.from_numpy(np.random.default_rng(1).uniform(0.05, 0.3, sh.num_x)).to(dev)
    lam = torch.from_numpy(np.random.default_rng(2).normal(size=sh.num_c)).to(dev)
    ref = NlpEngine(prob, device=0, threads_per_block=sh.engine.info["threads_per_block"])
    rc, rG, rH = (torch.empty(n, dtype=torch.float64, device=dev) for n in (ref.num_c, ref.nnz_jac, ref.nnz_hess))
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        ref.evaluate_all_device(x, 1.0, lam, rc, rG, rH, st)
    st.synchronize()
    for mode in (dict(), dict(overlap=True), dict(unpadded=True), dict(overlap=True, unpadded=True), dict(root=0)):
        sh.buf.zero_()
        with torch.cuda.stream(st):
            c, G, H = sh.evaluate_all_device(x, 1.0, lam, st, **mode)
        torch.cuda.synchronize()
        same = lambda a, b: bool(torch.equal(torch.nan_to_num(a, nan=1.25e300), torch.nan_to_num(b, nan=1.25e300)))
        ok = same(rc, c) and same(rG, G) and same(rH, H)
        print(f"RCCL {name} {mode}: {ok}", flush=True)
        if not ok:
            sys.exit(1)
    ref.close()
dist.destroy_process_group()
print("RCCL world of one: all modes equal")
'''


def test_rccl_collectives_carry_the_exchange(built):
    """The RCCL (backend "nccl") path itself -- all_gather_into_tensor, broadcast, gather on device buffers, ordered
    against the tile kernels by streams and events -- with the one rank a one-GPU box can give it: every output of the
    rank travels through the send / receive buffers and the collectives, and must come back bit-identical to the
    unsharded evaluation, in every exchange form."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    res = subprocess.run([sys.executable, "-c", _RCCL_WORLD_OF_ONE], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, (res.stdout + res.stderr)[-3000:]
    assert "RCCL world of one: all modes equal" in res.stdout


_LOCAL_TWO_RANKS = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("gloo")
rank = dist.get_rank()
from pycollo_amd import problems
from pycollo_amd.engine import NlpEngine
from pycollo_amd.sharding import LocalShardedNlp
dev = torch.device("cuda", 0)
for name, kw in (("shuttle", dict(K=300, order=4)), ("delta_iii", dict(K=40, order=5)), ("time_coupled_transfer", dict(K=60, order=4))):
    prob = problems.REGISTRY[name](**kw)
    sh = LocalShardedNlp(prob, device=0, root=0)
    st = torch.cuda.Stream()
    x = lam = None
    if sh.is_root:
        lo, hi = (0.05, 0.3) if name == "delta_iii" else (-0.45, 0.45)
        x = torch.from_numpy(np.random.default_rng(1).uniform(lo, hi, sh.num_x)).to(dev)
        lam = torch.from_numpy(np.random.default_rng(2).normal(size=sh.num_c)).to(dev)
    for rep in range(2):
        with torch.cuda.stream(st):
            out = sh.evaluate_all_device(x, 0.8, lam)
        torch.cuda.synchronize()
    whole = None
    if sh.is_root:
        ref = NlpEngine(prob, device=0)
        rc, rG, rH = (torch.empty(n, dtype=torch.float64, device=dev) for n in (ref.num_c, ref.nnz_jac, ref.nnz_hess))
        with torch.cuda.stream(st):
            ref.evaluate_all_device(x, 0.8, lam, rc, rG, rH, st.cuda_stream)
        st.synchronize()
        same = lambda a, b: bool(torch.equal(torch.nan_to_num(a, nan=1.25e300), torch.nan_to_num(b, nan=1.25e300)))
        ok = same(rc, out[0]) and same(rG, out[1]) and same(rH, out[2])
        whole = 8 * (ref.num_x + 2 * ref.num_c + 2 * (ref.nnz_jac + ref.nnz_hess))
        print(f"LOCAL {name}: root reassembled the unsharded evaluation: {ok}", flush=True)
        ref.close()
        if not ok:
            sys.exit(1)
    else:
        assert out is None and sh.engine is None
    print(f"LOCAL {name}: rank {rank} holds {sh.shard.device_bytes()} B of its own", flush=True)
    sh.close()
dist.barrier()
dist.destroy_process_group()
print(f"LOCAL rank {rank} done")
'''


def test_rank_local_shards_two_processes(built, tmp_path):
    """``LocalShardedNlp`` as processes run it (two ranks under torch.distributed.run sharing the one GPU, gloo): the root
    scatters the ranks' slices of x~ / lambda, every rank evaluates its section range on a handle that holds nothing else,
    one gather brings the packed segments to the root, whose tail finishes the evaluation -- bit for bit the unsharded one.
    Only the root holds the whole NLP."""
    script = tmp_path / "local_two_ranks.py"
    script.write_text(_LOCAL_TWO_RANKS)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)]
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    out = res.stdout + res.stderr
    assert res.returncode == 0, out[-3000:]
    assert out.count("root reassembled the unsharded evaluation: True") == 3, out[-3000:]
    assert "LOCAL rank 0 done" in out and "LOCAL rank 1 done" in out
