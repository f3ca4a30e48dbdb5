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
