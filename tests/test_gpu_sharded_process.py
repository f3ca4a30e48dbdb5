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
