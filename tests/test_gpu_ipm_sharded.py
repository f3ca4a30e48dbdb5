"""An NLP solved by several ranks (pycollo_amd/ipm_sharded.py): the interior-point loop on replicated vectors over the
section-sharded evaluation and the KKT factorisation cut across ranks -- against the single-process solver with the GPU
factorisation (ipm.GpuInteriorPointSolver): same optimum, same iteration count.  Two processes share the one GPU over gloo;
every rank's G~ / H~ buffer is NaN wherever its own tiles and the tail do not write."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from pycollo_amd import problems
from test_gpu_sharded_process import _free_port

pytestmark = pytest.mark.gpu


def _keep(lines: str, name: str):
    """The ranks' report lines, kept beside the other run outputs when the scratch directory exists (gpurun_out/)."""
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, name), "w") as f:
            f.write(lines + "\n")


@pytest.mark.parametrize("name,kw", [("hypersensitive", dict(K=200, order=6)), ("cart_pole", dict(K=100, order=4)),
                                     ("sliding_mass", dict(num_phases=2, K=10, order=4))])
def test_a_world_of_one_takes_the_single_process_path(built, monkeypatch, name, kw):
    """No other rank: the sharded solver's own plumbing (device-launch evaluation with the objective read from the tail,
    the adapter's refinement rule, the plan without cuts) must reproduce the host-vector loop's iterates."""
    from pycollo_amd.ipm_sharded import solve_sharded
    from pycollo_amd.iteration import MeshIteration
    monkeypatch.setenv("PYCOLLO_AMD_KKT_RESID_TOL", "0")
    a = MeshIteration(problems.REGISTRY[name](**kw), device=0).solve_with_ipm(max_iter=1500, tol=1e-8, linear_solver="gpu")
    b, sh = solve_sharded(MeshIteration(problems.REGISTRY[name](**kw), device=0), max_iter=1500, tol=1e-8)
    assert a.status == b.status == "optimal"
    assert a.iterations == b.iterations
    assert np.max(np.abs(a.x - b.x)) <= 1e-9 * max(1.0, float(np.max(np.abs(a.x))))
    assert abs(a.objective - b.objective) <= 1e-9 * max(1.0, abs(a.objective))


_TWO_RANKS = r'''
import os, sys, traceback
def _excepthook(t, v, tb):
    with open(os.environ["IPM_LOG"] + f".{os.environ.get('RANK', '0')}", "w") as f:
        traceback.print_exception(t, v, tb, file=f)
    traceback.print_exception(t, v, tb)
sys.excepthook = _excepthook
import numpy as np
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("gloo")
rank = dist.get_rank()
from pycollo_amd import problems
from pycollo_amd.ipm_sharded import solve_sharded
from pycollo_amd.iteration import MeshIteration
CASES = {2: (("hypersensitive", dict(K=400, order=6)), ("cart_pole", dict(K=300, order=4)), ("sliding_mass", dict(num_phases=3, K=120, order=5))),
         4: (("hypersensitive", dict(K=800, order=6)), ("sliding_mass", dict(num_phases=2, K=400, order=4)))}[dist.get_world_size()]
for name, kw in CASES:
    prob = problems.REGISTRY[name](**kw)
    b, sh = solve_sharded(MeshIteration(prob, device=0), max_iter=1500, tol=1e-8, poison=True)
    cuts = [len(c) for c in b.evaluations.get("cuts", [])]
    a = MeshIteration(prob, device=0).solve_with_ipm(max_iter=1500, tol=1e-8, linear_solver="gpu")
    dx = float(np.max(np.abs(a.x - b.x)) / max(1.0, float(np.max(np.abs(a.x)))))
    df = abs(a.objective - b.objective) / max(1.0, abs(a.objective))
    ok = a.status == b.status == "optimal" and dx <= 1e-6 and df <= 1e-9 and abs(a.iterations - b.iterations) <= 2
    busy = sum(1 for r in range(sh.world) if any(te > tb for tb, te in sh.plan.tile_ranges[r]))
    print(f"SHARDED IPM {name} rank {rank}: {busy} ranks with tiles, iterations {b.iterations} (single process {a.iterations}), "
          f"objective difference {df:.1e}, x difference {dx:.1e}, {b.evaluations['sharded']['evaluations']} sharded evaluations, ok: {ok}", flush=True)
    if not ok or busy < dist.get_world_size():
        sys.exit(1)
dist.barrier()
dist.destroy_process_group()
print(f"SHARDED IPM rank {rank} done")
'''


@pytest.mark.parametrize("nproc,n_cases", [(2, 3), (4, 2)])
def test_several_processes_solve_one_nlp(built, tmp_path, nproc, n_cases):
    """``nproc`` = 4: the middle ranks keep BOTH ends of their chain segments un-eliminated (a first node and an anchor)."""
    script = tmp_path / "ipm_two_ranks.py"
    script.write_text(_TWO_RANKS)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", PYCOLLO_AMD_KKT_RESID_TOL="0",
               PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), IPM_LOG=str(tmp_path / "trace"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)]
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1100)
    out = res.stdout + res.stderr
    traces = "".join(p.read_text() for p in tmp_path.glob("trace.*"))
    lines = "\n".join(ln for ln in out.splitlines() if "SHARDED IPM" in ln)
    assert res.returncode == 0, traces + lines + out[-1500:]
    assert out.count("ok: True") == nproc * n_cases, lines
    _keep(lines, f"sharded_ipm_{nproc}_ranks.txt")
    assert all(f"SHARDED IPM rank {r} done" in out for r in range(nproc))


_OCP_TWO_RANKS = r'''
import os, sys, traceback
def _excepthook(t, v, tb):
    with open(os.environ["IPM_LOG"] + f".{os.environ.get('RANK', '0')}", "w") as f:
        traceback.print_exception(t, v, tb, file=f)
    traceback.print_exception(t, v, tb)
sys.excepthook = _excepthook
import numpy as np
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("gloo")
rank = dist.get_rank()
from pycollo_amd import problems
from pycollo_amd.solve import solve_ocp
prob = problems.hypersensitive(K=300, order=4)
b = solve_ocp(prob, linear_solver="sharded", max_mesh_iterations=6)
a = solve_ocp(prob, linear_solver="gpu", max_mesh_iterations=6)
same = a.mesh_iterations == b.mesh_iterations and abs(a.objective - b.objective) <= 1e-9 * abs(a.objective)
ref = abs(b.objective - 3.36206) <= 1e-5 * 3.36206          # the reference's assertion, tests/integration/test_hypersensitive_problem.py:129-130
meshes = [it["N"][0] for it in b.iterations]
print(f"SHARDED OCP rank {rank}: objective {b.objective:.8f} (single process {a.objective:.8f}), mesh iterations {b.mesh_iterations} / {a.mesh_iterations}, "
      f"nodes {meshes}, tolerance met {b.mesh_tolerance_met}, ok: {bool(same and ref and b.mesh_tolerance_met)}", flush=True)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if (same and ref and b.mesh_tolerance_met) else 1)
'''


def test_two_processes_solve_an_optimal_control_problem_through_the_mesh_loop(built, tmp_path):
    """``solve_ocp(..., linear_solver="sharded")`` on two ranks: every mesh iteration's NLP solved by both, the engine handed back
    whole for the mesh-error estimate and the next mesh -- the reference's hypersensitive objective (rtol 1e-5) and the
    single-process run's, mesh for mesh."""
    script = tmp_path / "ocp_two_ranks.py"
    script.write_text(_OCP_TWO_RANKS)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0",
               PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), IPM_LOG=str(tmp_path / "trace"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)]
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1100)
    out = res.stdout + res.stderr
    traces = "".join(p.read_text() for p in tmp_path.glob("trace.*"))
    lines = "\n".join(ln for ln in out.splitlines() if "SHARDED OCP" in ln)
    assert res.returncode == 0, traces + lines + out[-1500:]
    assert out.count("ok: True") == 2, lines
    _keep(lines, "sharded_ocp_2_ranks.txt")
