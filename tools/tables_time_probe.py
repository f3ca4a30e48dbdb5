"""Probe: kkt.build_tables timed in the contexts solve_with_ipm creates (gc frozen, BLAS limited) -- development aid."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pycollo_amd import problems, kkt
from pycollo_amd.iteration import MeshIteration
from pycollo_amd.engine import PycolloGpuProblem
from pycollo_amd.ipm import GpuInteriorPointSolver

it = MeshIteration(problems.hypersensitive(K=2000, order=6), device=0)
pobj = PycolloGpuProblem(it.engine)
s = GpuInteriorPointSolver(pobj, pobj.n, pobj.m, it.x_bnd_l, it.x_bnd_u, it.c_bnd_l, it.c_bnd_u)
def t(label):
    t0 = time.perf_counter(); kkt.build_tables(it.engine, s.ineq, s.fixed, s.sc); print(f"{label:40s} {1e3 * (time.perf_counter() - t0):7.1f} ms", flush=True)
t("first"); t("second"); t("third")
gc.freeze(); gc.disable(); t("gc frozen + disabled"); t("again")
from threadpoolctl import threadpool_limits
lim = threadpool_limits(limits=8); t("BLAS limited to 8"); t("again"); lim.restore_original_limits()
gc.enable(); gc.unfreeze(); t("restored")
x = np.random.default_rng(0).normal(size=30000); y = float(x @ x); t("after a 30k dot product"); t("again")
it2 = MeshIteration(problems.hypersensitive(K=2000, order=6), device=0); t("after building another MeshIteration"); t("again")
