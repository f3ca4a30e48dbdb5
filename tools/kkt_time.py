"""Development aid: wall time of pc_kkt_factor / pc_kkt_solve / pc_kkt_matvec / pc_kkt_solve_refined at config-2 size;
under rocprofv3 --kernel-trace --stats it gives the per-kernel table (profiles/r0N_kkt_*)."""
import os, sys, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'tests'))
import numpy as np
from test_kkt_cpu import kkt_case
from pycollo_amd.kkt import GpuKkt
name = os.environ.get("KKT_PROBLEM", "hypersensitive")
kw = dict(K=int(os.environ.get("KKT_K", "2000")), order=int(os.environ.get("KKT_ORDER", "6")))
eng, ora, x, lam, ineq, fixed, sc, dvec = kkt_case(name, kw, device=0)
eng.evaluate_resident(x, 1.0, lam)
groups = [None if a == 'auto' else int(a) for a in sys.argv[1:]] or [None]
for g in groups:
    t0=time.perf_counter(); k = GpuKkt(eng, ineq, fixed, sc, group=g); t1=time.perf_counter()
    rhs=np.random.default_rng(0).normal(size=k.nu)
    for nm,fn in (("factor",lambda: k.factor(dvec)),("solve",lambda: k.solve(rhs)),("matvec",lambda: k.matvec(dvec,rhs)),
                  ("solve_refined",lambda: k.solve_refined(rhs, dvec))):
        fn(); ts=[]
        for _ in range(20):
            t=time.perf_counter(); fn(); ts.append(time.perf_counter()-t)
        print(name, kw, "group",g,"leaves",k.tables.n_leaf,nm,"median ms",round(1e3*float(np.median(ts)),3),"min",round(1e3*min(ts),3), flush=True)
    print("   create s",round(t1-t0,3), "refined back-substitutions", k.solve_refined(rhs, dvec)[1])
    k.close()
    # the same tables a second time in the process: host table build against the device-side create
    from pycollo_amd import kkt as _kkt
    t0=time.perf_counter(); _kkt.build_tables(eng, ineq, fixed, sc, g); t1=time.perf_counter()
    k2 = GpuKkt(eng, ineq, fixed, sc, group=g); t2=time.perf_counter()
    print("   second create: build_tables ms", round(1e3*(t1-t0),1), "| GpuKkt (tables + pc_kkt_create) ms", round(1e3*(t2-t1),1), flush=True)
    k2.close()
