"""Development aid: wall time of pc_kkt_factor / pc_kkt_solve / pc_kkt_matvec at config-2 size for several leaf groupings."""
import os, sys, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'tests'))
import numpy as np
from test_kkt_cpu import kkt_case
from pycollo_amd.kkt import GpuKkt
eng, ora, x, lam, ineq, fixed, sc, dvec = kkt_case("hypersensitive", dict(K=2000, order=6), device=0)
eng.evaluate_resident(x, 1.0, lam)
groups = [None if a == 'auto' else int(a) for a in sys.argv[1:]] or [None, 8, 16]
for g in groups:
    t0=time.perf_counter(); k = GpuKkt(eng, ineq, fixed, sc, group=g); t1=time.perf_counter()
    rhs=np.random.default_rng(0).normal(size=k.nu)
    for name,fn in (("factor",lambda: k.factor(dvec)),("solve",lambda: k.solve(rhs)),("matvec",lambda: k.matvec(dvec,rhs))):
        fn(); ts=[]
        for _ in range(10):
            t=time.perf_counter(); fn(); ts.append(time.perf_counter()-t)
        print("group",g,"leaves",k.tables.n_leaf,name,"median ms",round(1e3*float(np.median(ts)),3),"min",round(1e3*min(ts),3), flush=True)
    print("   create s",round(t1-t0,3))
    k.close()
