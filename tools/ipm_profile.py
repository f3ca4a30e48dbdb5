import sys, os, cProfile, pstats, time
sys.path.insert(0, '/root/repo' if os.path.isdir('/root/repo/pycollo_amd') else os.getcwd())
from pycollo_amd import problems
from pycollo_amd.iteration import MeshIteration
it = MeshIteration(problems.hypersensitive(K=2000, order=6), device=0)
it.solve_with_ipm(max_iter=3, tol=1e-8, linear_solver="gpu")   # warm everything
it2 = MeshIteration(problems.hypersensitive(K=2000, order=6), device=0)
pr = cProfile.Profile(); pr.enable(); t=time.perf_counter()
res = it2.solve_with_ipm(max_iter=200, tol=1e-8, linear_solver="gpu")
print("wall", time.perf_counter()-t, res.status, res.iterations); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
