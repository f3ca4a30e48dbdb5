"""Development aid: the outer solve loop (pycollo_amd.solve.solve_ocp) on a registered problem."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pycollo_amd import problems
from pycollo_amd.solve import solve_ocp
name = sys.argv[1] if len(sys.argv) > 1 else "brachistochrone"
kw = eval(sys.argv[2]) if len(sys.argv) > 2 else {}
t0 = time.time()
res = solve_ocp(problems.REGISTRY[name](**kw), verbose=1, max_mesh_iterations=int(os.environ.get("MAX_MESH", "10")),
                mesh_tolerance=float(os.environ.get("MESH_TOL", "1e-7")), nlp_tol=float(os.environ.get("NLP_TOL", "1e-8")),
                nlp_max_iter=int(os.environ.get("NLP_MAX_ITER", "1000")), linear_solver=os.environ.get("LINEAR_SOLVER", "gpu"))
for row in res.iterations:
    print("   ", {k: (v if k != "evaluations" else {kk: vv for kk, vv in v.items() if kk in ("gpu_linear_solver_gave_up", "factorisations")}) for k, v in row.items() if k not in ("K",)})
print(f"{name}: objective {res.objective:.10g}, mesh tolerance met: {res.mesh_tolerance_met}, {res.mesh_iterations} mesh iterations, {time.time() - t0:.1f} s")
