"""Diagnostic: the free-flying robot at mesh tolerance 1e-6 with the interior-point log of every NLP (tools/, not product)."""
import sys
import warnings

from pycollo_amd import problems
from pycollo_amd.solve import solve_ocp

warnings.simplefilter("always")
stop = int(sys.argv[1]) if len(sys.argv) > 1 else 6
res = solve_ocp(problems.free_flying_robot(), mesh_tolerance=1e-6, max_mesh_iterations=stop, verbose=2)
print("objective", res.objective, "gave up", res.gpu_linear_solver_gave_up)
