"""The free-flying robot at the reference's settings (tests/integration/test_free_flying_robot.py:186-204: mesh tolerance 1e-5,
at most 15 mesh iterations): objective against the two published values and the reference's rtol 1e-4."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pycollo_amd import problems  # noqa: E402
from pycollo_amd.solve import solve_ocp  # noqa: E402

for ls in (sys.argv[1:] or ["resident"]):
    res = solve_ocp(problems.free_flying_robot(), mesh_tolerance=1e-5, max_mesh_iterations=15, linear_solver=ls)
    print(f"[{ls}] objective {res.objective:.7f}  rel. to 7.9101902: {res.objective / 7.9101902 - 1:+.2e}  to 7.910154646: "
          f"{res.objective / 7.910154646 - 1:+.2e}  met {res.mesh_tolerance_met}  mesh iterations {res.mesh_iterations}  "
          f"nodes {res.iterations[-1]['N']}  statuses {[r['status'][:3] for r in res.iterations]}")
