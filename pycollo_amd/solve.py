"""The outer solve loop: mesh iteration -> NLP solve -> ph mesh-error estimate -> next mesh.

SURVEY.md section 8f rows N1-N3 put together, i.e. what ``OptimalControlProblem.solve()`` does around the hot
path (pycollo/optimal_control_problem.py:497-541: ``_solve_iteration`` in a loop until the mesh tolerance is met
or ``max_mesh_iterations`` is reached; each pass = ``Iteration.solve`` (pycollo/iteration.py:474-526) followed by
``PattersonRaoMeshRefinement`` (pycollo/mesh_refinement.py:61-392)).  Every NLP callback, the guess interpolation,
the constraint-scaling row norms and the mesh-error estimate run on the GPU; the NLP solver is the interior-point
stand-in of ``pycollo_amd.ipm`` (IPOPT is not installed in this image).
"""
from __future__ import annotations

import copy
from dataclasses import dataclass, field

import numpy as np

from .iteration import MeshIteration
from .refinement import MESH_TOLERANCE, mesh_error, next_phase_mesh


@dataclass
class OcpResult:
    objective: float
    mesh_tolerance_met: bool
    mesh_iterations: int
    iterations: list = field(default_factory=list)   # per mesh iteration: dict(K, N, objective, status, nlp_iterations, max_rel_err)
    final: MeshIteration | None = None
    gpu_linear_solver_gave_up: list = field(default_factory=list)   # mesh iterations whose NLP was repeated with the host factorisation
    # mesh iterations whose NLP ended "acceptable" because a line search failed at a point already inside the acceptable
    # tolerances (ipm.py, IPOPT's STOP_AT_ACCEPTABLE_POINT): converged to acceptable_tol (1e-6), not to nlp_tol
    acceptable_after_failed_line_search: list = field(default_factory=list)


def solve_ocp(problem, *, max_mesh_iterations: int = 10, mesh_tolerance: float = MESH_TOLERANCE, device: int = 0,
              nlp_tol: float = 1e-10, nlp_max_iter: int = 2000, verbose: int = 0, update_scaling: bool = False,
              scaling_weight: float = 0.8, linear_solver: str = "resident", warm_start: bool = False,
              host_retry: bool = False) -> OcpResult:
    """Solve ``problem`` (a :class:`pycollo_amd.problem.ProblemSpec`) on its initial mesh, refine, repeat.
    ``nlp_tol`` / ``nlp_max_iter``: the reference's defaults (pycollo/settings.py:60-61: 1e-10, 2000) -- with 1e-8 the tumour
    problem stops 2e-7 short of its mesh tolerance after ten mesh iterations, with 1e-10 it meets it as the reference does;
    ``update_scaling`` / ``scaling_weight``: pycollo/settings.py:272-296 (scalings averaged over the mesh iterations);
    ``warm_start``: pycollo/settings.py:228 (IPOPT's ``warm_start_init_point``, every mesh iteration);
    ``host_retry``: repeat a GPU-factorised NLP solve that did not succeed with the host factorisation (off: a failed
    solve ends the loop and the result says so)."""
    import os
    linear_solver = os.environ.get("PYCOLLO_AMD_LINEAR_SOLVER", linear_solver)   # (A/B knob: "gpu" | "resident" | "host")
    prob = copy.deepcopy(problem)
    prev = None
    log = []
    it = None
    met = False
    history = []
    gave_up = []
    soft = []
    for k in range(max_mesh_iterations):
        it = MeshIteration(prob, device=device, prev=prev, number=k + 1, update_scaling=update_scaling,
                           scaling_weight=scaling_weight, history=history)
        history.append(it.scaling_record)
        res = it.solve_with_ipm(max_iter=nlp_max_iter, tol=nlp_tol, verbose=max(0, verbose - 1), linear_solver=linear_solver,
                                warm_start=warm_start, host_retry=host_retry)
        errs = mesh_error(it.engine, it.x_tilde)
        worst = max(float(np.max(rel)) for rel, _ in errs)
        log.append({"K": [int(m.K) for m in it.meshes], "N": [int(pl.N) for pl in it.layout.phases],
                    "objective": float(it.objective), "status": res.status, "nlp_iterations": int(res.iterations),
                    "max_rel_err": worst, "seconds": float(res.seconds), "evaluations": dict(res.evaluations)})
        if "gpu_linear_solver_gave_up" in res.evaluations:
            gave_up.append(k + 1)
        if res.evaluations.get("acceptable_after_failed_line_search"):
            import warnings
            soft.append(k + 1)
            warnings.warn(f"mesh iteration {k + 1}: the NLP solve stopped at an acceptable point after a failed line search "
                          f"(inf_pr {res.inf_pr:.1e}, inf_du {res.inf_du:.1e}); the requested tolerance was {nlp_tol:g}",
                          RuntimeWarning, stacklevel=2)
        if verbose:
            print(f"mesh iteration {k + 1}: K={log[-1]['K']} N={log[-1]['N']} J={it.objective:.10g} "
                  f"[{res.status}, {res.iterations} NLP iterations, {res.seconds:.2f} s] max rel. mesh error {worst:.3e}", flush=True)
        if not res.success:
            break
        done_all = True
        new_meshes = []
        for mesh, (rel, _) in zip(it.meshes, errs):
            sizes, nodes, done = next_phase_mesh(mesh.sizes, mesh.n, rel, mesh_tol=mesh_tolerance)
            done_all = done_all and done
            new_meshes.append((sizes, nodes))
        if done_all:
            met = True
            break
        prev = it.solution()
        for ph, (sizes, nodes) in zip(prob.phases, new_meshes):
            ph.mesh.number_mesh_sections = len(nodes)
            ph.mesh.mesh_section_sizes = sizes
            ph.mesh.number_mesh_section_nodes = nodes
    return OcpResult(objective=float(it.objective), mesh_tolerance_met=met, mesh_iterations=len(log), iterations=log, final=it,
                     gpu_linear_solver_gave_up=gave_up, acceptable_after_failed_line_search=soft)
