"""Delta III (BASELINE.json configs[4]'s model) solved end to end by the build's own loop: the flown guess
(problems.delta_iii_flown_guess), the interior-point stand-in with the KKT systems on the GPU, the ph mesh-error estimate and
next mesh, repeated until the mesh tolerance is met.  Prints one line per mesh iteration; --save writes the final mesh
(section widths and orders per phase) and the objective.

    python tools/solve_delta_iii.py [--mesh-tol 1e-6] [--max-mesh-iterations 12] [--K 10] [--order 4] [--save path.npz]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mesh-tol", type=float, default=None)
    ap.add_argument("--max-mesh-iterations", type=int, default=12)
    ap.add_argument("--K", type=int, default=10)
    ap.add_argument("--order", type=int, default=4)
    ap.add_argument("--nlp-tol", type=float, default=1e-10)
    ap.add_argument("--nlp-max-iter", type=int, default=2000)
    ap.add_argument("--linear-solver", default="resident")
    ap.add_argument("--example-guess", action="store_true", help="the example's own guess (on the pad): NaN partials")
    ap.add_argument("--as-published", action="store_true", help="phase D's final mass pinned to the payload: infeasible")
    ap.add_argument("--refine-to", type=int, default=0, metavar="NODES",
                    help="after the mesh tolerance is met: tighten it a decade at a time, solving again from the last solution, "
                         "until the meshes hold this many nodes (BASELINE.json configs[4]: 'refinement to ~50k nodes')")
    ap.add_argument("--verbose", type=int, default=1)
    ap.add_argument("--save", default="")
    args = ap.parse_args()
    from pycollo_amd import problems
    from pycollo_amd.solve import MESH_TOLERANCE, solve_ocp
    prob = problems.delta_iii(K=args.K, order=args.order, burnout_mass=not args.as_published)
    if not args.example_guess:
        problems.delta_iii_flown_guess(prob)
    tol = args.mesh_tol if args.mesh_tol is not None else MESH_TOLERANCE
    t0 = time.perf_counter()
    res = solve_ocp(prob, max_mesh_iterations=args.max_mesh_iterations, mesh_tolerance=tol, nlp_tol=args.nlp_tol,
                    nlp_max_iter=args.nlp_max_iter, verbose=args.verbose, linear_solver=args.linear_solver)
    wall = time.perf_counter() - t0
    print(f"Delta III: objective {res.objective:.10g} (final altitude {-res.objective / 1e3:.3f} km), mesh tolerance {tol:g} "
          f"{'met' if res.mesh_tolerance_met else 'NOT met'} after {res.mesh_iterations} mesh iterations, {wall:.1f} s")
    for i, r in enumerate(res.iterations):
        print(f"  {i + 1}: K {r['K']} N {r['N']} total {sum(r['N'])} J {r['objective']:.10g} {r['status']} "
              f"{r['nlp_iterations']} NLP iterations {r['seconds']:.2f} s, max rel. mesh error {r['max_rel_err']:.3e}")
    it = res.final
    while args.refine_to and sum(int(pl.N) for pl in it.layout.phases) < args.refine_to and tol > 1e-16:
        tol *= 0.1
        import copy
        nxt = copy.deepcopy(prob)
        taus, ys, us, _, _, _ = it.solution()
        for ph, mesh, tau, y, u in zip(nxt.phases, it.meshes, taus, ys, us):
            ta, tb = float(ph.bounds.initial_time), float(ph.bounds.final_time)
            ph.guess.time = 0.5 * (ta + tb) + 0.5 * (tb - ta) * np.asarray(tau)
            ph.guess.state_variables, ph.guess.control_variables = y, u
            ph.mesh.number_mesh_sections = int(mesh.K)
            ph.mesh.mesh_section_sizes = np.asarray(mesh.sizes, float)
            ph.mesh.number_mesh_section_nodes = np.asarray(mesh.n, np.int64)
        t0 = time.perf_counter()
        res = solve_ocp(nxt, max_mesh_iterations=args.max_mesh_iterations, mesh_tolerance=tol, nlp_tol=args.nlp_tol,
                        nlp_max_iter=args.nlp_max_iter, verbose=args.verbose, linear_solver=args.linear_solver, warm_start=True)
        it = res.final
        last = res.iterations[-1]
        print(f"tolerance {tol:.0e}: {'met' if res.mesh_tolerance_met else 'NOT met'} after {res.mesh_iterations} mesh iterations, "
              f"K {last['K']} N {last['N']} total {sum(last['N'])}, J {res.objective:.10g}, {time.perf_counter() - t0:.1f} s", flush=True)
        if not all(r["status"] in ("optimal", "acceptable") for r in res.iterations):
            print("  an NLP solve failed:", [r["status"] for r in res.iterations])
            break
    orders = [np.unique(np.asarray(m.n), return_counts=True) for m in it.meshes]
    print("  final mesh orders per phase:", [{int(o): int(c) for o, c in zip(*oc)} for oc in orders])
    if args.save:
        np.savez_compressed(args.save, objective=res.objective, met=res.mesh_tolerance_met,
                            **{f"sizes{i}": np.asarray(m.sizes, float) for i, m in enumerate(it.meshes)},
                            **{f"nodes{i}": np.asarray(m.n, np.int64) for i, m in enumerate(it.meshes)})


if __name__ == "__main__":
    main()
