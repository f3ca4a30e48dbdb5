"""Config 5 as BASELINE.json states it -- Delta III, four phases, ph-adaptive refinement to ~50 k nodes -- with the
build's own refinement loop: the example's guess trajectory is carried from mesh to mesh (pc_interp_linear), its
mesh error is estimated on the GPU (pc_mesh_err_p<i>), and next_phase_mesh produces the next mesh, with the mesh
tolerance tightened until the node count reaches the target.  (The NLP itself is not solved on the way: the stand-in
solver does not converge Delta III)  Prints the mesh sequence; --save writes the final mesh.

    python tools/refine_delta_iii.py [target_nodes] [--save path.npz]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402


def refine_to(target_nodes=50000, verbose=True, max_rounds=25):
    import copy
    from pycollo_amd import problems
    from pycollo_amd.iteration import MeshIteration
    from pycollo_amd.refinement import mesh_error, next_phase_mesh
    prob = copy.deepcopy(problems.delta_iii(K=10, order=4))
    prev, tol, log = None, 1e-3, []
    for k in range(max_rounds):
        t0 = time.perf_counter()
        it = MeshIteration(prob, device=0, prev=prev, number=k + 1)
        it.x_tilde = it.guess_x_tilde                    # the carried trajectory stands where a solution would
        errs = mesh_error(it.engine, it.x_tilde)
        N = [int(pl.N) for pl in it.layout.phases]
        worst = max(float(np.max(rel)) for rel, _ in errs)
        log.append({"round": k + 1, "K": [int(m.K) for m in it.meshes], "N": N, "tol": tol, "max_rel_err": worst})
        if verbose:
            orders = sorted({int(n) for m in it.meshes for n in np.unique(m.n)})
            print(f"round {k + 1}: K={log[-1]['K']} N={N} total {sum(N)} orders {orders} tol {tol:.1e} "
                  f"max rel err {worst:.2e} ({time.perf_counter() - t0:.2f} s)", flush=True)
        if sum(N) >= target_nodes:
            meshes = [(np.asarray(m.sizes, float), np.asarray(m.n, np.int64)) for m in it.meshes]
            it.engine.close()
            return meshes, log
        new, done_all = [], True
        for mesh, (rel, _) in zip(it.meshes, errs):
            sizes, nodes, done = next_phase_mesh(mesh.sizes, mesh.n, rel, mesh_tol=tol)
            new.append((sizes, nodes))
            done_all = done_all and done
        if done_all or sum(int(np.sum(n - 1)) + 1 for _, n in new) <= sum(N):
            tol *= 0.1                                   # this tolerance is met (or no longer adds nodes): tighten it
        prev = it.solution()
        it.engine.close()
        for ph, (sizes, nodes) in zip(prob.phases, new):
            ph.mesh.number_mesh_sections = len(nodes)
            ph.mesh.mesh_section_sizes = sizes
            ph.mesh.number_mesh_section_nodes = nodes
    raise RuntimeError("target node count not reached")


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    meshes, log = refine_to(int(args[0]) if args else 50000)
    if "--save" in sys.argv:
        path = sys.argv[sys.argv.index("--save") + 1]
        np.savez_compressed(path, **{f"sizes{i}": s for i, (s, _) in enumerate(meshes)},
                            **{f"nodes{i}": n for i, (_, n) in enumerate(meshes)})
        print("saved", path)
