"""Development aid: solve -> estimate -> refine loop on the brachistochrone problem (scipy stand-in solver)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pycollo_amd import problems
from pycollo_amd.iteration import MeshIteration
from pycollo_amd.refinement import mesh_error, next_phase_mesh
K0 = int(sys.argv[1]) if len(sys.argv) > 1 else 3
prob = problems.brachistochrone(K=K0, order=4)
prev = None
for itn in range(4):
    it = MeshIteration(prob, prev=prev)
    res = it.solve_with_scipy(maxiter=1000)
    (rel, _), = mesh_error(it.engine, it.x_tilde)
    print(f"iter {itn}: K={it.meshes[0].K} N={it.layout.phases[0].N} J={it.objective:.10f} viol={res.constr_violation:.2e} "
          f"status={res.status} nit={res.nit} max_rel_err={rel.max():.3e}", flush=True)
    sizes, nodes, done = next_phase_mesh(it.meshes[0].sizes, it.meshes[0].n, rel)
    if done:
        print("mesh tolerance met"); break
    x = it.V * it.x_tilde + it.r
    pl = it.layout.phases[0]
    prev = ([it.meshes[0].tau], [x[pl.x_off:pl.x_off + 3 * pl.N].reshape(3, -1)], [x[pl.x_off + 3 * pl.N:pl.q_off].reshape(1, -1)],
            [np.zeros(0)], [x[pl.t_off:pl.t_off + 1]], np.zeros(0))
    prob = problems.brachistochrone()
    prob.phases[0].mesh.number_mesh_sections = len(nodes)
    prob.phases[0].mesh.mesh_section_sizes = sizes
    prob.phases[0].mesh.number_mesh_section_nodes = nodes
