import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from oracle.ref_numpy import OracleNlp
from pycollo_amd import problems
from pycollo_amd.quadrature import QuadratureTables
from pycollo_amd.engine import NlpEngine
tab = QuadratureTables("lobatto")
rng = np.random.default_rng(5)
K = 157
prob = problems.two_phase_transfer()
A, B = prob.phases
A.mesh.number_mesh_sections = K
A.mesh.mesh_section_sizes = rng.uniform(0.2, 1.0, K)
A.mesh.number_mesh_section_nodes = rng.integers(2, 11, K)
B.mesh.number_mesh_sections = 3
B.mesh.mesh_section_sizes = [0.2, 0.5, 0.3]
B.mesh.number_mesh_section_nodes = [10, 2, 7]
np.set_printoptions(linewidth=200, precision=6)
for defs in ("", "PC_PIN_BUDGET=400"):
    os.environ["PYCOLLO_AMD_DEFINES"] = defs
    eng = NlpEngine(prob, device=0, threads_per_block=64)
    ora = OracleNlp(prob, tab, V_ocp=eng.V_ocp, r_ocp=eng.r_ocp, W_ocp=eng.W_ocp, w_J=1.0)
    r = np.random.default_rng(64)
    x = r.uniform(-0.45, 0.45, eng.num_x); lam = r.normal(size=eng.num_c)
    cr, Gr, Hr = ora.c(x), ora.G(x), ora.H(x, 0.6, lam)
    c, G, H = eng.evaluate_all(x, 0.6, lam)
    print(f"[{defs}] wpt {eng.info['waves_per_tile']} max|c-cr| {np.abs(c-cr).max():.3e} max|G-Gr| {np.abs(G-Gr).max():.3e} max|H-Hr| {np.abs(H-Hr).max():.3e}")
    print(" c  ", c[:12]); print(" cr ", cr[:12])
    print(" G  ", G[:12]); print(" Gr ", Gr[:12])
    eng.close()
