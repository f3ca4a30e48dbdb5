"""Per-block error breakdown against the oracle (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle.ref_numpy import OracleNlp
from pycollo_amd import problems
from pycollo_amd.engine import NlpEngine
from pycollo_amd.quadrature import QuadratureTables

tab = QuadratureTables("lobatto")
names = sys.argv[1:] or ["hypersensitive", "brachistochrone"]
kws = {"hypersensitive": dict(K=5, order=4), "brachistochrone": dict(K=3, order=4), "cart_pole": dict(K=3, order=4),
       "two_phase_transfer": {}, "double_pendulum": dict(K=3, order=3), "shuttle": dict(K=3, order=4), "delta_iii": dict(K=2, order=3)}
np.set_printoptions(linewidth=200, precision=4)
for name in names:
    prob = problems.REGISTRY[name](**kws.get(name, {}))
    eng = NlpEngine(prob, device=0)
    ora = OracleNlp(prob, tab, V_ocp=eng.V_ocp, r_ocp=eng.r_ocp, W_ocp=eng.W_ocp, w_J=1.0)
    rng = np.random.default_rng(1)
    x = rng.uniform(-0.45, 0.45, eng.num_x)
    lam = rng.normal(size=eng.num_c)
    c, G, H = eng.evaluate_all(x, 0.6, lam)
    cr, Gr, Hr = ora.c(x), ora.G(x), ora.H(x, 0.6, lam)
    print("====", name, "num_x", eng.num_x, "num_c", eng.num_c)
    bad = np.nonzero(np.abs(c - cr) > 1e-9 * (1 + np.abs(cr)))[0]
    print("c bad rows:", bad[:40], "of", eng.num_c)
    for i in bad[:10]:
        print("   row", i, "got", c[i], "ref", cr[i])
    r, cc = eng.evaluate_G_structure()
    bad = np.nonzero(np.abs(G - Gr) > 1e-9 * (1 + np.abs(Gr)))[0]
    print("G bad:", len(bad), "of", len(G))
    for i in bad[:12]:
        print("   (row %d, col %d) got %.6g ref %.6g" % (r[i], cc[i], G[i], Gr[i]))
    r, cc = eng.evaluate_H_structure()
    bad = np.nonzero(np.abs(H - Hr) > 1e-9 * (1 + np.abs(Hr)))[0]
    print("H bad:", len(bad), "of", len(H))
    for i in bad[:16]:
        print("   (row %d, col %d) got %.6g ref %.6g ratio %.4g" % (r[i], cc[i], H[i], Hr[i], H[i] / Hr[i] if Hr[i] else np.nan))
    eng.close()
