"""Quick GPU parity sweep (development aid; the real tests live in tests/)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle.ref_numpy import OracleNlp
from pycollo_amd import problems
from pycollo_amd.engine import NlpEngine
from pycollo_amd.quadrature import QuadratureTables

tab = QuadratureTables("lobatto")
cases = [("brachistochrone", {}), ("hypersensitive", dict(K=2000, order=6)), ("cart_pole", dict(K=50, order=4)),
         ("shuttle", dict(K=30, order=5)), ("double_pendulum", {}), ("two_phase_transfer", {}),
         ("delta_iii", dict(K=7, order=4))]
bad = 0
for name, kw in cases:
    prob = problems.REGISTRY[name](**kw)
    for tb in (64, 256):
        eng = NlpEngine(prob, device=0, threads_per_block=tb)
        rng = np.random.default_rng(1)
        W = rng.uniform(0.5, 2.0, eng.layout.num_ocp_c)
        eng.set_scaling(eng.V_ocp, eng.r_ocp, W, 1.7)
        ora = OracleNlp(prob, tab, V_ocp=eng.V_ocp, r_ocp=eng.r_ocp, W_ocp=W, w_J=1.7)
        x = rng.uniform(-0.45, 0.45, eng.num_x)
        lam = rng.normal(size=eng.num_c)
        sigma = 0.6
        c, G, H = eng.evaluate_all(x, sigma, lam)
        res = {}
        for nm, got, ref in (("c", c, ora.c(x)), ("G", G, ora.G(x)), ("H", H, ora.H(x, sigma, lam)),
                             ("J", np.array([eng.evaluate_J(x)]), np.array([ora.J(x)])),
                             ("g", eng.evaluate_g(x), ora.grad_J(x)),
                             ("c2", eng.evaluate_c(x), ora.c(x)), ("G2", eng.evaluate_G_nonzeros(x, new_x=False), ora.G(x)),
                             ("H2", eng.evaluate_H_nonzeros(x, sigma, lam), ora.H(x, sigma, lam)),
                             ("rn", eng.G_row_norms(x), ora.G_row_norms(x))):
            scale = max(1e-300, np.max(np.abs(ref)))
            res[nm] = float(np.max(np.abs(got - ref)) / scale) if ref.size else 0.0
        ok = all(v < 1e-10 for v in res.values())
        bad += (not ok)
        print(name, "TB", tb, "tiles", eng.info["n_tiles_total"], "OK" if ok else "FAIL", {k: f"{v:.1e}" for k, v in res.items()}, flush=True)
        eng.close()
print("FAILURES", bad)
sys.exit(1 if bad else 0)
