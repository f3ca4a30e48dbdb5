"""The oracle's C port (CPU baseline of bench.py) agrees with the NumPy oracle."""
import numpy as np
import pytest

from conftest import entry_err
from pycollo_amd import problems
from pycollo_amd.quadrature import QuadratureTables


@pytest.mark.parametrize("name,kw", [("hypersensitive", dict(K=20, order=6)), ("brachistochrone", {}),
                                     ("two_phase_transfer", {}), ("double_pendulum", dict(K=4, order=5)),
                                     ("delta_iii", dict(K=3, order=3))])
def test_cport_matches_numpy_oracle(name, kw):
    from oracle.cport import CPort
    cp = CPort(problems.REGISTRY[name](**kw), QuadratureTables("lobatto"))
    o = cp.ora
    rng = np.random.default_rng(1)
    x = rng.uniform(0.05, 0.4, o.num_x)
    lam = rng.normal(size=o.num_c)
    c, G, H = cp.eval_all(x, 0.7, lam)
    # entry by entry, each held to the rounding of its own terms
    assert entry_err(c, o.c(x), o.c_mag(x), rtol=1e-12) <= 1.0
    assert entry_err(G, o.G(x), o.G_mag(x), rtol=1e-12) <= 1.0
    assert entry_err(H, o.H(x, 0.7, lam), o.H_mag(x, 0.7, lam), rtol=1e-12) <= 1.0
