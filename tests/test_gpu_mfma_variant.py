"""The optional matrix-core build of the defect contraction (-DPC_MFMA_DEFECT; SURVEY.md row X1, reference math
pycollo/backend.py:1601-1603: defect = y_0 - y_j + stretch h_k sum_i A[j][i] f(z_i)): v_mfma_f64_16x16x4_f64 in place of
the per-row multiply-add chain.  Same parity bar as the default build (entry-wise against the oracle); against the
default build itself c~ (and the time columns of G~) may differ in the last bits (fused accumulation); H~ does not
depend on the contraction and must be identical."""
import numpy as np
import pytest

from conftest import assert_matches_oracle, golden_tables
from oracle.ref_numpy import OracleNlp
from pycollo_amd import problems

pytestmark = pytest.mark.gpu

CASES = [("hypersensitive", dict(K=40, order=6)), ("hypersensitive", dict(K=13, order=6)),     # 1 state; a last tile of 1 row group
         ("cart_pole", dict(K=50, order=4)),                                                   # 4 states
         ("shuttle", dict(K=37, order=4)), ("shuttle", dict(K=2000, order=4))]                 # 5 states (+ the bench size / 10)


@pytest.mark.parametrize("name,kw", CASES)
def test_mfma_contraction_build(built, monkeypatch, name, kw):
    from pycollo_amd import codegen
    from pycollo_amd.engine import NlpEngine
    tab = golden_tables("lobatto")
    prob = problems.REGISTRY[name](**kw)
    base = NlpEngine(prob, device=0)
    monkeypatch.setenv("PYCOLLO_AMD_DEFINES", "PC_MFMA_DEFECT")
    eng = NlpEngine(prob, device=0)
    assert "_d" in eng.code_object and eng.code_object != base.code_object
    ora = OracleNlp(prob, tab, V_ocp=eng.V_ocp, r_ocp=eng.r_ocp, W_ocp=eng.W_ocp, w_J=1.0)
    rng = np.random.default_rng(5)
    x = rng.uniform(-0.45, 0.45, eng.num_x)
    lam = rng.normal(size=eng.num_c)
    c, G, H = eng.evaluate_all(x, 0.7, lam)
    assert_matches_oracle(ora, x, c=c, G=G, H=H, sigma=0.7, lam=lam)
    c0, G0, H0 = base.evaluate_all(x, 0.7, lam)
    # against the default build: H~ does not depend on the contraction; c~ and (through the time columns of a free
    # final time, W dstretch/dt h A f) G~ may differ by the rounding of a five-term sum
    assert np.array_equal(H, H0)
    eps = np.finfo(float).eps
    for got, ref, mag in ((c, c0, ora.c_mag(x)), (G, G0, ora.G_mag(x))):
        assert np.max(np.abs(got - ref) / np.maximum(np.maximum(np.abs(ref), mag), 1e-300)) <= 16 * eps
    eng.close()
    base.close()
