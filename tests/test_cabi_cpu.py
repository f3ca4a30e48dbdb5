"""CPU-side checks of the C ABI: the library loads, exports every symbol include/pycollo_amd.h declares,
builds structures bit-identical to the oracle's, and refuses to compute without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from oracle.ref_numpy import OracleNlp
from pycollo_amd import problems
from pycollo_amd.quadrature import QuadratureTables


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "pycollo_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pc_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(built):
    lib = ctypes.CDLL(os.path.join(ROOT, "pycollo_amd", "libpycollo_amd.so"))
    names = _declared_symbols()
    assert len(names) >= 18
    for name in names:
        assert hasattr(lib, name), f"{name} is declared in include/pycollo_amd.h but not exported"


CASES = [("brachistochrone", {}), ("hypersensitive", dict(K=2000, order=6)), ("cart_pole", dict(K=50, order=4)),
         ("shuttle", dict(K=30, order=5)), ("double_pendulum", {}), ("two_phase_transfer", {}),
         ("delta_iii", dict(K=7, order=4)), ("time_coupled_transfer", {}), ("time_coupled_transfer", dict(K=40, order=6))]


@pytest.mark.parametrize("name,kw", CASES)
def test_structures_bit_exact_vs_oracle(built, name, kw):
    """CSR index arrays (G and lower-triangular H) from the host pattern builder == the oracle's."""
    from pycollo_amd.engine import NlpEngine
    prob = problems.REGISTRY[name](**kw)
    eng = NlpEngine(prob, device=None)
    ora = OracleNlp(prob, QuadratureTables("lobatto"))
    assert (eng.num_x, eng.num_c) == (ora.num_x, ora.num_c)
    for got, ref in ((eng.evaluate_G_structure(), ora.G_structure()), (eng.evaluate_H_structure(), ora.H_structure())):
        assert got[0].dtype == np.int32 and got[1].dtype == np.int32
        np.testing.assert_array_equal(got[0], ref[0])
        np.testing.assert_array_equal(got[1], ref[1])
    assert eng.evaluate_G_num_nonzero() == len(ora.G_structure()[0])
    # CCS permutation (CasADi order, backend.py:1754-1761): columns ascending, rows ascending inside
    r, c = eng.evaluate_G_structure()
    p = eng.csr_to_ccs_permutation()
    key = c[p].astype(np.int64) * eng.num_c + r[p]
    assert np.all(np.diff(key) > 0)


def test_survey_sizes_config2_config3(built):
    """num_x / num_c / nnz for configs 2 and 3."""
    from pycollo_amd.engine import NlpEngine
    e = NlpEngine(problems.hypersensitive(K=2000, order=6), device=None)
    assert (e.num_x, e.num_c, e.nnz_jac, e.nnz_hess) == (20003, 10001, 140003, 20002)
    assert e.info["algorithmic_bytes"] == 1600080
    e = NlpEngine(problems.cart_pole(K=5000, order=4), device=None)
    assert (e.num_x, e.num_c, e.nnz_jac, e.nnz_hess) == (75006, 60001, 585002, 75005)


def test_ragged_mesh_structure(built):
    from pycollo_amd.engine import NlpEngine
    prob = problems.two_phase_transfer()
    prob.phases[0].mesh.number_mesh_sections = 5
    prob.phases[0].mesh.mesh_section_sizes = [0.1, 0.25, 0.05, 0.3, 0.3]
    prob.phases[0].mesh.number_mesh_section_nodes = [4, 7, 2, 5, 10]
    eng = NlpEngine(prob, device=None)
    ora = OracleNlp(prob, QuadratureTables("lobatto"))
    for got, ref in ((eng.evaluate_G_structure(), ora.G_structure()), (eng.evaluate_H_structure(), ora.H_structure())):
        np.testing.assert_array_equal(got[0], ref[0])
        np.testing.assert_array_equal(got[1], ref[1])


def test_no_cpu_fallback(built):
    """A structure-only handle must refuse to evaluate: the product has no CPU path."""
    from pycollo_amd.engine import NlpEngine
    eng = NlpEngine(problems.brachistochrone(), device=None)
    x = np.zeros(eng.num_x)
    for call in (lambda: eng.evaluate_c(x), lambda: eng.evaluate_G_nonzeros(x), lambda: eng.evaluate_J(x),
                 lambda: eng.evaluate_H_nonzeros(x, 1.0, np.zeros(eng.num_c)),
                 lambda: eng.evaluate_all(x, 1.0, np.zeros(eng.num_c))):
        with pytest.raises(RuntimeError, match="no CPU fallback|structure only"):
            call()


def test_bad_descriptors_are_rejected(built):
    from pycollo_amd.engine import NlpEngine
    with pytest.raises(RuntimeError, match="threads_per_block"):
        NlpEngine(problems.brachistochrone(), device=None, threads_per_block=96)
    eng = NlpEngine(problems.brachistochrone(), device=None)
    with pytest.raises(ValueError):
        eng.set_scaling(np.ones(3), np.zeros(3), np.ones(3))
    with pytest.raises(ValueError):
        eng.evaluate_c(np.zeros(eng.num_x + 1))


def test_cyipopt_surface_names(built):
    """PycolloGpuProblem exposes exactly the method names of IPOPTProblem (pycollo/nlp.py:47-76)."""
    from pycollo_amd.engine import NlpEngine, PycolloGpuProblem
    p = PycolloGpuProblem(NlpEngine(problems.brachistochrone(), device=None))
    for name in ("objective", "gradient", "constraints", "jacobian", "jacobianstructure", "hessian",
                 "hessianstructure", "intermediate"):
        assert callable(getattr(p, name))
    assert (p.n, p.m) == (125, 90)
    r, c = p.jacobianstructure()
    assert len(r) == len(c) == 870
    r, c = p.hessianstructure()
    assert np.all(r >= c)            # lower triangle
