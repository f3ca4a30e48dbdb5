"""Quadrature / mesh tables against fixtures generated from the reference (tests/golden/make_golden.py)
and against the reference's own unit-test known answers (tests/unit/test_quadrature.py:48-55)."""
import numpy as np
import pytest

from pycollo_amd.mesh import build_phase_mesh, uniform_phase_mesh
from pycollo_amd.quadrature import QuadratureTables


@pytest.mark.parametrize("method", ["lobatto", "radau"])
def test_quadrature_tables_match_reference(method, golden_quadrature):
    """Every order the reference allows (2..20, quadrature.py:36-37).  The reference solves an ill-conditioned system
    for the Butcher rows (quadrature.py:209-241; at n = 20 its rows miss their exact sums by 1e-5): the tables only
    agree because the product restates that very system and hands it to the same solver -- bit for bit on the machine
    that generated the fixtures, and within the conditioning's reach of a last-digit difference in LAPACK elsewhere."""
    q = QuadratureTables(method)
    for n in range(2, 21):
        np.testing.assert_allclose(q.points(n), golden_quadrature[f"{method}_{n}_points"], rtol=0, atol=1e-14)
        np.testing.assert_allclose(q.weights(n), golden_quadrature[f"{method}_{n}_weights"], rtol=1e-13, atol=1e-15)
        tol = 1e-13 if n <= 8 else (1e-11 if n <= 12 else 1e-6)
        np.testing.assert_allclose(q.A(n), golden_quadrature[f"{method}_{n}_A"], rtol=0, atol=tol)
        np.testing.assert_array_equal(q.D(n), golden_quadrature[f"{method}_{n}_D"])


def test_lobatto_weights_known_answers():
    # tests/unit/test_quadrature.py:48-55
    q = QuadratureTables("lobatto")
    np.testing.assert_allclose(q.weights(2), [0.5, 0.5])
    np.testing.assert_allclose(q.weights(3), [1 / 6, 2 / 3, 1 / 6])


def test_weight_sums_keep_reference_quirk():
    # Lobatto weights sum to 1, Radau's to 2
    assert abs(QuadratureTables("lobatto").weights(5).sum() - 1.0) < 1e-14
    assert abs(QuadratureTables("radau").weights(5).sum() - 2.0) < 1e-13


@pytest.mark.parametrize("method", ["lobatto", "radau"])
@pytest.mark.parametrize("case", ["k1n2", "k3n4", "k10n4", "k10n6", "ragged"])
def test_mesh_tables_match_reference(method, case, golden_mesh):
    key = f"{method}_{case}"
    pm = build_phase_mesh(QuadratureTables(method), golden_mesh[key + "_sizes"], golden_mesh[key + "_nodes"])
    assert pm.N == int(golden_mesh[key + "_N"])
    np.testing.assert_array_equal(pm.s, golden_mesh[key + "_bounds"])
    np.testing.assert_allclose(pm.tau, golden_mesh[key + "_tau"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(pm.h, golden_mesh[key + "_hK"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(pm.w, golden_mesh[key + "_W"], rtol=0, atol=1e-15)


def test_oracle_mesh_matrices_match_reference(golden_mesh, golden_quadrature):
    """The oracle's sparse difference / integration matrices equal mesh.sA_matrix / mesh.sI_matrix."""
    import os
    from oracle.ref_numpy import GoldenTables, OracleMesh
    from conftest import GOLDEN
    tab = GoldenTables(os.path.join(GOLDEN, "quadrature_tables.npz"))
    for case in ("k3n4", "k10n6", "ragged"):
        key = f"lobatto_{case}"
        om = OracleMesh(tab, golden_mesh[key + "_sizes"], golden_mesh[key + "_nodes"])
        for nm, mat in (("sI", om.I_mat), ("sA", om.A_mat)):
            mat = mat.tocsr(); mat.sort_indices()
            np.testing.assert_array_equal(mat.indptr, golden_mesh[f"{key}_{nm}_indptr"])
            np.testing.assert_array_equal(mat.indices, golden_mesh[f"{key}_{nm}_indices"])
            np.testing.assert_allclose(mat.data, golden_mesh[f"{key}_{nm}_data"], rtol=0, atol=1e-15)
        np.testing.assert_allclose(om.w, golden_mesh[key + "_W"], rtol=0, atol=1e-15)


def test_uniform_mesh_helper():
    pm = uniform_phase_mesh(QuadratureTables("lobatto"), 7, 5)
    assert pm.N == 29 and pm.K == 7
    assert abs(pm.h.sum() - 2.0) < 1e-14 and abs(pm.w.sum() - 2.0) < 1e-14


def test_mesh_rejects_bad_input():
    q = QuadratureTables("lobatto")
    with pytest.raises(ValueError):
        build_phase_mesh(q, [0.5, 0.5], [4, 1])
    with pytest.raises(ValueError):
        QuadratureTables("gauss")  # unsupported in the reference too (quadrature.py:34-35)
    with pytest.raises(ValueError):
        q.A(21)
