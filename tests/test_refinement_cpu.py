"""Host side of row N2: ph tables and the next-mesh rules (no GPU)."""
import numpy as np

from pycollo_amd.quadrature import QuadratureTables
from pycollo_amd.refinement import ph_tables


def test_ph_tables_equal_the_reference_polynomial_fits():
    """B / E reproduce Legendre.fit(...).integ(k=y0) and Polynomial.fit(...) of solution_abc.py:70-100."""
    q = QuadratureTables("lobatto")
    for n in (3, 4, 6, 8):
        B, E, A = ph_tables(q, n)
        assert B.shape == E.shape == (n - 1, n) and A.shape == (n, n + 1)
        c = 0.5 * (q.points(n) + 1)
        cp = 0.5 * (q.points(n + 1)[1:-1] + 1)
        rng = np.random.default_rng(n)
        f, u = rng.normal(size=n), rng.normal(size=n)
        yref = np.polynomial.Legendre.fit(c, f, deg=n - 1, window=[0, 1]).integ(k=0.3)(cp)
        uref = np.polynomial.Polynomial.fit(c, u, deg=n - 1, window=[0, 1])(cp)
        np.testing.assert_allclose(0.3 + B @ f, yref, atol=1e-11)
        np.testing.assert_allclose(E @ u, uref, atol=1e-10)
        np.testing.assert_allclose(E.sum(axis=1), 1.0, atol=1e-12)       # partition of unity
        np.testing.assert_allclose(B.sum(axis=1), cp, atol=1e-12)        # integral of 1 up to c


def test_next_mesh_rules():
    from pycollo_amd.refinement import next_phase_mesh
    sizes, nodes, done = next_phase_mesh(np.full(4, 0.25), np.full(4, 4), [1e-9, 1e-8, 1e-10, 5e-8])
    assert done and np.array_equal(nodes, [4, 4, 4, 4])
    sizes, nodes, done = next_phase_mesh(np.full(4, 0.25), np.full(4, 4), [1e-9, 1e-5, 1e-3, 1e-1])
    assert not done
    assert np.array_equal(nodes, [4, 8, 4, 4, 4, 4, 4, 4, 4])          # +4 nodes; 11 -> 3 sections; 14 -> 4 sections
    np.testing.assert_allclose(sizes, [0.25, 0.25] + [0.25 / 3] * 3 + [0.0625] * 4)
