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


def test_merge_runs_of_over_resolved_sections():
    """mesh_refinement.py:339-347,354-372: MERGE_TOLERANCE_FACTOR = 0 zeroes the threshold, it does not disable the
    branch -- sections whose predicted order P + n is negative are merged (:252-285).  Expected values worked from the
    reference's formulas by hand:
    n = 4, tol = 1e-7, e = 1e-13: P = ceil(log(1e-6) / log 4) = -9 -> -9 + ceil(log 10) = -6, predicted -2 < 0;
    merge ratio 4 / (4 + 6) = 0.4 per section, two neighbours -> ceil(0.8) = 1 section of their joint width."""
    from pycollo_amd.refinement import next_phase_mesh
    sizes, nodes, done = next_phase_mesh(np.full(4, 0.25), np.full(4, 4), [1e-5, 1e-13, 1e-13, 1e-6])
    assert not done
    assert np.array_equal(nodes, [8, 4, 6])                     # +4 nodes | merged pair at n_min | +2 nodes
    np.testing.assert_allclose(sizes, [0.25, 0.5, 0.25])
    # a run that needs two sections (order 6, uneven widths and errors): P = [-7, -8, -7], ratios 6/11, 6/12, 6/11
    # -> ceil(1.59) = 2; knots at [1/6, 1/2, 1] carry the densities [0.16176471, 0.51470588, 1], the new knots are the
    # density function evaluated at [0.5, 1] -> widths 0.6 * [0.51470588, 0.48529412]; then +3 nodes; then a section
    # predicted at 4 + 9 = 13 >= 10 nodes, cut into ceil(13 / 4) = 4
    sizes, nodes, done = next_phase_mesh([0.1, 0.2, 0.3, 0.25, 0.15], [6, 6, 6, 6, 4],
                                         [9.2e-16, 1.5e-16, 9.2e-16, 50e-7, 1e-2])
    assert np.array_equal(nodes, [4, 4, 9, 4, 4, 4, 4])
    np.testing.assert_allclose(sizes, [0.6 * 0.5147058823529411, 0.6 * 0.4852941176470589, 0.25] + [0.0375] * 4, rtol=1e-12)
    # a merge run at the end of the mesh and one at the start
    sizes, nodes, _ = next_phase_mesh(np.full(4, 0.25), np.full(4, 4), [1e-13, 1e-13, 1e-5, 1e-13])
    assert np.array_equal(nodes, [4, 8, 4])
    np.testing.assert_allclose(sizes, [0.5, 0.25, 0.25])
