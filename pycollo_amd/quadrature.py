"""Quadrature tables for the collocation transcription (host side, once per mesh iteration).

The engine consumes quadrature tables **as data** (SURVEY.md F5): for every section order ``n``
(number of nodes in a mesh section, both ends included) it needs

* ``points(n)``   -- n abscissae on [-1, 1]
* ``weights(n)``  -- n quadrature weights (Lobatto: sum 1; Radau: sum 2, reference quirk)
* ``A(n)``        -- the (n-1) x n integration matrix = rows 1.. of the Butcher array
* ``D(n)``        -- the (n-1) x n difference matrix [1 | -I]

Reference behaviour restated: ``pycollo/quadrature.py:116-187`` (Radau) and ``:189-261`` (Lobatto).
The reference obtains the interior Butcher rows from the order conditions
``sum_i w_i c_i^k a_ij = w_j (1 - c_j^(k+1)) / (k+1) - w_last w_j`` (quadrature.py:214-241) as one
block-expanded linear system; it is restated in that form (``_butcher``) because the system is
ill-conditioned at the higher orders and the tables must equal the reference's to the last digit.
Tables are checked against fixtures generated from the reference for every order 2..20
(tests/golden/quadrature_tables.npz).
"""
from __future__ import annotations

import functools

import numpy as np
from numpy.polynomial import legendre as _leg

LOBATTO = "lobatto"
RADAU = "radau"
SUPPORTED = (LOBATTO, RADAU)
ORDER_MIN = 2
ORDER_MAX = 20


def _unit_legendre(k: int) -> _leg.Legendre:
    coef = np.zeros(k + 1)
    coef[k] = 1.0
    return _leg.Legendre(coef)


def _lobatto_points_weights(n: int):
    # quadrature.py:190-203: interior points are the roots of P'_{n-1}; w = 1/(n(n-1)P_{n-1}(x)^2)
    P = _unit_legendre(n - 1)
    x = np.concatenate(([-1.0], P.deriv().roots(), [1.0]))
    w = 1.0 / (n * (n - 1) * P(x) ** 2)
    return x, w


def _radau_points_weights(n: int):
    # quadrature.py:117-134: roots of P_{n-2}+P_{n-1}; trailing placeholder point/weight of 0
    coef = np.zeros(n)
    coef[n - 2:] = 1.0
    x = np.concatenate((_leg.Legendre(coef).roots(), [0.0]))
    P = _unit_legendre(n - 2)
    w = np.zeros(n)
    w[0] = 2.0 / (n - 1) ** 2
    xi = x[1:-1]
    w[1:-1] = (1.0 - xi) / ((n - 1) ** 2 * P(xi) ** 2)
    return x, w


def _butcher(n: int, x: np.ndarray, w: np.ndarray, last_row: np.ndarray) -> np.ndarray:
    """Butcher array with zero first row, ``last_row`` last and order-condition interior rows.

    The interior rows come from the reference's own linear system, assembled entry by entry in its arrangement
    (quadrature.py:140-157 / :205-241): unknown a[i, j] sits at column ``i + j (n - 2)``, the condition of power k for
    column j at row ``j + k n``, and the whole ``n (n - 2)``-square system goes to one ``numpy.linalg.solve``.  The
    system is a row-permuted block-diagonal copy of one small Vandermonde-like matrix and could be solved as such,
    but it is badly conditioned from n ~ 10 on (the solution's trailing digits then depend on the elimination order),
    and the tables have to be the reference's, digit for digit -- so the same matrix meets the same LAPACK routine."""
    c = 0.5 * (x + 1.0)  # abscissae on [0, 1] (quadrature.py:85-91 with domain=[0, 1])
    B = np.zeros((n, n))
    B[-1, :] = last_row
    if n > 2:
        m = n - 2
        A = np.zeros((n * m, n * m))
        rhs = np.zeros(n * m)
        for k in range(m):
            for j in range(n):
                row = j + k * n
                for i in range(m):
                    A[row, i + j * m] = w[i + 1] * c[i + 1] ** k
                rhs[row] = (w[j] / (k + 1)) * (1 - c[j] ** (k + 1)) - w[-1] * w[j]
        B[1:-1, :] = np.linalg.solve(A, rhs).reshape(m, -1, order="F")
    return B


class QuadratureTables:
    """Lazily cached per-order tables for one scheme (mirrors ``Quadrature`` accessors)."""

    def __init__(self, method: str = LOBATTO):
        if method not in SUPPORTED:
            raise ValueError(f"quadrature method {method!r} is not supported; use one of {SUPPORTED}")
        self.method = method
        self._cache: dict[int, dict[str, np.ndarray]] = {}

    def _get(self, n: int) -> dict[str, np.ndarray]:
        n = int(n)
        if not (ORDER_MIN <= n <= ORDER_MAX):
            raise ValueError(f"section order {n} outside [{ORDER_MIN}, {ORDER_MAX}]")
        tab = self._cache.get(n)
        if tab is None:
            if self.method == LOBATTO:
                x, w = _lobatto_points_weights(n)
                B = _butcher(n, x, w, w)
            else:
                x, w = _radau_points_weights(n)
                B = _butcher(n, x, w, w / 2.0)
            D = np.hstack([np.ones((n - 1, 1)), -np.eye(n - 1)])
            tab = {"points": x, "weights": w, "butcher": B, "A": np.ascontiguousarray(B[1:, :]), "D": D}
            self._cache[n] = tab
        return tab

    def points(self, n, domain=None):
        x = self._get(n)["points"]
        if domain is not None:
            return 0.5 * (domain[1] - domain[0]) * x + 0.5 * (domain[0] + domain[1])
        return x

    def weights(self, n):
        return self._get(n)["weights"]

    def A(self, n):
        return self._get(n)["A"]

    def D(self, n):
        return self._get(n)["D"]

    def butcher(self, n):
        return self._get(n)["butcher"]


@functools.lru_cache(maxsize=None)
def tables(method: str = LOBATTO) -> QuadratureTables:
    return QuadratureTables(method)
