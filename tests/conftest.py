"""pytest configuration: `gpu` marker, import path, shared helpers."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_quadrature():
    return np.load(os.path.join(GOLDEN, "quadrature_tables.npz"))


@pytest.fixture(scope="session")
def golden_mesh():
    return np.load(os.path.join(GOLDEN, "mesh_tables.npz"))


@pytest.fixture(scope="session")
def known_answers():
    return np.load(os.path.join(GOLDEN, "known_answers.npz"))


@pytest.fixture(scope="session")
def built():
    """Build the C-ABI library + code objects once per session (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as entry
    entry.build()
    return entry


def rel_err(got, ref):
    """max |got - ref| relative to the largest reference magnitude of the block (inf == inf allowed)."""
    got, ref = np.asarray(got, float), np.asarray(ref, float)
    if ref.size == 0:
        return 0.0
    both_inf = np.isinf(got) & np.isinf(ref) & (np.sign(got) == np.sign(ref))
    both_nan = np.isnan(got) & np.isnan(ref)
    d = np.where(both_inf | both_nan, 0.0, np.abs(got - ref))
    fin = ref[np.isfinite(ref)]
    scale = max(np.max(np.abs(fin)) if fin.size else 1.0, 1e-300)
    return float(np.max(d) / scale)
