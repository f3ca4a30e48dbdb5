"""Generate golden fixtures from the reference (runs ONLY in the build container).

The reference package cannot be imported as a whole here (casadi / pyproprop are not
installed).  Two of its numeric modules do load by file path:

* ``pycollo/mesh.py``        -- needs numpy/scipy only.
* ``pycollo/quadrature.py``  -- needs ``pyproprop.Options`` at import time, used for one
  module-level constant (``QUADRATURES``, quadrature.py:34-35).  A 10-line container class
  with no arithmetic is placed in ``sys.modules`` for that single name; every
  number written below is computed by the reference's own code.

Also copies the *data arrays* of the reference's unit-test data modules
(tests/unit/iteration_scaling_test_data_{brachistochrone,double_pendulum}.py, numpy only).

Outputs (committed): tests/golden/quadrature_tables.npz, mesh_tables.npz, known_answers.npz
The reference never travels to the GPU box; only these .npz files do.
"""
import importlib.util
import os
import sys
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class _Options:  # stand-in for pyproprop.Options: a bag of names, no arithmetic
    def __init__(self, options, default=None, unsupported=None, handles=None):
        self.options = tuple(options)
        self.default = default
        self.unsupported = unsupported
        self.handles = handles


def main():
    sys.modules.setdefault("pyproprop", types.SimpleNamespace(Options=_Options))
    quad_mod = _load("ref_quadrature", f"{REF}/pycollo/quadrature.py")
    mesh_mod = _load("ref_mesh", f"{REF}/pycollo/mesh.py")

    def backend(method, quad=None):
        settings = types.SimpleNamespace(collocation_points_min=2, collocation_points_max=20,
                                         quadrature_method=method)
        b = types.SimpleNamespace(ocp=types.SimpleNamespace(settings=settings))
        b.quadrature = quad
        return b

    # (i) quadrature tables, every order the reference allows (2..20, quadrature.py:36-37), lobatto + radau
    out = {}
    quads = {}
    for method in ("lobatto", "radau"):
        q = quad_mod.Quadrature(backend(method))
        quads[method] = q
        for n in range(2, 21):
            out[f"{method}_{n}_points"] = np.asarray(q.quadrature_point(n), dtype=float)
            out[f"{method}_{n}_weights"] = np.asarray(q.quadrature_weight(n), dtype=float)
            out[f"{method}_{n}_A"] = np.asarray(q.A_matrix(n), dtype=float)
            out[f"{method}_{n}_D"] = np.asarray(q.D_matrix(n), dtype=float)
    np.savez_compressed(f"{HERE}/quadrature_tables.npz", **out)

    # (ii) mesh tables
    cases = {
        "k1n2": ([1.0], [2]),
        "k3n4": ([1 / 3] * 3, [4] * 3),
        "k10n4": ([0.1] * 10, [4] * 10),
        "k10n6": ([0.1] * 10, [6] * 10),
        "ragged": ([0.1, 0.25, 0.05, 0.3, 0.3], [4, 7, 2, 5, 10]),
    }
    mout = {}
    for method in ("lobatto", "radau"):
        b = backend(method, quads[method])
        for name, (sizes, nodes) in cases.items():
            pm = types.SimpleNamespace(mesh_section_sizes=np.array(sizes, dtype=float),
                                       number_mesh_section_nodes=np.array(nodes, dtype=int),
                                       number_mesh_sections=len(nodes))
            m = mesh_mod.Mesh(b, [pm])
            key = f"{method}_{name}"
            mout[f"{key}_sizes"] = np.array(sizes, dtype=float)
            mout[f"{key}_nodes"] = np.array(nodes, dtype=np.int64)
            mout[f"{key}_tau"] = m.tau[0]
            mout[f"{key}_N"] = np.array(m.N[0])
            mout[f"{key}_bounds"] = np.asarray(m.mesh_index_boundaries[0], dtype=np.int64)
            mout[f"{key}_hK"] = m.h_K[0]
            mout[f"{key}_W"] = m.W_matrix[0]
            for nm, mat in (("sI", m.sI_matrix[0]), ("sA", m.sA_matrix[0])):
                mat = mat.tocsr()
                mat.sort_indices()
                mout[f"{key}_{nm}_indptr"] = mat.indptr.astype(np.int64)
                mout[f"{key}_{nm}_indices"] = mat.indices.astype(np.int64)
                mout[f"{key}_{nm}_data"] = mat.data.astype(float)
                mout[f"{key}_{nm}_shape"] = np.array(mat.shape, dtype=np.int64)
    np.savez_compressed(f"{HERE}/mesh_tables.npz", **mout)

    # (iii) unit-test data arrays (tests/unit/iteration_scaling_test_data_*.py)
    kout = {}
    for tag, fname in (("BR", "brachistochrone"), ("DP", "double_pendulum")):
        mod = _load(f"ref_data_{tag}", f"{REF}/tests/unit/iteration_scaling_test_data_{fname}.py")
        for nm in ("V", "R", "V_INV", "X", "X_TILDE"):
            kout[f"EXPECT_{nm}_{tag}"] = np.asarray(getattr(mod, f"EXPECT_{nm}_{tag}"), dtype=float)
    np.savez_compressed(f"{HERE}/known_answers.npz", **kout)
    print("wrote", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
