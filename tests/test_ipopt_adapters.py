"""The IPOPT-typed callbacks (include/pycollo_amd.h, pc_ipopt_eval_*) driven from C through function pointers of
IpStdCInterface.h's types, in IPOPT's call order (tests/c/ipopt_protocol.c).  Reference call site they stand in for:
pycollo/nlp.py:84-115 (ipopt.problem(n, m, problem_obj, lb, ub, cl, cu) via cyipopt)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, assert_matches_oracle, golden_tables
from pycollo_amd import problems

BUILD = os.path.join(ROOT, "tests", "_build")


class _Callbacks(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("eval_f", "eval_g", "eval_grad_f", "eval_jac_g", "eval_h")]


@pytest.fixture(scope="module")
def harness(built):
    os.makedirs(BUILD, exist_ok=True)
    so = os.path.join(BUILD, "ipopt_protocol.so")
    src = os.path.join(ROOT, "tests", "c", "ipopt_protocol.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        tmp = so + f".tmp{os.getpid()}"
        subprocess.run(["gcc", "-O1", "-Wall", "-Werror", "-fPIC", "-shared", "-o", tmp, src], check=True)
        os.replace(tmp, so)
    lib = C.CDLL(so)
    vp, ci = C.c_void_p, C.c_int
    lib.drive_structure.argtypes = [vp, ci, ci, ci, ci, vp, vp, vp, vp, vp]
    lib.drive_point.argtypes = [vp, ci, ci, ci, ci, vp, C.c_double] + [vp] * 7
    lib.drive_wrong_sizes.argtypes = [vp, ci, ci, ci, vp, vp, vp]
    from pycollo_amd.engine import load_library
    pc = load_library()
    cb = _Callbacks(*[C.cast(getattr(pc, "pc_ipopt_" + k), C.c_void_p) for k in
                      ("eval_f", "eval_g", "eval_grad_f", "eval_jac_g", "eval_h")])
    return lib, cb


def _i32(n):
    return np.empty(n, dtype=np.int32)


def test_structure_query_through_ipopt_typed_pointers(harness):
    """values == NULL => structure, as IPOPT asks once at start-up; no GPU needed (structure-only handle)."""
    from pycollo_amd.engine import NlpEngine
    lib, cb = harness
    eng = NlpEngine(problems.two_phase_transfer(K=9, order=4), device=None)
    n, m, nj, nh = eng.num_x, eng.num_c, eng.nnz_jac, eng.nnz_hess
    jr, jc, hr, hc = _i32(nj), _i32(nj), _i32(nh), _i32(nh)
    ok = lib.drive_structure(C.byref(cb), n, m, nj, nh, jr.ctypes.data, jc.ctypes.data, hr.ctypes.data, hc.ctypes.data,
                             eng._h)
    assert ok == 1
    for got, ref in ((jr, eng.evaluate_G_structure()[0]), (jc, eng.evaluate_G_structure()[1]),
                     (hr, eng.evaluate_H_structure()[0]), (hc, eng.evaluate_H_structure()[1])):
        np.testing.assert_array_equal(got, ref)
    assert np.all(hc <= hr)          # lower triangle, 0-based
    x, g = np.zeros(n), np.zeros(m + 1)
    assert lib.drive_wrong_sizes(C.byref(cb), n, m, nj, x.ctypes.data, g.ctypes.data, eng._h) == 1
    # evaluation on a structure-only handle fails loudly (returns 0, IPOPT's "evaluation error"), it does not fall back
    f = C.c_double()
    fn = C.CFUNCTYPE(C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p)(cb.eval_f)
    assert fn(n, x.ctypes.data, 1, C.addressof(f), eng._h.value) == 0
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,kw", [("double_pendulum", {}), ("hypersensitive", dict(K=500, order=6))])
def test_ipopt_call_sequence_on_gpu(harness, name, kw):
    """f(new_x=1), grad_f(0), g(0), jac_g(0), h(0) at several points, against the oracle."""
    from oracle.ref_numpy import OracleNlp
    from pycollo_amd.engine import NlpEngine
    from pycollo_amd.quadrature import QuadratureTables
    lib, cb = harness
    prob = problems.REGISTRY[name](**kw)
    eng = NlpEngine(prob, device=0)
    ora = OracleNlp(prob, golden_tables("lobatto"), V_ocp=eng.V_ocp, r_ocp=eng.r_ocp, W_ocp=eng.W_ocp, w_J=1.0)
    n, m, nj, nh = eng.num_x, eng.num_c, eng.nnz_jac, eng.nnz_hess
    rng = np.random.default_rng(3)
    for _ in range(3):
        x, lam = rng.uniform(-0.4, 0.4, n), rng.normal(size=m)
        f = C.c_double()
        grad, g, jac, hess = np.empty(n), np.empty(m), np.empty(nj), np.empty(nh)
        ok = lib.drive_point(C.byref(cb), n, m, nj, nh, x.ctypes.data, 0.7, lam.ctypes.data, C.addressof(f),
                             grad.ctypes.data, g.ctypes.data, jac.ctypes.data, hess.ctypes.data, eng._h)
        assert ok == 1
        assert abs(f.value - ora.J(x)) <= 1e-10 * max(1.0, abs(ora.J(x)))
        assert_matches_oracle(ora, x, g=grad, c=g, G=jac, H=hess, sigma=0.7, lam=lam)
    eng.close()
