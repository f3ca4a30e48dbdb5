"""GPU tests of the host-pointer callbacks (what IPOPT / cyipopt see): IPOPT's new_x protocol, the packed staging
blocks, the zero-copy hand-over and the three data-movement modes.  Reference surface: pycollo/nlp.py:47-63
(objective / gradient / constraints / jacobian / hessian on host numpy arrays), IpStdCInterface.h's new_x flag."""
import numpy as np
import pytest

from conftest import assert_matches_oracle, golden_tables, vec_err
from oracle.ref_numpy import OracleNlp
from pycollo_amd import problems
from pycollo_amd.quadrature import QuadratureTables

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _engine(prob, **kw):
    from pycollo_amd.engine import NlpEngine
    return NlpEngine(prob, device=0, **kw)


@pytest.mark.parametrize("name,kw", [("double_pendulum", {}), ("hypersensitive", dict(K=300, order=6))])
def test_new_x_protocol(built, name, kw):
    """The first callback at a new point carries new_x = 1 -- usually eval_f or eval_grad_f -- and the companion calls
    carry 0: g / jac_g must then be those of the new point, never a cached c~/G~ of the previous one."""
    prob = problems.REGISTRY[name](**kw)
    eng = _engine(prob)
    ora = OracleNlp(prob, golden_tables("lobatto"), V_ocp=eng.V_ocp, r_ocp=eng.r_ocp, W_ocp=eng.W_ocp, w_J=1.0)
    rng = np.random.default_rng(5)
    x0, x1, x2, x3 = (rng.uniform(-0.4, 0.4, eng.num_x) for _ in range(4))
    lam = rng.normal(size=eng.num_c)
    c0 = eng.evaluate_c(x0)                        # caches c~, G~ at x0
    assert_matches_oracle(ora, x0, c=c0)
    J1 = eng.evaluate_J(x1, new_x=True)            # new point announced to eval_f
    assert abs(J1 - ora.J(x1)) <= TOL * max(1.0, abs(J1))
    assert_matches_oracle(ora, x1, c=eng.evaluate_c(x1, new_x=False), G=eng.evaluate_G_nonzeros(x1, new_x=False),
                          g=eng.evaluate_g(x1, new_x=False))
    g2 = eng.evaluate_g(x2, new_x=True)            # ... to eval_grad_f
    assert_matches_oracle(ora, x2, g=g2, G=eng.evaluate_G_nonzeros(x2, new_x=False), c=eng.evaluate_c(x2, new_x=False))
    H3 = eng.evaluate_H_nonzeros(x3, 0.7, lam, new_x=True)   # ... to eval_h
    assert_matches_oracle(ora, x3, H=H3, sigma=0.7, lam=lam, c=eng.evaluate_c(x3, new_x=False))
    assert abs(eng.evaluate_J(x3, new_x=False) - ora.J(x3)) <= TOL * max(1.0, abs(ora.J(x3)))
    # a change of scaling invalidates the cache even if the caller says the point is old
    W = eng.W_ocp * 2.0
    eng.set_scaling(eng.V_ocp, eng.r_ocp, W, 1.0)
    assert_matches_oracle(ora, x3, c=eng.evaluate_c(x3, new_x=False), c_scale=2.0)
    eng.close()


def test_cyipopt_object_recovers_new_x(built):
    """PycolloGpuProblem (nlp.py:36-76 method names): cyipopt passes no new_x, the object compares points."""
    from pycollo_amd.engine import PycolloGpuProblem
    prob = problems.cart_pole(K=60, order=4)
    eng = _engine(prob)
    ora = OracleNlp(prob, golden_tables("lobatto"), V_ocp=eng.V_ocp, r_ocp=eng.r_ocp, W_ocp=eng.W_ocp, w_J=1.0)
    p = PycolloGpuProblem(eng)
    rng = np.random.default_rng(9)
    lam = rng.normal(size=eng.num_c)
    for _ in range(3):
        x = rng.uniform(-0.4, 0.4, eng.num_x)
        assert abs(p.objective(x) - ora.J(x)) <= TOL * max(1.0, abs(ora.J(x)))
        assert_matches_oracle(ora, x, g=p.gradient(x), c=p.constraints(x), G=p.jacobian(x), H=p.hessian(x, lam, 0.5),
                              sigma=0.5, lam=lam)
    eng.close()


@pytest.mark.parametrize("name,kw", [("hypersensitive", dict(K=2000, order=6)), ("delta_iii", dict(K=40, order=4)),
                                     ("double_pendulum", {})])
def test_host_modes_and_inplace_views_write_the_same_bits(built, name, kw):
    """DMA copies (mode 0), kernels reading / writing pinned host memory (modes 1-3) and the in-place views of the
    pinned blocks all return exactly what the copying call returns."""
    prob = problems.REGISTRY[name](**kw)
    eng = _engine(prob)
    rng = np.random.default_rng(2)
    lo, hi = (0.05, 0.3) if name == "delta_iii" else (-0.45, 0.45)
    x = rng.uniform(lo, hi, eng.num_x)
    lam = rng.normal(size=eng.num_c)
    ref = eng.evaluate_all(x, 0.8, lam)
    refJ, refg = eng.evaluate_J(x), eng.evaluate_g(x)
    hx, hl, hc, hG, hH = eng.host_buffers()
    for mode in (0, 1, 2, 3):
        eng.set_host_mode(mode)
        got = eng.evaluate_all(x, 0.8, lam)
        for a, b in zip(ref, got):
            np.testing.assert_array_equal(a, b)
        hx[:] = x
        hl[:] = lam
        hc[:] = np.nan
        hG[:] = np.nan
        hH[:] = np.nan
        c, G, H = eng.evaluate_all_inplace(0.8)
        assert c.ctypes.data == hc.ctypes.data
        for a, b in zip(ref, (c, G, H)):
            np.testing.assert_array_equal(a, b)
        # the separate callbacks in this mode
        assert eng.evaluate_J(x) == refJ
        np.testing.assert_array_equal(eng.evaluate_g(x, new_x=False), refg)
        np.testing.assert_array_equal(eng.evaluate_c(x, new_x=False), ref[0])
        np.testing.assert_array_equal(eng.evaluate_G_nonzeros(x, new_x=False), ref[1])
        np.testing.assert_array_equal(eng.evaluate_H_nonzeros(x, 0.8, lam, new_x=False), ref[2])
    eng.set_host_mode(0)
    # row by row; at Delta III's random point some entries exceed 1e154 and their squares overflow on both sides: the
    # row norm is then +inf on the device and here alike (vec_err requires the same non-finite value)
    with np.errstate(over="ignore"):
        ref_norms = np.sqrt(np.add.reduceat(ref[1] ** 2, _indptr(eng)[:-1]))
    assert vec_err(eng.G_row_norms(x), ref_norms, rtol=1e-13) <= 1.0
    eng.close()


def _indptr(eng):
    r, _ = eng.evaluate_G_structure()
    return np.concatenate([[0], np.cumsum(np.bincount(r, minlength=eng.num_c))])


@pytest.mark.parametrize("then", ["hessian", "row_norms"])
def test_prefetch_off_jacobian_after_other_calls(built, then):
    """pc_set_prefetch_jac(0), kernels not writing host memory: G~ stays on the device until pc_eval_jac_g asks for
    it.  A pc_eval_h (or a row-norm pass) in between drains the stream; the Jacobian requested afterwards at the same
    point (new_x = 0) must still be fetched -- not the pinned block's previous contents."""
    prob = problems.cart_pole(K=40, order=4)
    eng = _engine(prob)
    ora = OracleNlp(prob, golden_tables("lobatto"), V_ocp=eng.V_ocp, r_ocp=eng.r_ocp, W_ocp=eng.W_ocp, w_J=1.0)
    rng = np.random.default_rng(17)
    x0, x1 = rng.uniform(-0.4, 0.4, eng.num_x), rng.uniform(-0.4, 0.4, eng.num_x)
    lam = rng.normal(size=eng.num_c)
    eng.set_host_mode(0)
    eng.set_prefetch_jac(True)
    eng.evaluate_G_nonzeros(x0)            # the pinned block now holds G~(x0)
    eng.set_prefetch_jac(False)
    assert_matches_oracle(ora, x1, c=eng.evaluate_c(x1))          # new point: G~(x1) computed, left on the device
    if then == "hessian":
        assert_matches_oracle(ora, x1, H=eng.evaluate_H_nonzeros(x1, 0.9, lam, new_x=False), sigma=0.9, lam=lam)
        assert_matches_oracle(ora, x1, G=eng.evaluate_G_nonzeros(x1, new_x=False))
    else:
        eng.G_row_norms(x1)
        assert_matches_oracle(ora, x1, G=eng.evaluate_G_nonzeros(x1, new_x=False))
        assert_matches_oracle(ora, x0, c=eng.evaluate_c(x0), G=eng.evaluate_G_nonzeros(x0, new_x=False))
    eng.close()


def test_cyipopt_object_survives_direct_engine_calls(built):
    """PycolloGpuProblem recovers new_x from the ENGINE's record of its cached point: a direct engine call in between
    (evaluate_all at another point) must not make the next callback return that other point's values."""
    from pycollo_amd.engine import PycolloGpuProblem
    prob = problems.cart_pole(K=30, order=4)
    eng = _engine(prob)
    ora = OracleNlp(prob, golden_tables("lobatto"), V_ocp=eng.V_ocp, r_ocp=eng.r_ocp, W_ocp=eng.W_ocp, w_J=1.0)
    p = PycolloGpuProblem(eng)
    rng = np.random.default_rng(23)
    x, x2 = rng.uniform(-0.4, 0.4, eng.num_x), rng.uniform(-0.4, 0.4, eng.num_x)
    lam = rng.normal(size=eng.num_c)
    assert abs(p.objective(x) - ora.J(x)) <= TOL * max(1.0, abs(ora.J(x)))
    eng.evaluate_all(x2, 1.0, lam)                        # the library's cache now describes x2
    assert_matches_oracle(ora, x, c=p.constraints(x), G=p.jacobian(x), g=p.gradient(x))
    eng.evaluate_c(x2)                                    # protocol callback at another point, by hand
    assert_matches_oracle(ora, x, c=p.constraints(x), G=p.jacobian(x))
    eng.close()


def test_cyipopt_object_survives_device_api_calls(built):
    """The device-pointer entry points (pc_eval_all_device, pc_launch_*) write the handle's own J / grad J block:
    a callback that follows one of them at the point of the previous callback must be re-evaluated, not served from
    the cache (engine.cache_holds is cleared by every device-API wrapper, the library clears its own flag too)."""
    import torch
    from pycollo_amd.engine import PycolloGpuProblem
    prob = problems.cart_pole(K=30, order=4)
    eng = _engine(prob)
    ora = OracleNlp(prob, golden_tables("lobatto"), V_ocp=eng.V_ocp, r_ocp=eng.r_ocp, W_ocp=eng.W_ocp, w_J=1.0)
    p = PycolloGpuProblem(eng)
    rng = np.random.default_rng(29)
    x, x2 = rng.uniform(-0.4, 0.4, eng.num_x), rng.uniform(-0.4, 0.4, eng.num_x)
    lam = rng.normal(size=eng.num_c)
    dev = torch.device("cuda", 0)
    dx2, dl = torch.from_numpy(x2).to(dev), torch.from_numpy(lam).to(dev)
    dc = torch.empty(eng.num_c, dtype=torch.float64, device=dev)
    dG = torch.empty(eng.nnz_jac, dtype=torch.float64, device=dev)
    dH = torch.empty(eng.nnz_hess, dtype=torch.float64, device=dev)
    s = torch.cuda.Stream(device=dev)
    assert abs(p.objective(x) - ora.J(x)) <= TOL * max(1.0, abs(ora.J(x)))
    assert eng.cache_holds(x)
    eng.evaluate_all_device(dx2, 1.0, dl, dc, dG, dH, s.cuda_stream)     # overwrites the handle's f block with J(x2)
    s.synchronize()
    assert not eng.cache_holds(x)
    assert abs(p.objective(x) - ora.J(x)) <= TOL * max(1.0, abs(ora.J(x)))
    assert_matches_oracle(ora, x, c=p.constraints(x), g=p.gradient(x))
    # the library's own flag: a protocol call with new_x = 0 straight after a device-API launch re-evaluates as well
    eng.evaluate_J(x, new_x=True)
    step = eng.bind_device(dx2, dl, dc, dG, dH, s.cuda_stream)
    step(1.0)
    s.synchronize()
    assert abs(eng.evaluate_J(x, new_x=False) - ora.J(x)) <= TOL * max(1.0, abs(ora.J(x)))
    eng.close()
