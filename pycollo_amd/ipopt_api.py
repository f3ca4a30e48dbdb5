"""The legacy cyipopt interface over the stand-in solver: ``problem(n, m, problem_obj, lb, ub, cl, cu)``,
``addOption``, ``setProblemScaling``, ``solve(x0) -> (x, info)``.

The reference reaches IPOPT through exactly these four calls (pycollo/nlp.py:84-115) and IPOPT is not in this image,
so the lines of ``initialise_nlp_backend`` can be run verbatim against this module::

    from pycollo_amd import ipopt_api as ipopt
    nlp = ipopt.problem(n=..., m=..., problem_obj=PycolloGpuProblem(engine), lb=..., ub=..., cl=..., cu=...)
    nlp.addOption('mu_strategy', 'adaptive'); nlp.addOption('tol', 1e-8); nlp.addOption('max_iter', 2000)
    nlp.addOption('nlp_scaling_method', 'user-scaling'); nlp.setProblemScaling(J_scale, x_scales, c_scales)
    x, info = nlp.solve(x0)

It is an interface adapter, not IPOPT: the algorithm behind it is ``pycollo_amd.ipm`` (same family -- line-search filter
interior point -- no claim of equal iteration counts).  With a :class:`pycollo_amd.engine.PycolloGpuProblem` and no user
scaling the KKT systems are factorised on the GPU (``linear_solver`` option "gpu", the default there) -- the role of
IPOPT's ``linear_solver`` option (pycollo/backend.py:1703-1711).  Where cyipopt exists, ``import ipopt`` instead.
"""
from __future__ import annotations

import numpy as np

from .ipm import GpuInteriorPointSolver, InteriorPointSolver

# IPOPT's ApplicationReturnStatus values for the outcomes the stand-in distinguishes
_STATUS = {"optimal": (0, b"Algorithm terminated successfully at a locally optimal point, satisfying the convergence tolerances (can be specified by options)."),
           "acceptable": (1, b"Algorithm stopped at a point that was converged, not to \"desired\" tolerances, but to \"acceptable\" tolerances (see the acceptable-... options)."),
           "max_iter": (-1, b"Maximum number of iterations exceeded (can be specified by an option)."),
           "restoration_failed": (-2, b"Restoration phase failed, algorithm doesn't know how to proceed."),
           "step_failed": (-3, b"An unrecoverable error occurred while IPOPT tried to compute the search direction.")}


class _UserScaled:
    """IPOPT's nlp_scaling_method = user-scaling: the solver sees f~ = s_f f, x~ = s_x x, g~ = s_g g."""

    def __init__(self, inner, so, sx, sg):
        self.p, self.so, self.sx, self.sg = inner, float(so), np.asarray(sx, float), np.asarray(sg, float)
        self.jr, self.jc = (np.asarray(a, np.int64) for a in inner.jacobianstructure())
        self.hr, self.hc = (np.asarray(a, np.int64) for a in inner.hessianstructure())

    def objective(self, xt):
        return self.so * float(self.p.objective(xt / self.sx))

    def gradient(self, xt):
        return self.so * np.asarray(self.p.gradient(xt / self.sx), float) / self.sx

    def constraints(self, xt):
        return self.sg * np.asarray(self.p.constraints(xt / self.sx), float)

    def jacobianstructure(self):
        return self.jr, self.jc

    def jacobian(self, xt):
        return np.asarray(self.p.jacobian(xt / self.sx), float) * self.sg[self.jr] / self.sx[self.jc]

    def hessianstructure(self):
        return self.hr, self.hc

    def hessian(self, xt, lam, obj_factor):
        h = np.asarray(self.p.hessian(xt / self.sx, self.sg * np.asarray(lam, float), obj_factor * self.so), float)
        return h / (self.sx[self.hr] * self.sx[self.hc])


class problem:                                   # (cyipopt's class name)
    def __init__(self, n, m, problem_obj=None, lb=None, ub=None, cl=None, cu=None):
        if problem_obj is None:
            raise ValueError("problem_obj is required")
        self.n, self.m, self.problem_obj = int(n), int(m), problem_obj
        big = 2.0e19                             # IPOPT's nlp_*_bound_inf
        self.lb = np.full(self.n, -big) if lb is None else np.asarray(lb, float)
        self.ub = np.full(self.n, big) if ub is None else np.asarray(ub, float)
        self.cl = np.full(self.m, -big) if cl is None else np.asarray(cl, float)
        self.cu = np.full(self.m, big) if cu is None else np.asarray(cu, float)
        self.options: dict = {}
        self.scaling = None

    def addOption(self, keyword, value):
        self.options[keyword.decode() if isinstance(keyword, bytes) else str(keyword)] = value

    add_option = addOption

    def setProblemScaling(self, obj_scaling=None, x_scaling=None, g_scaling=None):
        so = 1.0 if obj_scaling is None else float(obj_scaling)
        sx = np.ones(self.n) if x_scaling is None else np.asarray(x_scaling, float)
        sg = np.ones(self.m) if g_scaling is None else np.asarray(g_scaling, float)
        if len(sx) != self.n or len(sg) != self.m or np.any(sx <= 0) or np.any(sg <= 0) or so <= 0:
            raise ValueError("scaling factors must be positive and of length n / m")
        self.scaling = (so, sx, sg)

    set_problem_scaling = setProblemScaling

    def close(self):
        pass

    def solve(self, x, lagrange=None, zl=None, zu=None):
        """Returns ``(x, info)`` with cyipopt's ``info`` keys."""
        o = self.options
        method = str(o.get("nlp_scaling_method", "gradient-based"))
        user = method == "user-scaling" and self.scaling is not None
        kw = dict(tol=float(o.get("tol", 1e-8)), max_iter=int(o.get("max_iter", 3000)),
                  verbose=1 if int(o.get("print_level", 0)) >= 5 else 0,
                  warm_start=str(o.get("warm_start_init_point", "no")) == "yes",
                  gradient_scaling=(method == "gradient-based"))
        if "acceptable_tol" in o:
            kw["acceptable_tol"] = float(o["acceptable_tol"])
        x0 = np.asarray(x, float)
        if user:
            so, sx, sg = self.scaling
            inner = _UserScaled(self.problem_obj, so, sx, sg)
            # IPOPT tests "infinite" (|b| >= 1e19, nlp_lower/upper_bound_inf) on the UNSCALED bounds: a scale below
            # 0.5 must not turn an absent bound into a finite one (-2e19 * 0.4 = -8e18 would get barrier terms)
            def scaled(b, sc):
                b = np.asarray(b, float)
                return np.where(np.abs(b) >= 1e19, np.sign(b) * 2e19, b * sc)
            solver = InteriorPointSolver(inner, self.n, self.m, scaled(self.lb, sx), scaled(self.ub, sx),
                                         scaled(self.cl, sg), scaled(self.cu, sg), **kw)
            res = solver.solve(x0 * sx)
            xs = res.x / sx
            mult_g, zl_, zu_ = res.lam * sg / so, res.zl * sx / so, res.zu * sx / so
        else:
            gpu = hasattr(self.problem_obj, "engine") and str(o.get("linear_solver", "gpu")) == "gpu"
            cls = GpuInteriorPointSolver if gpu else InteriorPointSolver
            res = cls(self.problem_obj, self.n, self.m, self.lb, self.ub, self.cl, self.cu, **kw).solve(x0)
            xs, mult_g, zl_, zu_ = res.x, res.lam, res.zl, res.zu
        code, msg = _STATUS.get(res.status, (-199, res.status.encode()))
        info = {"x": xs, "g": np.asarray(self.problem_obj.constraints(xs), float), "obj_val": float(self.problem_obj.objective(xs)),
                "mult_g": mult_g, "mult_x_L": zl_, "mult_x_U": zu_, "status": code, "status_msg": msg}
        self.last_result = res
        return xs, info
