"""One mesh iteration: everything between a problem description and a ready-to-solve NLP.

SURVEY.md section 8f row N1 -- the direct caller of the hot path.  Restates ``Iteration.initialise``
(pycollo/iteration.py:69-79): interpolate the guess to the mesh (:86-194), lay out variables and
constraints (:196-342, done by ``NlpLayout``), scale the guess (:360-373, pycollo/scaling.py:172-174),
generate the NLP callbacks + objective / constraint scaling (:375-394, scaling.py:271-281) and the scaled
bounds (:396-453).  The O(N) pieces run on the GPU: guess interpolation (``pc_interp_linear``), the
Jacobian row norms behind the constraint scaling (``pc_row_norms_jac``); the reference does the first with
one scipy ``interp1d`` per variable and the second on a dense ``num_c x num_x`` array (scaling.py:394).

``solve_with_scipy`` drives the callbacks with ``scipy.optimize.minimize(method="trust-constr")`` -- a
stand-in for IPOPT (not installed here) that lets the reference's converged-objective known
answers act as end-to-end checks on small meshes.
"""
from __future__ import annotations

import os

import numpy as np

from .engine import NlpEngine, interp_linear
from .mesh import build_phase_mesh
from .model import compile_model
from .quadrature import QuadratureTables
from .scaling import constraint_scaling, objective_scaling, scaling_from_previous


class MeshIteration:
    def __init__(self, problem, *, device: int = 0, meshes=None, prev=None, threads_per_block: int = 0, number: int = 1,
                 update_scaling: bool = False, scaling_weight: float = 0.8, history=None):
        """``prev`` = (tau per phase, y per phase, u per phase, q per phase, t per phase, s) in *unscaled*
        variables -- the previous iteration's solution; default: the problem's user guess.
        ``update_scaling`` / ``scaling_weight`` (pycollo/settings.py:272-296, defaults False / 0.8) with ``history`` =
        the ``scaling_record`` of every earlier mesh iteration, oldest first: scalings averaged over the mesh
        iterations (pycollo/scaling.py:283-344) instead of regenerated from the bounds."""
        self.problem = problem
        self.model = compile_model(problem)
        self.quad = QuadratureTables(self.model.quadrature_method)
        self.meshes = meshes or [build_phase_mesh(self.quad, *ph.mesh.resolved()) for ph in problem.phases]
        self.device = device
        self.engine = NlpEngine(problem, self.meshes, device=device, threads_per_block=threads_per_block)
        self.layout = self.engine.layout
        self.guess_x = self._interpolate_guess(prev if prev is not None else self._user_guess())
        V, r = self.layout.base_variable_scaling()
        self.V, self.r = self.layout.expand_x(V), self.layout.expand_x(r)
        self.guess_x_tilde = (self.guess_x - self.r) / self.V                    # scaling.py:172-174
        # first mesh iteration: w = 1, W from the Jacobian row norms at the guess (scaling.py:271-275); later ones
        # (default update_scaling = False -> _generate_from_base, scaling.py:277-281): w = 1 / ||grad J|| as well
        self.number = int(number)
        self.w = 1.0
        self.engine.set_scaling(V, r, np.ones(self.layout.num_ocp_c), 1.0)
        if self.number > 1 and update_scaling and history:
            # scaling.py:204-210,283-344.  As in the reference the NLP functions keep the V, r they were generated with
            # (backend.py:1459-1463 runs before generate_scaling, iteration.py:375-394); the averaged V, r scale the
            # bounds and unscale the solution (self.V, self.r below).
            w_now = objective_scaling(self.engine, self.guess_x_tilde)
            self.w, V_upd, r_upd, self.W_ocp = scaling_from_previous(
                self.layout, self.model, self.guess_x_tilde, V, r, w_now, history, scaling_weight,
                lambda V_rows: constraint_scaling(self.engine, self.guess_x_tilde, V_rows))
            self.V, self.r = self.layout.expand_x(V_upd), self.layout.expand_x(r_upd)
            self.scaling_record = (self.w, V_upd, r_upd, self.W_ocp)
        else:
            if self.number > 1:
                self.w = objective_scaling(self.engine, self.guess_x_tilde)
            self.W_ocp = constraint_scaling(self.engine, self.guess_x_tilde)
            self.scaling_record = (self.w, V.copy(), r.copy(), self.W_ocp)
        self.engine.set_scaling(V, r, self.W_ocp, self.w)
        self.x_bnd_l, self.x_bnd_u, self.c_bnd_l, self.c_bnd_u = self._bounds()

    # ---- guess ------------------------------------------------------------------------------------
    def _user_guess(self):
        from . import problem as pb
        taus, ys, us, qs, ts = [], [], [], [], []
        for ph, pm in zip(self.problem.phases, self.model.phases):
            g = ph.guess
            if g.time is None:
                raise ValueError(f"phase {ph.name}: a guess is required")
            time = np.asarray(g.time, dtype=np.float64)
            t0, tF = time[0], time[-1]
            tau = (time - 0.5 * (t0 + tF)) / (0.5 * (tF - t0))                   # guess.py:165-170
            y_b, u_b, q_b = pb.phase_variable_bounds(ph)
            t_b = pb.phase_time_bounds(ph)
            yk, uk, qk, tk = pb.needed(y_b), pb.needed(u_b), pb.needed(q_b), pb.needed(list(t_b))
            y = np.asarray(g.state_variables, dtype=np.float64).reshape(len(ph._y), -1)[yk]
            u = np.asarray(g.control_variables if g.control_variables is not None else np.empty((0, len(time))),
                           dtype=np.float64).reshape(len(ph._u), -1)[uk]
            q = np.asarray(g.integral_variables if g.integral_variables is not None else [], dtype=np.float64).reshape(-1)[qk]
            taus.append(tau); ys.append(y); us.append(u); qs.append(q)
            ts.append(np.array([t0, tF])[tk])
        pg = self.problem.guess.parameter_variables
        s_b = pb._bounds_for(list(self.problem.parameter_variables), self.problem.bounds.parameter_variables, "parameter")
        s = np.asarray(pg if pg is not None else [], dtype=np.float64).reshape(-1)[pb.needed(s_b)] if s_b else np.zeros(0)
        return taus, ys, us, qs, ts, s

    def _interpolate_guess(self, prev):
        taus, ys, us, qs, ts, s = prev
        parts = []
        for tau_prev, y, u, q, t, mesh in zip(taus, ys, us, qs, ts, self.meshes):
            zu = np.vstack([y, u]) if (len(y) + len(u)) else np.empty((0, len(tau_prev)))
            parts.append(interp_linear(tau_prev, zu, mesh.tau, self.device).reshape(-1))
            parts.append(np.asarray(q, dtype=float).reshape(-1))
            parts.append(np.asarray(t, dtype=float).reshape(-1))
        parts.append(np.asarray(s, dtype=float).reshape(-1))
        x = np.concatenate(parts)
        if x.shape[0] != self.layout.num_x:
            raise ValueError("guess does not match the variable layout")
        return x

    # ---- bounds (iteration.py:408-453) -----------------------------------------------------------------
    def _bounds(self):
        xb = []
        cb = []
        for pm, pl in zip(self.model.phases, self.layout.phases):
            N = pl.N
            yb = pm.x_bounds[:pm.n_y]
            for b, b0, bF in zip(yb, pm.y_t0_bounds, pm.y_tF_bounds):
                xb.extend([b0] + [b] * (N - 2) + [bF])
            for b in pm.x_bounds[pm.n_y:pm.n_z]:
                xb.extend([b] * N)
            xb.extend(pm.x_bounds[pm.n_z:])
            cb.extend([(0.0, 0.0)] * (pm.n_y * (N - 1)))
            for b in pm.p_bounds:
                cb.extend([b] * N)
            cb.extend([(0.0, 0.0)] * pm.n_q)
        xb.extend(self.model.s_bounds)
        cb.extend(self.model.point.b_bounds)
        xb = np.asarray(xb, dtype=np.float64).reshape(-1, 2)
        cb = np.asarray(cb, dtype=np.float64).reshape(-1, 2)
        W = self.layout.expand_c(self.W_ocp)
        return ((xb[:, 0] - self.r) / self.V, (xb[:, 1] - self.r) / self.V, W * cb[:, 0], W * cb[:, 1])

    # ---- solve -------------------------------------------------------------------------------------
    def solve_with_ipm(self, max_iter: int = 500, tol: float = 1e-8, verbose: int = 0, linear_solver: str = "resident",
                       warm_start: bool = False, host_retry: bool = False):
        """Solve the scaled NLP with the interior-point stand-in for IPOPT (``pycollo_amd.ipm``), driven through
        the cyipopt-protocol object exactly as ``ipopt.problem(...).solve(x0)`` would be (pycollo/nlp.py:84-115).
        ``linear_solver``: "resident" (default) -- as "gpu", and the iteration's vectors stay on the device too
        (``ipm.ResidentInteriorPointSolver``, csrc/pc_ipm.hpp); "gpu" -- the KKT systems are assembled from device-resident G~ / H~ and factorised on the
        GPU (``pycollo_amd.kkt``; the role of IPOPT's ``linear_solver`` option, pycollo/backend.py:1703-1711);
        "host" -- the callbacks deliver host arrays and SuperLU factorises (what a host-side IPOPT would do);
        "sharded" -- called by every rank of a ``torch.distributed`` world: sharded evaluation, KKT factorisation cut across
        the ranks, replicated vectors (``ipm_sharded.solve_sharded``).
        ``warm_start``: pycollo's setting of that name (settings.py:228, backend.py:1703-1709).
        ``host_retry`` (off by default): repeat an NLP whose GPU-factorised solve did not succeed once more with the host
        factorisation (a second full solve on SuperLU; reported through a RuntimeWarning and the result's flags)."""
        if linear_solver == "sharded":
            # every rank of the torch.distributed world calls this with the same mesh iteration: the evaluation sharded by
            # section ranges, the KKT factorisation cut across the ranks, the iteration on replicated vectors (ipm_sharded.py)
            from .ipm_sharded import solve_sharded
            res, sh = solve_sharded(self, max_iter=max_iter, tol=tol, verbose=verbose, warm_start=warm_start,
                                    second_order_correction=os.environ.get("PYCOLLO_AMD_SOC", "1") != "0")
            sh.release()
            self.result = res
            self.x_tilde = res.x
            self.objective = res.objective / self.w
            return res
        from .engine import PycolloGpuProblem
        from .ipm import GpuInteriorPointSolver, InteriorPointSolver, ResidentInteriorPointSolver
        pobj = PycolloGpuProblem(self.engine)
        # "resident": the GPU factorisation AND the iteration's vectors on the device (csrc/pc_ipm.hpp)
        cls = {"gpu": GpuInteriorPointSolver, "host": InteriorPointSolver, "resident": ResidentInteriorPointSolver}[linear_solver]
        solver = cls(pobj, pobj.n, pobj.m, self.x_bnd_l, self.x_bnd_u, self.c_bnd_l, self.c_bnd_u,
                     tol=tol, max_iter=max_iter, verbose=verbose, warm_start=warm_start,
                     second_order_correction=os.environ.get("PYCOLLO_AMD_SOC", "1") != "0")   # (A/B knob: IPOPT's max_soc = 0)
        # The model's SymPy graphs are millions of long-lived objects: a full collection walking them takes ~80 ms and
        # strikes in the middle of whichever linear solve allocates the unlucky array (measured: 8 such stalls in
        # an 18-iteration solve, more than all factorisations together).  Collection is off for the duration.
        import gc
        was_enabled = gc.isenabled()
        gc.freeze()       # (no collection first: walking those graphs is a quarter of a second, twice the solve at config 2)
        gc.disable()      # (what the solve allocates is arrays: reference counting frees them)
        # The other source of ~80-100 ms stalls anywhere in the process: numpy's BLAS starts one spinning thread per
        # hardware thread (256 on the MI355X host) for a dot product of 30 k doubles; inside a CPU-quota cgroup they
        # burn the period's budget and the whole process is throttled until the next period.
        try:
            from threadpoolctl import threadpool_limits
            blas_limit = threadpool_limits(limits=min(8, os.cpu_count() or 8))
        except Exception:        # threadpoolctl not installed: nothing to limit with
            blas_limit = None
        try:
            res = solver.solve(self.guess_x_tilde)
        finally:
            if blas_limit is not None:
                blas_limit.restore_original_limits()
            if was_enabled:
                gc.enable()
            gc.unfreeze()
        if linear_solver in ("gpu", "resident") and not res.success and host_retry:
            # The two linear solvers round differently; on a degenerate NLP (a bang-bang solution on a coarse mesh) that
            # can send the filter line search into its restoration phase on one path and not on the other.  A failed
            # GPU-factorised solve is repeated once with the host factorisation before the mesh iteration is given up;
            # the result says so.
            import warnings
            first, first_seconds, first_iterations = res.status, res.seconds, res.iterations
            warnings.warn(f"the GPU-factorised interior-point solve ended with '{first}' after {first_iterations} iterations; "
                          f"repeating this NLP with the host factorisation", RuntimeWarning, stacklevel=2)
            solver = InteriorPointSolver(pobj, pobj.n, pobj.m, self.x_bnd_l, self.x_bnd_u, self.c_bnd_l, self.c_bnd_u,
                                         tol=tol, max_iter=max_iter, verbose=verbose, warm_start=warm_start)
            res = solver.solve(self.guess_x_tilde)
            res.evaluations["gpu_linear_solver_gave_up"] = first     # never silent: a warning, this flag, OcpResult's list
            res.seconds += first_seconds                              # both attempts are this mesh iteration's time
            res.evaluations["gpu_first_attempt_iterations"] = first_iterations
        self.result = res
        self.x_tilde = res.x
        self.objective = res.objective / self.w                                 # scaling.py:186-189
        return res

    def solution(self):
        """The NLP point ``x_tilde`` as unscaled (tau, y, u, q, t) per phase and s -- the ``prev`` of the next mesh
        iteration (pycollo/solution/solution_abc.py:37-58 extraction, pycollo/iteration.py:86-194 consumer)."""
        x = self.V * self.x_tilde + self.r
        taus, ys, us, qs, ts = [], [], [], [], []
        for pm, pl, mesh in zip(self.model.phases, self.layout.phases, self.meshes):
            N = pl.N
            z = x[pl.x_off:pl.x_off + pm.n_z * N].reshape(pm.n_z, N)
            taus.append(np.asarray(mesh.tau, dtype=float))
            ys.append(z[:pm.n_y].copy())
            us.append(z[pm.n_y:].copy())
            qs.append(x[pl.q_off:pl.q_off + pm.n_q].copy())
            ts.append(x[pl.t_off:pl.t_off + pl.n_t].copy())
        return taus, ys, us, qs, ts, x[self.layout.s_off:self.layout.s_off + self.layout.n_s].copy()

    def solve_with_scipy(self, maxiter: int = 500, tol: float = 1e-9, verbose: int = 0):
        """Solve the scaled NLP with scipy's trust-region interior point method (stand-in for IPOPT)."""
        import scipy.sparse as sp
        from scipy.optimize import Bounds, NonlinearConstraint, minimize
        e = self.engine
        gr, gc = e.evaluate_G_structure()
        hr, hc = e.evaluate_H_structure()
        n, m = e.num_x, e.num_c
        lo = hr != hc

        def sym(vals):
            H = sp.coo_matrix((vals, (hr, hc)), shape=(n, n))
            return (H + sp.coo_matrix((vals[lo], (hc[lo], hr[lo])), shape=(n, n))).tocsr()

        zero_lam = np.zeros(m)
        con = NonlinearConstraint(lambda x: e.evaluate_c(x), self.c_bnd_l, self.c_bnd_u,
                                  jac=lambda x: sp.csr_matrix((e.evaluate_G_nonzeros(x), (gr, gc)), shape=(m, n)),
                                  hess=lambda x, v: sym(e.evaluate_H_nonzeros(x, 0.0, v)))
        res = minimize(e.evaluate_J, self.guess_x_tilde, jac=e.evaluate_g,
                       hess=lambda x: sym(e.evaluate_H_nonzeros(x, 1.0, zero_lam)),
                       method="trust-constr", constraints=[con],
                       bounds=Bounds(self.x_bnd_l, self.x_bnd_u, keep_feasible=False),
                       options={"maxiter": maxiter, "gtol": tol, "xtol": 1e-12, "verbose": verbose})
        self.result = res
        self.x_tilde = res.x
        self.objective = res.fun / self.w                                       # scaling.py:186-189
        return res
