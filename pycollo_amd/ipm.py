"""A compact primal-dual interior-point NLP solver behind the cyipopt ``problem_obj`` protocol.

SURVEY.md section 8f row N3.  The reference hands its callbacks to IPOPT (pycollo/nlp.py:84-115, cyipopt, or
``ca.nlpsol(..., "ipopt")``, pycollo/backend.py:1681-1711); neither IPOPT nor cyipopt exists in this image, so
this module stands in for it: it drives *exactly* the callback surface IPOPT would -- ``objective``, ``gradient``,
``constraints``, ``jacobian`` / ``jacobianstructure``, ``hessian(x, lagrange, obj_factor)`` / ``hessianstructure``
of :class:`pycollo_amd.engine.PycolloGpuProblem` -- so the end-to-end ``solve()`` objectives of the reference's
integration tests can be used as tests of the GPU path.

Algorithm: the line-search filter interior-point method of Waechter & Biegler (Math. Program. 106, 2006), the
method IPOPT implements, without its restoration phase and second-order correction:

  min f(x)  s.t.  c_E(x) = c_E^L,  c_I^L <= c_I(x) <= c_I^U,  x^L <= x <= x^U
  ->  slacks s for the inequality rows;  barrier  phi_mu = f - mu sum log(v - v^L) - mu sum log(v^U - v);
  Newton on the primal-dual equations with Sigma = Z^L/(v - v^L) + Z^U/(v^U - v) folded into the (1,1) block,
  inertia-free regularisation (curvature test on the step), fraction-to-the-boundary rule, filter on
  (constraint violation, barrier objective), monotone mu update.

The linear algebra goes through four methods (``_JT``, ``_solve_kkt``, ``_ls_multipliers``, ``_gn_step``).
:class:`InteriorPointSolver` implements them on the host with SuperLU (``scipy.sparse.linalg.splu``) on matrices the
callbacks delivered as host arrays; :class:`GpuInteriorPointSolver` (row N4) implements them with the block L D L^T
of ``pycollo_amd.kkt`` on G~ / H~ that never leave device memory -- only vectors cross the bus.
Multiplier sign convention: L = obj_factor f + lambda^T c, as IPOPT's ``eval_h`` expects.
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

INF = 1e19   # IPOPT's nlp_lower/upper_bound_inf


@dataclass
class IpmResult:
    x: np.ndarray
    lam: np.ndarray
    objective: float
    status: str
    iterations: int
    inf_pr: float
    inf_du: float
    mu: float
    seconds: float
    evaluations: dict = field(default_factory=dict)
    history: list = field(default_factory=list)
    zl: np.ndarray | None = None       # multipliers of the variables' lower / upper bounds (of the unscaled problem)
    zu: np.ndarray | None = None

    @property
    def success(self) -> bool:
        return self.status in ("optimal", "acceptable")


class InteriorPointSolver:
    def __init__(self, problem_obj, n: int, m: int, lb, ub, cl, cu, tol: float = 1e-8, acceptable_tol: float = 1e-6,
                 max_iter: int = 300, mu_init: float = 0.1, verbose: int = 0, warm_start: bool = False,
                 gradient_scaling: bool = True, second_order_correction: bool = True):
        self.p, self.n, self.m = problem_obj, int(n), int(m)
        self.second_order_correction = bool(second_order_correction)
        # IPOPT's warm_start_init_point (the one thing pycollo's warm_start setting switches, backend.py:1703-1709):
        # the starting point is kept closer to where it was given -- bound push / fraction 1e-3 instead of 1e-2
        # (warm_start_bound_push, warm_start_bound_frac).  No multipliers are handed over: the reference passes none.
        self.warm_start = bool(warm_start)
        self.gradient_scaling = bool(gradient_scaling)   # IPOPT's nlp_scaling_method: gradient-based (default) or none
        self.lb, self.ub = np.asarray(lb, float).copy(), np.asarray(ub, float).copy()
        self.cl, self.cu = np.asarray(cl, float).copy(), np.asarray(cu, float).copy()
        self.tol, self.acceptable_tol, self.max_iter, self.mu_init, self.verbose = tol, acceptable_tol, max_iter, mu_init, verbose
        self.eq = np.isclose(self.cl, self.cu, rtol=0, atol=0) | (np.abs(self.cu - self.cl) <= 1e-14 * np.maximum(1.0, np.abs(self.cl)))
        self.ineq = np.nonzero(~self.eq)[0]
        self.ns = len(self.ineq)
        self.nv = self.n + self.ns
        # bounds on v = [x ; s]
        self.vl = np.concatenate([self.lb, self.cl[self.ineq]])
        self.vu = np.concatenate([self.ub, self.cu[self.ineq]])
        # fixed variables (x^L = x^U: pinned initial states, fixed times) are parameters, as in IPOPT's default
        # fixed_variable_treatment = make_parameter: no barrier terms, no step components
        self.fixed = (self.vu - self.vl) <= 1e-12 * np.maximum(1.0, np.abs(self.vl))
        self.free = np.nonzero(~self.fixed)[0]
        self.nf = len(self.free)
        self.hasl = (self.vl > -INF) & ~self.fixed
        self.hasu = (self.vu < INF) & ~self.fixed
        jr, jc = problem_obj.jacobianstructure()
        hr, hc = problem_obj.hessianstructure()
        self.jr, self.jc = np.asarray(jr, np.int64), np.asarray(jc, np.int64)
        self.hr, self.hc = np.asarray(hr, np.int64), np.asarray(hc, np.int64)
        self.hoff = self.hr != self.hc
        # constant part of the constraint Jacobian w.r.t. the slacks
        self.Js = sp.csr_matrix((-np.ones(self.ns), (self.ineq, np.arange(self.ns))), shape=(self.m, self.ns))
        self.rhs_c = np.where(self.eq, self.cl, 0.0)
        self.sf = 1.0                       # IPOPT's gradient-based NLP scaling (set in solve())
        self.sc = np.ones(self.m)
        self.counts = {"objective": 0, "gradient": 0, "constraints": 0, "jacobian": 0, "hessian": 0, "factorisations": 0}
        self.kkt_seconds = 0.0              # wall time inside _solve_kkt (assembly, factorisations, solves, refinement)

    # ---- callbacks --------------------------------------------------------------------------------
    def _f(self, x):
        self.counts["objective"] += 1
        return self.sf * float(self.p.objective(x))

    def _g(self, x):
        self.counts["gradient"] += 1
        return self.sf * np.asarray(self.p.gradient(x), float)

    def _c(self, v):
        self.counts["constraints"] += 1
        c = self.sc * (np.asarray(self.p.constraints(v[:self.n]), float) - self.rhs_c)
        if self.ns:
            c[self.ineq] -= v[self.n:]      # slacks live in the scaled constraint's units
        return c

    def _J(self, x):
        self.counts["jacobian"] += 1
        Jx = sp.csr_matrix((np.asarray(self.p.jacobian(x), float) * self.sc[self.jr], (self.jr, self.jc)), shape=(self.m, self.n))
        return sp.hstack([Jx, self.Js], format="csr") if self.ns else Jx

    def _W(self, x, lam):
        self.counts["hessian"] += 1
        vals = np.asarray(self.p.hessian(x, self.sc * lam, self.sf), float)
        W = sp.coo_matrix((vals, (self.hr, self.hc)), shape=(self.nv, self.nv))
        Wt = sp.coo_matrix((vals[self.hoff], (self.hc[self.hoff], self.hr[self.hoff])), shape=(self.nv, self.nv))
        return (W + Wt).tocsr()

    # ---- pieces -----------------------------------------------------------------------------------
    def _push_interior(self, v, k1=1e-2, k2=1e-2):
        v = v.copy()
        both = self.hasl & self.hasu
        onlyl = self.hasl & ~self.hasu
        onlyu = self.hasu & ~self.hasl
        with np.errstate(invalid="ignore"):   # (an infinite bound gives inf - inf in entries the masks leave out)
            pl = np.minimum(k1 * np.maximum(1.0, np.abs(self.vl)), k2 * (self.vu - self.vl))
            pu = np.minimum(k1 * np.maximum(1.0, np.abs(self.vu)), k2 * (self.vu - self.vl))
            v[both] = np.clip(v[both], (self.vl + pl)[both], (self.vu - pu)[both])
            v[onlyl] = np.maximum(v[onlyl], (self.vl + k1 * np.maximum(1.0, np.abs(self.vl)))[onlyl])
            v[onlyu] = np.minimum(v[onlyu], (self.vu - k1 * np.maximum(1.0, np.abs(self.vu)))[onlyu])
        return v

    def _barrier(self, v, f, mu):
        b = 0.0
        if self.hasl.any():
            b -= np.sum(np.log((v - self.vl)[self.hasl]))
        if self.hasu.any():
            b -= np.sum(np.log((self.vu - v)[self.hasu]))
        return f + mu * b

    def _alpha_max(self, v, dv, tau):
        a = 1.0
        dl = (v - self.vl)
        du = (self.vu - v)
        neg = self.hasl & (dv < 0)
        if neg.any():
            a = min(a, float(np.min(-tau * dl[neg] / dv[neg])))
        pos = self.hasu & (dv > 0)
        if pos.any():
            a = min(a, float(np.min(tau * du[pos] / dv[pos])))
        return a

    @staticmethod
    def _alpha_dual(z, dz, tau):
        neg = dz < 0
        return min(1.0, float(np.min(-tau * z[neg] / dz[neg]))) if neg.any() else 1.0

    def _solve_kkt(self, W, Sigma, J, r1, r2, dw_last):
        """Solve [[W + Sigma + dw I, J^T], [J, -dc I]] [dv; dlam] = [r1; r2] with the inertia-free curvature test."""
        nv, m, fr = self.nf, self.m, self.free
        dw, dc = 0.0, 0.0
        base = (W + sp.diags(Sigma))[fr][:, fr]
        J = J[:, fr]
        r1 = r1[fr]
        rhs = np.concatenate([r1, r2])
        for attempt in range(40):
            # The (2,2) block always carries a tiny -dc I: the matrix is then quasi-definite whenever the (1,1) block
            # is positive definite, a symmetric-mode LU that pivots on the diagonal exists, and the signs of U's
            # diagonal are the inertia (Sylvester) -- which is what IPOPT's regularisation is driven by
            # (n positive, m negative eigenvalues).  SuperLU with diag_pivot_thresh = 0 is that factorisation as
            # long as it reports the same row and column permutation.
            dc_eff = max(dc, 1e-9)
            K = sp.bmat([[base + dw * sp.identity(nv), J.T], [J, -dc_eff * sp.identity(m)]], format="csc")
            self.counts["factorisations"] += 1
            good, sol = False, None
            try:
                with np.errstate(all="ignore"):
                    lu = spla.splu(K, permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
                    d = lu.U.diagonal()
                    sym = np.array_equal(lu.perm_r, lu.perm_c)
                    npos, nneg = int(np.sum(d > 0)), int(np.sum(d < 0))
                    if sym and npos == nv and nneg == m:
                        # refine against the system actually wanted (dc, not dc_eff): the tiny (2,2) shift only
                        # makes the factorisation exist, it must not limit the accuracy of the step
                        K0 = K if dc_eff == dc else sp.bmat([[base + dw * sp.identity(nv), J.T],
                                                             [J, -dc * sp.identity(m) if dc > 0 else None]], format="csc")
                        sol = lu.solve(rhs)
                        res = rhs - K0 @ sol
                        for _ in range(3):
                            trial = sol + lu.solve(res)
                            res_t = rhs - K0 @ trial
                            if not np.all(np.isfinite(trial)) or np.linalg.norm(res_t) >= 0.5 * np.linalg.norm(res):
                                break          # (nearly) singular unshifted system: keep the regularised step
                            sol, res = trial, res_t
                        good = bool(np.all(np.isfinite(sol)))
                    elif not sym:
                        # off-diagonal pivots were needed: fall back to the curvature test on the step
                        sol = lu.solve(rhs)
                        if np.all(np.isfinite(sol)):
                            dv = sol[:nv]
                            good = bool(dv @ (base @ dv) + dw * (dv @ dv) >= 1e-11 * (dv @ dv))
            except RuntimeError:
                dc = 1e-8 if dc == 0.0 else dc * 10.0
            if good or (dw >= 1e20 and sol is not None and np.all(np.isfinite(sol))):
                full = np.zeros(self.nv)
                full[fr] = sol[:nv]
                self._last_factor = (lu, K)      # (second-order corrections solve with it again: _resolve_kkt)
                return full, sol[nv:], dw
            # wrong inertia / curvature (or singular): raise the primal regularisation, IPOPT's delta_w schedule
            if dw == 0.0:
                dw = 1e-4 if dw_last == 0.0 else max(1e-20, dw_last / 3.0)
            else:
                dw *= 100.0 if dw_last == 0.0 else 8.0
            if dw > 1e20:
                break
        raise RuntimeError("KKT regularisation failed")

    def _resolve_kkt(self, r1, r2, v=None, lam=None):
        """Another right-hand side for the matrix ``_solve_kkt`` factorised last (second-order correction); ``v``, ``lam``:
        the point the matrix belongs to (for a subclass whose matrix lives where trial evaluations overwrite it)."""
        lu, K = self._last_factor
        rhs = np.concatenate([r1[self.free], r2])
        sol = lu.solve(rhs)
        res = rhs - K @ sol
        for _ in range(2):
            trial = sol + lu.solve(res)
            res_t = rhs - K @ trial
            if not np.all(np.isfinite(trial)) or np.linalg.norm(res_t) >= 0.5 * np.linalg.norm(res):
                break
            sol, res = trial, res_t
        full = np.zeros(self.nv)
        full[self.free] = sol[:self.nf]
        return full, sol[self.nf:]

    def _JT(self, J, lam):
        """J^T lambda over v = [x ; s]."""
        return J.T @ lam

    def _ls_multipliers(self, J, gz):
        """Least-squares multiplier estimate: lambda of  min ||gz + J^T lambda||  over the free unknowns (IPOPT's
        initial multipliers); zeros when the estimate is unusable."""
        lam = np.zeros(self.m)
        try:
            Jf = J[:, self.free]
            K = sp.bmat([[sp.identity(self.nf), Jf.T], [Jf, None]], format="csc")
            sol = spla.splu(K).solve(np.concatenate([-gz[self.free], np.zeros(self.m)]))
            if self.m and np.all(np.isfinite(sol)) and np.max(np.abs(sol[self.nf:])) <= 1e3:
                lam = sol[self.nf:]
        except RuntimeError:
            pass
        return lam

    def _gn_step(self, J, c):
        """Minimum-norm step onto the linearised constraints (restoration): dv of [[I, J^T], [J, -1e-10 I]]."""
        Jf = J[:, self.free]
        self.counts["factorisations"] += 1
        K = sp.bmat([[sp.identity(self.nf), Jf.T], [Jf, -1e-10 * sp.identity(self.m)]], format="csc")
        sol = spla.splu(K).solve(np.concatenate([np.zeros(self.nf), -c]))
        dr = np.zeros(self.nv)
        dr[self.free] = sol[:self.nf]
        return dr

    # ---- main loop --------------------------------------------------------------------------------
    def solve(self, x0) -> IpmResult:
        t_start = time.perf_counter()
        n, nv, m = self.n, self.nv, self.m
        x0 = np.asarray(x0, float)
        # gradient-based scaling at the starting point, IPOPT's default nlp_scaling_method (the reference leaves it
        # on, pycollo/backend.py:1704-1710): objective and every constraint row are scaled down so that no
        # gradient entry exceeds nlp_scaling_max_gradient = 100
        self.sf, self.sc = 1.0, np.ones(m)
        # (evaluated at the starting point as the algorithm will use it, i.e. pushed off its bounds: a guess that sits
        #  exactly where a model is not differentiable -- Delta III on the pad, zero air speed under a square root --
        #  gives NaN partials at the raw guess; entries that are not finite anyway take no part in the maxima)
        push = (1e-3, 1e-3) if self.warm_start else (1e-2, 1e-2)
        x0c = self._push_interior(np.concatenate([x0, np.zeros(self.ns)]), *push)[:n]
        x0c[self.fixed[:n]] = self.vl[:n][self.fixed[:n]]
        g0 = np.nan_to_num(np.asarray(self.p.gradient(x0c), float), nan=0.0, posinf=0.0, neginf=0.0)
        gmax = float(np.max(np.abs(g0))) if n else 0.0
        if gmax > 100.0 and self.gradient_scaling:
            self.sf = max(100.0 / gmax, 1e-8)
        if m and self.gradient_scaling:
            jv = np.abs(np.nan_to_num(np.asarray(self.p.jacobian(x0c), float), nan=0.0, posinf=0.0, neginf=0.0))
            rowmax = np.zeros(m)
            np.maximum.at(rowmax, self.jr, jv)
            big = rowmax > 100.0
            self.sc[big] = np.maximum(100.0 / rowmax[big], 1e-8)
        if self.ns:
            self.vl[n:] = np.where(self.cl[self.ineq] > -INF, self.sc[self.ineq] * self.cl[self.ineq], self.cl[self.ineq])
            self.vu[n:] = np.where(self.cu[self.ineq] < INF, self.sc[self.ineq] * self.cu[self.ineq], self.cu[self.ineq])
        v = np.concatenate([x0, np.zeros(self.ns)])
        if self.ns:
            v[n:] = (self.sc * np.asarray(self.p.constraints(x0), float))[self.ineq]
        v = self._push_interior(v, 1e-3, 1e-3) if self.warm_start else self._push_interior(v)
        v[self.fixed] = self.vl[self.fixed]
        mu = self.mu_init
        zl = np.where(self.hasl, 1.0, 0.0)
        zu = np.where(self.hasu, 1.0, 0.0)
        f, g = self._f(v[:n]), np.concatenate([self._g(v[:n]), np.zeros(self.ns)])
        c, J = self._c(v), self._J(v[:n])
        lam = self._ls_multipliers(J, g - zl + zu)
        filt: list[tuple[float, float]] = []
        theta0 = float(np.sum(np.abs(c)))
        theta_max, theta_min = 1e4 * max(1.0, theta0), 1e-4 * max(1.0, theta0)
        dw_last = 0.0
        status, hist = "max_iter", []
        k_eps, k_mu, th_mu, s_max, g_th, g_phi, eta = 10.0, 0.2, 1.5, 100.0, 1e-5, 1e-8, 1e-4
        accept_count = 0
        restarts = 0
        it = 0
        last_alpha = last_amax = 0.0
        last_tag = ""
        inf_pr = inf_du = np.inf

        def errors(mu_):
            dL = (g + JTlam - zl + zu)[self.free]
            nz = max(1, int(self.hasl.sum() + self.hasu.sum()))
            sd = max(s_max, (np.sum(np.abs(lam)) + np.sum(zl) + np.sum(zu)) / (m + nz)) / s_max
            scz = max(s_max, (np.sum(zl) + np.sum(zu)) / nz) / s_max
            comp = 0.0
            with np.errstate(invalid="ignore"):   # (inf x 0 where there is no bound: masked out)
                if self.hasl.any():
                    comp = max(comp, float(np.max(np.abs(((v - self.vl) * zl - mu_)[self.hasl]))))
                if self.hasu.any():
                    comp = max(comp, float(np.max(np.abs(((self.vu - v) * zu - mu_)[self.hasu]))))
            e_du = float(np.max(np.abs(dL))) if self.nf else 0.0
            e_pr = float(np.max(np.abs(c))) if m else 0.0
            return max(e_du / sd, e_pr, comp / scz), e_pr, e_du

        phase = {"setup": time.perf_counter() - t_start, "errors": 0.0, "line_search": 0.0}
        for it in range(self.max_iter + 1):
            JTlam = self._JT(J, lam)
            t_ph = time.perf_counter()
            e0, inf_pr, inf_du = errors(0.0)
            phase["errors"] += time.perf_counter() - t_ph
            hist.append((it, f, inf_pr, inf_du, mu))
            if self.verbose:
                print(f"{it:4d}  f {f: .8e}  inf_pr {inf_pr:.2e}  inf_du {inf_du:.2e}  lg(mu) {np.log10(mu):5.1f}  dw {dw_last:.1e}"
                      f"  alpha {last_alpha:.2e} (max {last_amax:.2e}){last_tag}")
            if e0 <= self.tol:
                status = "optimal"
                break
            accept_count = accept_count + 1 if e0 <= self.acceptable_tol else 0
            if accept_count >= 15:
                status = "acceptable"
                break
            if it == self.max_iter:
                break
            while errors(mu)[0] <= k_eps * mu and mu > self.tol / 10:
                mu = max(self.tol / 10, min(k_mu * mu, mu ** th_mu))
                filt = []
            tau = max(0.99, 1.0 - mu)
            # Newton step
            dlv, duv = np.where(self.hasl, v - self.vl, 1.0), np.where(self.hasu, self.vu - v, 1.0)
            Sigma = np.where(self.hasl, zl / dlv, 0.0) + np.where(self.hasu, zu / duv, 0.0)
            W = self._W(v[:n], lam)
            grad_phi = g - np.where(self.hasl, mu / dlv, 0.0) + np.where(self.hasu, mu / duv, 0.0)
            t_kkt = time.perf_counter()
            try:
                dv, dlam, dw_last = self._solve_kkt(W, Sigma, J, -(grad_phi + JTlam), -c, dw_last)
            except RuntimeError:
                status = "kkt_failure"
                break
            finally:
                self.kkt_seconds += time.perf_counter() - t_kkt
            dzl = np.where(self.hasl, mu / dlv - zl - zl / dlv * dv, 0.0)
            dzu = np.where(self.hasu, mu / duv - zu + zu / duv * dv, 0.0)
            a_max = self._alpha_max(v, dv, tau)
            a_z = min(self._alpha_dual(zl[self.hasl], dzl[self.hasl], tau) if self.hasl.any() else 1.0,
                      self._alpha_dual(zu[self.hasu], dzu[self.hasu], tau) if self.hasu.any() else 1.0)
            # filter line search
            t_ph = time.perf_counter()
            theta = float(np.sum(np.abs(c)))
            phi = self._barrier(v, f, mu)
            dphi = float(grad_phi @ dv)
            alpha, accepted = a_max, False
            a_min = 1e-12

            def acceptable(alpha_, ft_, th_t_, phi_t_):
                """The filter's verdict on a trial point reached with step size alpha_ (IPOPT A-5.4 .. A-5.8); a point
                accepted by the sufficient-decrease test augments the filter (returned as the second value)."""
                if not (np.isfinite(ft_) and np.isfinite(th_t_) and np.isfinite(phi_t_) and th_t_ <= theta_max):
                    return False, False
                if any(th_t_ >= th_f and phi_t_ >= ph_f for th_f, ph_f in filt):
                    return False, False
                if dphi < 0 and alpha_ * (-dphi) ** 2.3 > theta ** 1.1 and theta <= theta_min:
                    return phi_t_ <= phi + eta * alpha_ * dphi, False
                ok = th_t_ <= (1 - g_th) * theta or phi_t_ <= phi - g_phi * theta
                return ok, ok

            first = True
            while alpha > a_min:
                vt = v + alpha * dv
                with np.errstate(all="ignore"):
                    ft = self._f(vt[:n])
                    ct = self._c(vt)
                    th_t = float(np.sum(np.abs(ct)))
                    phi_t = self._barrier(vt, ft, mu)
                accepted, augment = acceptable(alpha, ft, th_t, phi_t)
                if self.verbose >= 2 and first:
                    print(f"      first trial: alpha {alpha:.2e} theta {theta:.3e} -> {th_t:.3e}  phi {phi:.10e} -> {phi_t:.10e}  dphi {dphi:.3e} "
                          f"switching {dphi < 0 and alpha * (-dphi) ** 2.3 > theta ** 1.1 and theta <= theta_min} theta_min {theta_min:.1e} "
                          f"in filter {any(th_t >= th_f and phi_t >= ph_f for th_f, ph_f in filt)} accepted {accepted}")
                if accepted:
                    if augment:
                        filt.append(((1 - g_th) * theta, phi - g_phi * theta))
                    break
                if first and self.second_order_correction and np.isfinite(th_t) and th_t >= theta:
                    # Second-order correction (IPOPT A-5.7 .. A-5.9): the full step was rejected and did not even reduce
                    # the violation -- the linearisation of c is what fails.  Up to four corrected steps from the same
                    # factorisation, each aimed at the violation the previous trial point was left with.
                    c_soc, th_old, a_soc = alpha * c + ct, theta, alpha
                    for _ in range(4):
                        self.counts["second_order_corrections"] = self.counts.get("second_order_corrections", 0) + 1
                        try:
                            dv_s, dlam_s = self._resolve_kkt(-(grad_phi + JTlam), -c_soc, v, lam)
                        except RuntimeError:
                            break
                        if not (np.all(np.isfinite(dv_s)) and np.all(np.isfinite(dlam_s))):
                            break
                        a_soc = self._alpha_max(v, dv_s, tau)
                        vs = v + a_soc * dv_s
                        with np.errstate(all="ignore"):
                            fs = self._f(vs[:n])
                            cs = self._c(vs)
                            th_s = float(np.sum(np.abs(cs)))
                            phi_s = self._barrier(vs, fs, mu)
                        ok_s, augment = acceptable(alpha, fs, th_s, phi_s)
                        if self.verbose >= 2:
                            print(f"      correction: alpha {a_soc:.2e} theta -> {th_s:.3e}  phi -> {phi_s:.10e}  accepted {ok_s}")
                        if ok_s:
                            if augment:
                                filt.append(((1 - g_th) * theta, phi - g_phi * theta))
                            accepted, dv, dlam, alpha, vt, ft, ct = True, dv_s, dlam_s, a_soc, vs, fs, cs
                            dzl = np.where(self.hasl, mu / dlv - zl - zl / dlv * dv, 0.0)
                            dzu = np.where(self.hasu, mu / duv - zu + zu / duv * dv, 0.0)
                            a_z = min(self._alpha_dual(zl[self.hasl], dzl[self.hasl], tau) if self.hasl.any() else 1.0,
                                      self._alpha_dual(zu[self.hasu], dzu[self.hasu], tau) if self.hasu.any() else 1.0)
                            self.counts["second_order_steps"] = self.counts.get("second_order_steps", 0) + 1
                            break
                        if not np.isfinite(th_s) or th_s > 0.99 * th_old:
                            break
                        c_soc, th_old = a_soc * c_soc + cs, th_s
                    if accepted:
                        break
                first = False
                alpha *= 0.5
            phase["line_search"] += time.perf_counter() - t_ph
            if not accepted and e0 <= self.acceptable_tol:
                # IPOPT's rule for a line search that fails at a point already inside the acceptable tolerances
                # (BacktrackingLineSearch -> STOP_AT_ACCEPTABLE_POINT): no restoration from a converged point whose
                # remaining error is the linear solver's floor
                status = "acceptable"
                self.counts["acceptable_after_failed_line_search"] = 1   # (surfaced by solve_ocp: not the requested tolerance)
                break
            if not accepted:
                # feasibility restoration, reduced to its core: Gauss-Newton steps on ||c||_1 (minimum-norm
                # solution of the linearised constraints, fraction-to-the-boundary, backtracking) until the
                # violation has dropped by a tenth and the point is acceptable to the filter
                filt.append(((1 - g_th) * theta, phi - g_phi * theta))
                vr, cr_, Jr, ok_r = v.copy(), c, J, False
                for _ in range(30):
                    th_r = float(np.sum(np.abs(cr_)))
                    try:
                        dr = self._gn_step(Jr, cr_)
                    except RuntimeError:
                        break
                    if not np.all(np.isfinite(dr)):
                        break
                    a = self._alpha_max(vr, dr, tau)
                    moved = False
                    while a > 1e-10:
                        vt = vr + a * dr
                        with np.errstate(all="ignore"):
                            ct = self._c(vt)
                        if np.all(np.isfinite(ct)) and float(np.sum(np.abs(ct))) < (1 - 1e-4 * a) * th_r:
                            vr, cr_, moved = vt, ct, True
                            break
                        a *= 0.5
                    if not moved:
                        break
                    Jr = self._J(vr[:n])
                    th_t = float(np.sum(np.abs(cr_)))
                    if th_t <= 0.9 * theta:
                        with np.errstate(all="ignore"):
                            ft = self._f(vr[:n])
                            phi_t = self._barrier(vr, ft, mu)
                        if np.isfinite(phi_t) and not any(th_t >= th_f and phi_t >= ph_f for th_f, ph_f in filt):
                            ok_r = True
                            break
                if not ok_r:
                    # The Gauss-Newton restoration could not find a point the filter accepts.  Before giving the NLP up:
                    # a barrier restart from the best restoration point -- mu back up (two decades, at most mu_init), the
                    # filter emptied, multipliers re-estimated and the bound multipliers put back on the central path
                    # -- at most three times per solve.  (IPOPT recovers from the same situation inside its restoration
                    # NLP; the degenerate cases here are bang-bang solutions on coarse meshes, where the two
                    # factorisations' rounding decides which side of the filter a trial point falls on.)
                    restarts += 1
                    if restarts > 3:
                        status = "restoration_failed"
                        break
                    if float(np.sum(np.abs(cr_))) < theta:      # keep whatever feasibility the restoration did gain
                        v = vr
                    mu = min(self.mu_init, max(100.0 * mu, 1e-6))
                    filt = []
                    # the fraction-to-the-boundary steps can have left components ON a bound in floating point: back
                    # inside by the distance the new central path keeps from it, then everything at that point anew
                    v = self._push_interior(v, min(1e-2, mu), min(1e-2, mu))
                    c, J, f = self._c(v), self._J(v[:n]), self._f(v[:n])
                    g = np.concatenate([self._g(v[:n]), np.zeros(self.ns)])
                    dlv, duv = np.where(self.hasl, v - self.vl, 1.0), np.where(self.hasu, self.vu - v, 1.0)
                    zl = np.where(self.hasl, mu / dlv, 0.0)
                    zu = np.where(self.hasu, mu / duv, 0.0)
                    lam = self._ls_multipliers(J, g - zl + zu)
                    last_alpha, last_amax, last_tag = 0.0, a_max, " r"
                    continue
                v, c, J, f = vr, cr_, Jr, ft
                last_alpha, last_amax, last_tag = 0.0, a_max, " R"
                g = np.concatenate([self._g(v[:n]), np.zeros(self.ns)])
                # multipliers after restoration: least squares, as at the start
                lam = self._ls_multipliers(J, g - zl + zu)
                continue
            last_alpha, last_amax, last_tag = alpha, a_max, ""
            v = v + alpha * dv
            lam = lam + alpha * dlam
            zl = zl + a_z * dzl
            zu = zu + a_z * dzu
            # keep the duals within a factor of the primal estimates
            ks = 1e10
            dlv, duv = np.where(self.hasl, v - self.vl, 1.0), np.where(self.hasu, self.vu - v, 1.0)
            zl = np.where(self.hasl, np.clip(zl, mu / (ks * dlv), ks * mu / dlv), 0.0)
            zu = np.where(self.hasu, np.clip(zu, mu / (ks * duv), ks * mu / duv), 0.0)
            f, c = ft, ct
            g = np.concatenate([self._g(v[:n]), np.zeros(self.ns)])
            J = self._J(v[:n])
        ev = dict(self.counts)
        ev["barrier_restarts"] = restarts
        ev["kkt_seconds"] = self.kkt_seconds
        ev["phase_seconds"] = {k: round(v, 6) for k, v in phase.items()}   # wall time by part of the loop (with kkt_seconds)
        return IpmResult(x=v[:n].copy(), lam=(self.sc * lam / self.sf).copy(), objective=f / self.sf, status=status, iterations=it, inf_pr=inf_pr,
                         inf_du=inf_du, mu=mu, seconds=time.perf_counter() - t_start, evaluations=ev, history=hist,
                         zl=(zl[:n] / self.sf).copy(), zu=(zu[:n] / self.sf).copy())


class GpuInteriorPointSolver(InteriorPointSolver):
    """The same algorithm with its linear algebra on the GPU.

    G~ and H~ are evaluated into device memory (``NlpEngine.evaluate_resident``) and consumed there by the block
    L D L^T of ``pycollo_amd.kkt.GpuKkt``; what crosses the bus per iteration is a handful of vectors (x~, lambda,
    Sigma, right-hand sides, the step).  ``problem_obj`` must be a :class:`pycollo_amd.engine.PycolloGpuProblem`.
    Replaces the linear solver inside IPOPT that the reference picks by name (pycollo/backend.py:1703-1711)."""

    def __init__(self, problem_obj, n, m, lb, ub, cl, cu, **kw):
        super().__init__(problem_obj, n, m, lb, ub, cl, cu, **kw)
        self.engine = problem_obj.engine
        self.kkt = None          # built in solve(), once the row scaling is known
        self.times = {"evaluate": 0.0, "factor": 0.0, "solve": 0.0, "matvec": 0.0}

    def _ensure_kkt(self):
        if self.kkt is None:
            from .kkt import GpuKkt
            t0 = time.perf_counter()
            self.kkt = GpuKkt(self.engine, self.ineq, self.fixed, self.sc)
            self.times["tables"] = self.times.get("tables", 0.0) + time.perf_counter() - t0   # once per solve
            self.times["tables_host"] = self.kkt.seconds_tables
        return self.kkt

    # G~ (and H~) live on the device; the "matrix" handed around is a token saying at which point they were evaluated
    def _J(self, x):
        self.counts["jacobian"] += 1
        t0 = time.perf_counter()
        self.engine.evaluate_resident(x, self.sf, None, want_grad=False)
        self.times["evaluate"] += time.perf_counter() - t0
        return ("resident", self.counts["jacobian"])

    def _W(self, x, lam):
        self.counts["hessian"] += 1
        t0 = time.perf_counter()
        self.engine.evaluate_resident(x, self.sf, self.sc * lam, want_grad=False)
        self.times["evaluate"] += time.perf_counter() - t0
        return ("resident", self.counts["hessian"])

    def _JT(self, J, lam):
        k = self._ensure_kkt()
        t0 = time.perf_counter()
        y = k.matvec(np.zeros(self.nv + self.m), np.concatenate([np.zeros(self.nv), lam]), use_hess=False)
        self.times["matvec"] += time.perf_counter() - t0
        return y[:self.nv]

    def _factor(self, dvec, use_hess):
        self.counts["factorisations"] += 1
        k = self._ensure_kkt()
        t0 = time.perf_counter()
        out = k.factor(dvec, use_hess)
        self.times["factor"] += time.perf_counter() - t0
        return out

    def _refined_solve(self, rhs, dvec_true, use_hess):
        """Solve with the current factors, refined against the system with ``dvec_true`` on the diagonal."""
        k = self.kkt
        t0 = time.perf_counter()
        # solve, residual and up to three corrections (each kept only while it halves the residual and stays finite) in
        # one device-side call: the vectors cross the bus once (pc_kkt_solve_refined)
        sol, n_solves = k.solve_refined(rhs, dvec_true, use_hess, max_steps=3)
        self.times["solve"] += time.perf_counter() - t0
        self.counts["kkt_solves"] = self.counts.get("kkt_solves", 0) + n_solves
        self.counts["refined_solves"] = self.counts.get("refined_solves", 0) + 1
        return sol

    def _solve_kkt(self, W, Sigma, J, r1, r2, dw_last):
        nv, m = self.nv, self.m
        rhs = np.concatenate([np.where(self.fixed, 0.0, r1), r2])
        dw, dc = 0.0, 0.0
        for attempt in range(40):
            dc_eff = max(dc, 1e-9)
            dvec = np.concatenate([Sigma + dw, np.full(m, -dc_eff)])
            npos, nneg = self._factor(dvec, True)
            sol = None
            if npos == nv and nneg == m:        # fixed unknowns are unit pivots: counted with the primal ones
                with np.errstate(all="ignore"):
                    sol = self._refined_solve(rhs, np.concatenate([Sigma + dw, np.full(m, -dc)]), True)
                if np.all(np.isfinite(sol)):
                    self._last_dvec = np.concatenate([Sigma + dw, np.full(m, -dc)])
                    return np.where(self.fixed, 0.0, sol[:nv]), sol[nv:], dw
            if dw >= 1e20:
                break
            if dw == 0.0:
                dw = 1e-4 if dw_last == 0.0 else max(1e-20, dw_last / 3.0)
            else:
                dw *= 100.0 if dw_last == 0.0 else 8.0
        raise RuntimeError("KKT regularisation failed")

    def _resolve_kkt(self, r1, r2, v=None, lam=None):
        nv = self.nv
        # the callbacks evaluate G~ with every new point (pc_engine.hip::ensure_fcG), the trial points included: the
        # refinement's matrix-vector products need G~ / H~ of the point the factors belong to again
        self.engine.evaluate_resident(v[:self.n], self.sf, self.sc * lam, want_grad=False)
        with np.errstate(all="ignore"):
            sol = self._refined_solve(np.concatenate([np.where(self.fixed, 0.0, r1), r2]), self._last_dvec, True)
        return np.where(self.fixed, 0.0, sol[:nv]), sol[nv:]

    def _ls_multipliers(self, J, gz):
        lam = np.zeros(self.m)
        nv, m = self.nv, self.m
        dvec = np.concatenate([np.ones(nv), np.full(m, -1e-10)])
        self._factor(dvec, False)
        with np.errstate(all="ignore"):
            sol = self._refined_solve(np.concatenate([np.where(self.fixed, 0.0, -gz), np.zeros(m)]), dvec, False)
        if m and np.all(np.isfinite(sol)) and np.max(np.abs(sol[nv:])) <= 1e3:
            lam = sol[nv:]
        return lam

    def _gn_step(self, J, c):
        nv, m = self.nv, self.m
        dvec = np.concatenate([np.ones(nv), np.full(m, -1e-10)])
        self._factor(dvec, False)
        sol = self._refined_solve(np.concatenate([np.zeros(nv), -c]), dvec, False)
        return np.where(self.fixed, 0.0, sol[:nv])

    def solve(self, x0) -> IpmResult:
        self.engine.set_prefetch_jac(False)     # G~ is consumed where it is produced
        try:
            res = super().solve(x0)
        finally:
            self.engine.set_prefetch_jac(True)
        res.evaluations["gpu_seconds"] = dict(self.times)
        if self.kkt is not None:
            self.kkt.close()
            self.kkt = None
        return res


class _IpmDesc(__import__("ctypes").Structure):
    import ctypes as _C
    _fields_ = [("n", _C.c_int64), ("m", _C.c_int64), ("ns", _C.c_int64), ("ineq_rows", _C.c_void_p),
                ("vl", _C.c_void_p), ("vu", _C.c_void_p), ("hasl", _C.c_void_p), ("hasu", _C.c_void_p), ("fixed", _C.c_void_p),
                ("row_scale", _C.c_void_p), ("rhs_c", _C.c_void_p), ("obj_scale", _C.c_double)]


class ResidentInteriorPointSolver(GpuInteriorPointSolver):
    """The same algorithm with the ITERATION on the device as well (``csrc/pc_ipm.hpp``): v, lambda, z, the step and
    every right-hand side stay in device memory; per iteration this loop makes four or five C calls -- error measures,
    Newton step (Hessian, assembly, factorisation with the regularisation loop, refined solve, step limits), one call
    per line-search trial point, acceptance -- and reads a handful of scalars.  The filter, the barrier update and the
    termination tests are the scalar logic of :meth:`InteriorPointSolver.solve`, unchanged; the rare restoration branch
    pulls the state to the host, runs the parent's code and pushes it back.  The reference's boundary: one call into
    IPOPT per solve (pycollo/backend.py:1807-1827)."""

    def solve(self, x0) -> IpmResult:
        import ctypes as C
        self.engine.set_prefetch_jac(False)
        try:
            return self._solve_resident(x0, C)
        finally:
            self.engine.set_prefetch_jac(True)
            if getattr(self, "_ipm", None):
                self.engine._lib.pc_ipm_destroy(self._ipm)
                self._ipm = None
            if self.kkt is not None:
                self.kkt.close()
                self.kkt = None

    def _solve_resident(self, x0, C) -> IpmResult:
        t_start = time.perf_counter()
        lib = self.engine._lib
        for name in ("pc_ipm_create", "pc_ipm_set_state", "pc_ipm_get_state", "pc_ipm_eval_point", "pc_ipm_errors",
                     "pc_ipm_newton", "pc_ipm_trial", "pc_ipm_accept", "pc_ipm_soc", "pc_ipm_soc_restore"):
            getattr(lib, name).restype = C.c_int
        lib.pc_ipm_destroy.restype = None
        lib.pc_ipm_destroy.argtypes = [C.c_void_p]
        lib.pc_ipm_create.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.pc_ipm_set_state.argtypes = [C.c_void_p] * 5
        lib.pc_ipm_get_state.argtypes = [C.c_void_p] * 7
        lib.pc_ipm_eval_point.argtypes = [C.c_void_p, C.c_void_p]
        lib.pc_ipm_errors.argtypes = [C.c_void_p, C.c_void_p]
        lib.pc_ipm_newton.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_void_p]
        lib.pc_ipm_trial.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_void_p]
        lib.pc_ipm_accept.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double]
        lib.pc_ipm_soc.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_double, C.c_double, C.c_void_p]
        lib.pc_ipm_soc_restore.argtypes = [C.c_void_p, C.c_double, C.c_double]

        def check(ok):
            if not ok:
                raise RuntimeError(lib.pc_last_error().decode())

        n, nv, m = self.n, self.nv, self.m
        x0 = np.asarray(x0, float)
        # ---- set-up on the host, exactly as InteriorPointSolver.solve: gradient-based scaling, slacks, interior push ----
        self.sf, self.sc = 1.0, np.ones(m)
        push = (1e-3, 1e-3) if self.warm_start else (1e-2, 1e-2)
        x0c = self._push_interior(np.concatenate([x0, np.zeros(self.ns)]), *push)[:n]
        x0c[self.fixed[:n]] = self.vl[:n][self.fixed[:n]]
        g0 = np.nan_to_num(np.asarray(self.p.gradient(x0c), float), nan=0.0, posinf=0.0, neginf=0.0)
        gmax = float(np.max(np.abs(g0))) if n else 0.0
        if gmax > 100.0 and self.gradient_scaling:
            self.sf = max(100.0 / gmax, 1e-8)
        if m and self.gradient_scaling:
            jv = np.abs(np.nan_to_num(np.asarray(self.p.jacobian(x0c), float), nan=0.0, posinf=0.0, neginf=0.0))
            rowmax = np.zeros(m)
            np.maximum.at(rowmax, self.jr, jv)
            big = rowmax > 100.0
            self.sc[big] = np.maximum(100.0 / rowmax[big], 1e-8)
        if self.ns:
            self.vl[n:] = np.where(self.cl[self.ineq] > -INF, self.sc[self.ineq] * self.cl[self.ineq], self.cl[self.ineq])
            self.vu[n:] = np.where(self.cu[self.ineq] < INF, self.sc[self.ineq] * self.cu[self.ineq], self.cu[self.ineq])
        v = np.concatenate([x0, np.zeros(self.ns)])
        if self.ns:
            v[n:] = (self.sc * np.asarray(self.p.constraints(x0), float))[self.ineq]
        v = self._push_interior(v, 1e-3, 1e-3) if self.warm_start else self._push_interior(v)
        v[self.fixed] = self.vl[self.fixed]
        mu = self.mu_init
        zl = np.where(self.hasl, 1.0, 0.0)
        zu = np.where(self.hasu, 1.0, 0.0)
        f, g = self._f(v[:n]), np.concatenate([self._g(v[:n]), np.zeros(self.ns)])
        c, J = self._c(v), self._J(v[:n])
        lam = self._ls_multipliers(J, g - zl + zu)
        # ---- the device state ----
        kkt = self._ensure_kkt()
        keep = [np.ascontiguousarray(self.ineq, dtype=np.int64), np.ascontiguousarray(self.vl), np.ascontiguousarray(self.vu),
                np.ascontiguousarray(self.hasl, dtype=np.uint8), np.ascontiguousarray(self.hasu, dtype=np.uint8),
                np.ascontiguousarray(self.fixed, dtype=np.uint8), np.ascontiguousarray(self.sc), np.ascontiguousarray(self.rhs_c)]
        d = _IpmDesc(n, m, self.ns, *[a.ctypes.data for a in keep], float(self.sf))
        handle = C.c_void_p()
        check(lib.pc_ipm_create(self.engine._h, kkt._h, C.byref(d), C.byref(handle)))
        self._ipm = handle

        def push_state():
            check(lib.pc_ipm_set_state(handle, *(np.ascontiguousarray(a, dtype=np.float64).ctypes.data for a in (v, lam, zl, zu))))

        def pull_state():
            out = [np.empty(nv), np.empty(m), np.empty(nv), np.empty(nv), np.empty(m), np.empty(nv)]
            check(lib.pc_ipm_get_state(handle, *(a.ctypes.data for a in out)))
            return out

        r3, r8, r10 = np.empty(3), np.empty(8), np.empty(10)
        push_state()
        check(lib.pc_ipm_eval_point(handle, r3.ctypes.data))
        f, theta = float(r3[0]), float(r3[1])
        filt: list[tuple[float, float]] = []
        theta0 = theta
        theta_max, theta_min = 1e4 * max(1.0, theta0), 1e-4 * max(1.0, theta0)
        dw_last = 0.0
        status, hist = "max_iter", []
        k_eps, k_mu, th_mu, s_max, g_th, g_phi, eta = 10.0, 0.2, 1.5, 100.0, 1e-5, 1e-8, 1e-4
        accept_count = restarts = it = 0
        last_alpha = last_amax = 0.0
        last_tag = ""
        inf_pr = inf_du = np.inf
        nz = max(1, int(self.hasl.sum() + self.hasu.sum()))
        any_l, any_u = bool(self.hasl.any()), bool(self.hasu.any())

        def errors(e, mu_):
            sd = max(s_max, (e[7] + e[8] + e[9]) / (m + nz)) / s_max
            scz = max(s_max, (e[8] + e[9]) / nz) / s_max
            comp = 0.0
            if any_l:
                comp = max(comp, abs(e[3] - mu_), abs(e[4] - mu_))
            if any_u:
                comp = max(comp, abs(e[5] - mu_), abs(e[6] - mu_))
            e_du = float(e[0]) if self.nf else 0.0
            e_pr = float(e[1]) if m else 0.0
            return max(e_du / sd, e_pr, comp / scz), e_pr, e_du

        phase = {"setup": time.perf_counter() - t_start, "errors": 0.0, "line_search": 0.0, "newton": 0.0}
        for it in range(self.max_iter + 1):
            t_ph = time.perf_counter()
            check(lib.pc_ipm_errors(handle, r10.ctypes.data))
            e = r10.copy()
            theta = float(e[2])
            e0, inf_pr, inf_du = errors(e, 0.0)
            phase["errors"] += time.perf_counter() - t_ph
            hist.append((it, f, inf_pr, inf_du, mu))
            if self.verbose:
                print(f"{it:4d}  f {f: .8e}  inf_pr {inf_pr:.2e}  inf_du {inf_du:.2e}  lg(mu) {np.log10(mu):5.1f}  dw {dw_last:.1e}"
                      f"  alpha {last_alpha:.2e} (max {last_amax:.2e}){last_tag}")
            if e0 <= self.tol:
                status = "optimal"
                break
            accept_count = accept_count + 1 if e0 <= self.acceptable_tol else 0
            if accept_count >= 15:
                status = "acceptable"
                break
            if it == self.max_iter:
                break
            while errors(e, mu)[0] <= k_eps * mu and mu > self.tol / 10:
                mu = max(self.tol / 10, min(k_mu * mu, mu ** th_mu))
                filt = []
            tau = max(0.99, 1.0 - mu)
            t_kkt = time.perf_counter()
            self.counts["hessian"] += 1
            check(lib.pc_ipm_newton(handle, mu, tau, dw_last, r8.ctypes.data))
            dt = time.perf_counter() - t_kkt
            self.kkt_seconds += dt
            phase["newton"] += dt
            self.counts["factorisations"] += int(r8[5])
            self.counts["kkt_solves"] = self.counts.get("kkt_solves", 0) + int(r8[6])
            self.counts["refined_solves"] = self.counts.get("refined_solves", 0) + 1
            if r8[0] < 0.0:
                status = "kkt_failure"
                break
            dw_last, a_max, a_z, dphi = float(r8[0]), float(r8[1]), float(r8[2]), float(r8[3])
            phi = f + float(r8[4])
            # filter line search: the scalar logic of InteriorPointSolver.solve, one device call per trial point
            t_ph = time.perf_counter()
            alpha, accepted = a_max, False
            ft = th_t = phi_t = np.nan

            def acceptable(alpha_, ft_, th_t_, phi_t_):
                # (InteriorPointSolver.solve's test; the second value says the filter is to be augmented)
                if not (np.isfinite(ft_) and np.isfinite(th_t_) and np.isfinite(phi_t_) and th_t_ <= theta_max):
                    return False, False
                if any(th_t_ >= th_f and phi_t_ >= ph_f for th_f, ph_f in filt):
                    return False, False
                if dphi < 0 and alpha_ * (-dphi) ** 2.3 > theta ** 1.1 and theta <= theta_min:
                    return phi_t_ <= phi + eta * alpha_ * dphi, False
                ok = th_t_ <= (1 - g_th) * theta or phi_t_ <= phi - g_phi * theta
                return ok, ok

            def trial(a):
                self.counts["objective"] += 1
                self.counts["constraints"] += 1
                check(lib.pc_ipm_trial(handle, a, mu, r3.ctypes.data))
                return float(r3[0]), float(r3[1]), float(r3[0]) + float(r3[2])

            first = True
            while alpha > 1e-12:
                ft, th_t, phi_t = trial(alpha)
                accepted, augment = acceptable(alpha, ft, th_t, phi_t)
                if self.verbose >= 2 and first:
                    print(f"      first trial: alpha {alpha:.2e} theta {theta:.3e} -> {th_t:.3e}  phi {phi:.10e} -> {phi_t:.10e}  dphi {dphi:.3e} accepted {accepted}")
                if accepted:
                    if augment:
                        filt.append(((1 - g_th) * theta, phi - g_phi * theta))
                    break
                if first and self.second_order_correction and np.isfinite(th_t) and th_t >= theta:
                    # second-order correction on the device (pc_ipm_soc): up to four corrected steps from the same factors
                    th_old, a_soc, r8s = theta, alpha, np.empty(8)
                    for p_soc in range(4):
                        self.counts["second_order_corrections"] = self.counts.get("second_order_corrections", 0) + 1
                        check(lib.pc_ipm_soc(handle, a_soc, 1 if p_soc == 0 else 0, mu, tau, r8s.ctypes.data))
                        self.counts["kkt_solves"] = self.counts.get("kkt_solves", 0) + int(r8s[6])
                        if r8s[7] != 0.0:
                            break
                        a_soc = float(r8s[1])
                        fs, th_s, phi_s = trial(a_soc)
                        ok_s, augment = acceptable(alpha, fs, th_s, phi_s)
                        if self.verbose >= 2:
                            print(f"      correction: alpha {a_soc:.2e} theta -> {th_s:.3e}  phi -> {phi_s:.10e}  accepted {ok_s}")
                        if ok_s:
                            if augment:
                                filt.append(((1 - g_th) * theta, phi - g_phi * theta))
                            accepted, alpha, a_z, ft, th_t, phi_t = True, a_soc, float(r8s[2]), fs, th_s, phi_s
                            self.counts["second_order_steps"] = self.counts.get("second_order_steps", 0) + 1
                            break
                        if not np.isfinite(th_s) or th_s > 0.99 * th_old:
                            break
                        th_old = th_s
                    if accepted:
                        break
                    check(lib.pc_ipm_soc_restore(handle, mu, tau))
                first = False
                alpha *= 0.5
            phase["line_search"] += time.perf_counter() - t_ph
            if not accepted and e0 <= self.acceptable_tol:
                status = "acceptable"
                self.counts["acceptable_after_failed_line_search"] = 1
                break
            if not accepted:
                # restoration (rare): the state comes to the host, the parent's Gauss-Newton restoration / barrier restart
                # runs on host vectors through the same GPU factorisation, the result goes back
                v, lam, zl, zu, c, g = pull_state()
                J = ("resident", 0)
                filt.append(((1 - g_th) * theta, phi - g_phi * theta))
                vr, cr_, Jr, ok_r = v.copy(), c, J, False
                for _ in range(30):
                    th_r = float(np.sum(np.abs(cr_)))
                    self._J(vr[:n])              # G~ at the restoration point (the Newton step left the Hessian's launch behind it)
                    try:
                        dr = self._gn_step(Jr, cr_)
                    except RuntimeError:
                        break
                    if not np.all(np.isfinite(dr)):
                        break
                    a = self._alpha_max(vr, dr, tau)
                    moved = False
                    while a > 1e-10:
                        vt = vr + a * dr
                        with np.errstate(all="ignore"):
                            ct = self._c(vt)
                        if np.all(np.isfinite(ct)) and float(np.sum(np.abs(ct))) < (1 - 1e-4 * a) * th_r:
                            vr, cr_, moved = vt, ct, True
                            break
                        a *= 0.5
                    if not moved:
                        break
                    th_t = float(np.sum(np.abs(cr_)))
                    if th_t <= 0.9 * theta:
                        with np.errstate(all="ignore"):
                            ft = self._f(vr[:n])
                            phi_t = self._barrier(vr, ft, mu)
                        if np.isfinite(phi_t) and not any(th_t >= th_f and phi_t >= ph_f for th_f, ph_f in filt):
                            ok_r = True
                            break
                if not ok_r:
                    restarts += 1
                    if restarts > 3:
                        status = "restoration_failed"
                        break
                    if float(np.sum(np.abs(cr_))) < theta:
                        v = vr
                    mu = min(self.mu_init, max(100.0 * mu, 1e-6))
                    filt = []
                    v = self._push_interior(v, min(1e-2, mu), min(1e-2, mu))
                    g = np.concatenate([self._g(v[:n]), np.zeros(self.ns)])
                    self._J(v[:n])
                    dlv, duv = np.where(self.hasl, v - self.vl, 1.0), np.where(self.hasu, self.vu - v, 1.0)
                    zl = np.where(self.hasl, mu / dlv, 0.0)
                    zu = np.where(self.hasu, mu / duv, 0.0)
                    lam = self._ls_multipliers(J, g - zl + zu)
                    last_alpha, last_amax, last_tag = 0.0, a_max, " r"
                else:
                    v = vr
                    last_alpha, last_amax, last_tag = 0.0, a_max, " R"
                    g = np.concatenate([self._g(v[:n]), np.zeros(self.ns)])
                    self._J(v[:n])
                    lam = self._ls_multipliers(J, g - zl + zu)
                push_state()
                check(lib.pc_ipm_eval_point(handle, r3.ctypes.data))
                f = float(r3[0])
                continue
            last_alpha, last_amax, last_tag = alpha, a_max, ""
            self.counts["gradient"] += 1
            self.counts["jacobian"] += 1
            check(lib.pc_ipm_accept(handle, alpha, a_z, mu))
            f = ft
        v, lam, zl, zu, c, g = pull_state()
        ev = dict(self.counts)
        ev["barrier_restarts"] = restarts
        ev["kkt_seconds"] = self.kkt_seconds
        ev["phase_seconds"] = {k: round(val, 6) for k, val in phase.items()}
        ev["gpu_seconds"] = dict(self.times)
        ev["resident_iteration"] = True
        return IpmResult(x=v[:n].copy(), lam=(self.sc * lam / self.sf).copy(), objective=f / self.sf, status=status, iterations=it,
                         inf_pr=inf_pr, inf_du=inf_du, mu=mu, seconds=time.perf_counter() - t_start, evaluations=ev, history=hist,
                         zl=(zl[:n] / self.sf).copy(), zu=(zu[:n] / self.sf).copy())


def solve_nlp(problem_obj, x0, lb, ub, cl, cu, **options) -> IpmResult:
    """``ipopt.problem(n, m, problem_obj, lb, ub, cl, cu).solve(x0)`` with the stand-in solver."""
    return InteriorPointSolver(problem_obj, len(x0), len(cl), lb, ub, cl, cu, **options).solve(x0)
