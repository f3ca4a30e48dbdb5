"""An NLP solved by several ranks: the interior-point loop of ``ipm.GpuInteriorPointSolver`` over the section-sharded
evaluation (``sharding.ShardedNlp.evaluate_local_device``) and the KKT factorisation cut across ranks
(``kkt_sharded.ShardedKkt``).

The reference solves on one process (pycollo/backend.py:1807-1827: one ``ca.nlpsol`` call; IPOPT + MUMPS).  Here every
rank runs the same interior-point iteration on **replicated vectors** (x~, slacks, multipliers, steps: a few hundred KB)
while the **matrices never travel**: a rank evaluates G~ / H~ for its own section range, eliminates its own leaves and
chain segments from them, and what crosses between ranks per iteration is

  * the per-tile partial sums of the evaluation (a few doubles per tile) and the rank's rows of c~ (a vector),
  * per factorisation the reduced border system (nb_red^2 doubles) and two pivot counts,
  * per back-substitution the reduced right-hand side (nb_red doubles) and the solution vector,
  * per matrix-vector product (residuals of the refinement, J^T lambda) one vector.

All decisions of the loop (filter, step sizes, regularisation, termination) are taken on reduced values that are
bit-identical on every rank, so the ranks stay in step without a coordinator.  Once per solve, IPOPT's gradient-based
scaling looks at every Jacobian entry (ipm.InteriorPointSolver.solve): that one evaluation gathers G~.
"""
from __future__ import annotations

import os

import numpy as np

from .ipm import GpuInteriorPointSolver
from .kkt_sharded import ShardedKkt, ShardedKktPlan
from .sharding import SegmentExchange, _SubPlan


class _ShardedEngineView:
    """What ``GpuInteriorPointSolver`` asks of an engine, answered by one rank of a ``ShardedNlp``."""

    def __init__(self, sh, poison: bool = False):
        import torch
        self.torch = torch
        self.sh = sh
        eng = sh.engine
        self.num_x, self.num_c, self.device = eng.num_x, eng.num_c, eng.device
        dev = sh.buf.device
        self.dx = torch.zeros(self.num_x, dtype=torch.float64, device=dev)
        self.dlam = torch.zeros(self.num_c, dtype=torch.float64, device=dev)
        # the rank's rows of c~ go to every rank: the iteration's vectors are replicated
        self.c_exchange = SegmentExchange(_SubPlan(sh.world, [[(a, b) for a, b in seg if b <= sh.plan.num_c] for seg in sh.plan.segments]),
                                          sh.rank, dev, sh.exchange.group)
        self.poison = poison          # tests: NaN wherever this rank's tiles and the tail do not write
        self.evaluations = 0

    def _evaluate(self, x, obj_factor, lam, want_grad):
        torch, sh = self.torch, self.sh
        ts = sh.tstream
        self.evaluations += 1
        with torch.cuda.stream(ts):
            self.dx.copy_(torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64)))
            if lam is None:
                self.dlam.zero_()
            else:
                self.dlam.copy_(torch.from_numpy(np.ascontiguousarray(lam, dtype=np.float64)))
            eng = sh.engine
            eng.launch_bulk_only(self.dx, self.dlam, sh.c, sh.G, sh.H, ts.cuda_stream)
            if sh.world > 1:
                sh.exchange_partials.run(sh.buf)
            f, grad = eng.launch_tail_objective(self.dx, obj_factor, self.dlam, sh.c, sh.G, sh.H, ts.cuda_stream, want_grad)
            if sh.world > 1:
                self.c_exchange.run(sh.buf)
            c = sh.c.cpu().numpy()
        return f, grad, c

    def evaluate_resident(self, x, obj_factor=1.0, lagrange=None, want_grad=True):
        return self._evaluate(x, obj_factor, lagrange, want_grad)

    def gathered_jacobian(self, x):
        """G~ complete on every rank (one all-gather of the rows): the scaling pass at the start of a solve."""
        torch, sh = self.torch, self.sh
        ts = sh.tstream
        with torch.cuda.stream(ts):
            self.dx.copy_(torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64)))
            self.dlam.zero_()
            sh.evaluate_all_device(self.dx, 1.0, self.dlam, ts)
            G = sh.G.cpu().numpy()
            if self.poison:
                sh.buf.fill_(float("nan"))
        return G

    def set_prefetch_jac(self, on):
        pass


class ShardedGpuProblem:
    """The cyipopt-protocol object (pycollo/nlp.py:47-76 method names) of one rank."""

    def __init__(self, sh, poison: bool = False):
        self.sh = sh
        self.engine = _ShardedEngineView(sh, poison)
        self.n, self.m = sh.num_x, sh.num_c

    def objective(self, x):
        return self.engine._evaluate(x, 1.0, None, False)[0]

    def gradient(self, x):
        return self.engine._evaluate(x, 1.0, None, True)[1]

    def constraints(self, x):
        return self.engine._evaluate(x, 1.0, None, False)[2]

    def jacobian(self, x):
        return self.engine.gathered_jacobian(x)

    def jacobianstructure(self):
        return self.sh.engine.evaluate_G_structure()

    def hessianstructure(self):
        return self.sh.engine.evaluate_H_structure()


class _ShardedKktAdapter:
    """``GpuKkt``'s factor / matvec / solve_refined over this rank's part of the factorisation and the reductions."""

    def __init__(self, sh, ineq, fixed, sc):
        import time
        t0 = time.perf_counter()
        self.plan = ShardedKktPlan(sh.engine, ineq, fixed, sc, sh.plan)
        self.seconds_tables = time.perf_counter() - t0
        self.rank = sh.rank
        self.sk = ShardedKkt(sh.engine, self.plan, [sh.rank], d_jac=sh.G.data_ptr(), d_hess=sh.H.data_ptr(),
                             group=sh.exchange.group, distributed=sh.world > 1)
        self.nu = self.plan.nu
        self.resid_tol = float(os.environ.get("PYCOLLO_AMD_KKT_RESID_TOL", "1e-12"))

    def factor(self, dvec, use_hess=True):
        return self.sk.factor(dvec, use_hess)

    def solve(self, rhs):
        return self.sk.solve(rhs)

    def matvec(self, dvec, x, use_hess=True):
        """K x: every entry of the matrix sits in exactly one rank's tables, the diagonal with the unknown's owner."""
        y = np.zeros(self.nu)
        x = np.asarray(x, float)
        for r, h in self.sk.handles.items():
            R = self.plan.ranks[r]
            y[R.univ] += h.matvec(self.plan.local_vector(r, dvec), x[R.univ], use_hess)
        red = self.sk._reduce()
        return y if red is None else red(y)

    def solve_refined(self, rhs, dvec_true, use_hess=True, max_steps=3):
        """The rule of ``pc_kkt_solve_refined``: a correction is kept while it halves the residual and stays finite; stop
        once the residual is below ``resid_tol`` of the right-hand side."""
        rhs = np.asarray(rhs, float)
        sol = self.solve(rhs)
        solves = 1
        if max_steps <= 0:
            return sol, solves
        res = rhs - self.matvec(dvec_true, sol, use_hess)
        nres, nrhs = float(res @ res), float(rhs @ rhs)
        for _ in range(max_steps):
            if nres <= self.resid_tol ** 2 * nrhs:
                break
            trial = sol + self.solve(res)
            res_t = rhs - self.matvec(dvec_true, trial, use_hess)
            solves += 1
            nt = float(res_t @ res_t)
            if not (np.all(np.isfinite(trial)) and np.all(np.isfinite(res_t))) or np.sqrt(nt) >= 0.5 * np.sqrt(nres):
                break
            sol, res, nres = trial, res_t, nt
        return sol, solves

    def close(self):
        self.sk.close()


class ShardedInteriorPointSolver(GpuInteriorPointSolver):
    """``GpuInteriorPointSolver`` on one rank of a sharded NLP (``problem_obj``: a ``ShardedGpuProblem``)."""

    def _ensure_kkt(self):
        if self.kkt is None:
            self.kkt = _ShardedKktAdapter(self.p.sh, self.ineq, self.fixed, self.sc)
            self.times["tables"] = self.times.get("tables", 0.0) + self.kkt.seconds_tables
            self.times["tables_host"] = self.kkt.seconds_tables
        return self.kkt


def solve_sharded(iteration, group=None, max_iter: int = 500, tol: float = 1e-8, verbose: int = 0, poison: bool = False, **opts):
    """Solve a mesh iteration's scaled NLP over the ranks of ``group`` (``torch.distributed``; every rank calls this with
    the same ``iteration`` contents).  Returns the ``IpmResult`` (identical on every rank) and the sharded evaluation."""
    from .sharding import ShardedNlp
    sh = ShardedNlp(None, device=iteration.engine.device, group=group, engine=iteration.engine)
    if poison:
        sh.buf.fill_(float("nan"))
    pobj = ShardedGpuProblem(sh, poison)
    solver = ShardedInteriorPointSolver(pobj, pobj.n, pobj.m, iteration.x_bnd_l, iteration.x_bnd_u, iteration.c_bnd_l,
                                        iteration.c_bnd_u, tol=tol, max_iter=max_iter, verbose=verbose, **opts)
    res = solver.solve(iteration.guess_x_tilde)
    res.evaluations["sharded"] = {"world": sh.world, "rank": sh.rank, "evaluations": pobj.engine.evaluations}
    return res, sh
