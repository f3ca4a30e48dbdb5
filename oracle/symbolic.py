"""Fully symbolic assembly of the transcribed NLP on tiny meshes (TEST INFRASTRUCTURE ONLY).

Builds ``c~(x~)`` as one SymPy vector exactly the way the live reference builds it as ``ca.SX``
(pycollo/backend.py:1513-1672: per-node substitution of the OCP equations, ``A y + 1/2 (tF - t0) I f``,
``W p``, ``q - 1/2 (tF - t0) W.g``, endpoint rows) and then differentiates the *assembled* vector --
the operation ``ca.jacobian(c_iter, x_var_iter)`` (backend.py:1676) and, inside ``nlpsol``,
``hessian(sigma f + lambda.g)`` perform.  Nothing here shares code with ``ref_numpy``'s block
formulas; it is the referee for their values and for the structural (row, col) patterns.
"""
from __future__ import annotations

import numpy as np
import sympy as sym

from .ref_numpy import OracleMesh, _bnds, _pair, _same, _subs_all


def assemble(prob, tables, V_ocp, r_ocp, W_ocp, w_J, meshes=None):
    """Return (x symbols, J expr, c exprs list) in scaled variables."""
    xs = []
    c = []
    s_user = list(prob.parameter_variables)
    s_b = _bnds(s_user, prob.bounds.parameter_variables)
    s_keep = [not _same(*b) for b in s_b]
    s_syms = [s for s, k in zip(s_user, s_keep) if k]
    s_const = {s: 0.5 * sum(b) for s, b, k in zip(s_user, s_b, s_keep) if not k}
    n_s = len(s_syms)
    # count OCP variables first to find where s lives in V_ocp
    counts = []
    for ph in prob.phases:
        yk = [not _same(*b) for b in _bnds(list(ph.state_variables), ph.bounds.state_variables)]
        uk = [not _same(*b) for b in _bnds(list(ph.control_variables), ph.bounds.control_variables)]
        qk = [not _same(*b) for b in _bnds(list(ph.integral_variables), ph.bounds.integral_variables)]
        tk = [not _same(*_pair(ph.bounds.initial_time)), not _same(*_pair(ph.bounds.final_time))]
        counts.append((yk, uk, qk, tk))
    ocp_s = sum(sum(yk) + sum(uk) + sum(qk) + sum(tk) for yk, uk, qk, tk in counts)
    st = [sym.Symbol(f"xs{i}") for i in range(n_s)]
    s_unscaled = {s: V_ocp[ocp_s + i] * st[i] + r_ocp[ocp_s + i] for i, s in enumerate(s_syms)}
    point = dict(s_const)
    point.update(s_unscaled)
    ox = oc = 0
    for ip, ph in enumerate(prob.phases):
        yk, uk, qk, tk = counts[ip]
        ys = [y for y, k in zip(ph.state_variables, yk) if k]
        us = [u for u, k in zip(ph.control_variables, uk) if k]
        y_b = _bnds(list(ph.state_variables), ph.bounds.state_variables)
        u_b = _bnds(list(ph.control_variables), ph.bounds.control_variables)
        q_b = _bnds(list(ph.integral_variables), ph.bounds.integral_variables)
        t_b = [_pair(ph.bounds.initial_time), _pair(ph.bounds.final_time)]
        const = dict(s_const)
        const.update({y: 0.5 * sum(b) for y, b, k in zip(ph.state_variables, y_b, yk) if not k})
        const.update({u: 0.5 * sum(b) for u, b, k in zip(ph.control_variables, u_b, uk) if not k})
        sizes, nodes = meshes[ip] if meshes is not None else ph.mesh.resolved()
        mesh = OracleMesh(tables, sizes, nodes)
        N = mesh.N
        zs = ys + us
        n_y, n_z = len(ys), len(zs)
        zt = [[sym.Symbol(f"x_P{ip}_{j}_{i}") for i in range(N)] for j in range(n_z)]
        qt = [sym.Symbol(f"q_P{ip}_{m}") for m in range(sum(qk))]
        tt = [sym.Symbol(f"t_P{ip}_{e}") for e in range(sum(tk))]
        for row in zt:
            xs.extend(row)
        xs.extend(qt)
        xs.extend(tt)
        zu = [[V_ocp[ox + j] * zt[j][i] + r_ocp[ox + j] for i in range(N)] for j in range(n_z)]
        qu = [V_ocp[ox + n_z + m] * qt[m] + r_ocp[ox + n_z + m] for m in range(len(qt))]
        to = ox + n_z + len(qt)
        t, j = [], 0
        for e in (0, 1):
            if tk[e]:
                t.append(V_ocp[to + j] * tt[j] + r_ocp[to + j]); j += 1
            else:
                t.append(0.5 * sum(t_b[e]))
        stretch = 0.5 * (t[1] - t[0])
        aux = dict(prob.auxiliary_data); aux.update(ph.auxiliary_data)
        low = lambda e: _subs_all(_subs_all(e, aux), const)
        f = [low(e) for e, k in zip(ph.state_equations, yk) if k]
        p = [low(e) for e in ph.path_constraints]
        g = [low(e) for e, k in zip(ph.integrand_functions, qk) if k]

        # q, t0, tF are the same symbols at every node (backend.py:1526-1539 maps y and u only)
        q_all, jq = {}, 0
        for i, k in enumerate(qk):
            q_all[ph.integral_variables[i]] = qu[jq] if k else 0.5 * sum(q_b[i])
            jq += 1 if k else 0
        glob = dict(s_unscaled)
        glob.update(q_all)
        glob[ph.initial_time_variable] = t[0]
        glob[ph.final_time_variable] = t[1]

        def at_nodes(e):
            out = []
            for i in range(N):
                m = {zs[j]: zu[j][i] for j in range(n_z)}
                m.update(glob)
                out.append(e.subs(m, simultaneous=True))
            return out

        Am, Im = mesh.A_mat.toarray(), mesh.I_mat.toarray()
        Ims = mesh.I_mat.tocoo()
        Ams = mesh.A_mat.tocoo()
        for a in range(n_y):
            fa = at_nodes(f[a])
            rows = [0] * (N - 1)
            for r_, c_, v_ in zip(Ams.row, Ams.col, Ams.data):      # only structural entries
                rows[r_] = rows[r_] + v_ * zu[a][c_]
            for r_, c_, v_ in zip(Ims.row, Ims.col, Ims.data):
                rows[r_] = rows[r_] + stretch * v_ * fa[c_]
            c.extend(W_ocp[oc + a] * e for e in rows)
        for m_ in range(len(p)):
            c.extend(W_ocp[oc + n_y + m_] * e for e in at_nodes(p[m_]))
        for m_ in range(len(g)):
            gm = at_nodes(g[m_])
            c.append(W_ocp[oc + n_y + len(p) + m_] * (qu[m_] - stretch * sum(mesh.w[i] * gm[i] for i in range(N))))
        # point symbols
        j = 0
        for i, (y, k) in enumerate(zip(ph.state_variables, yk)):
            if k:
                point[ph.initial_state_variables[i]] = zu[j][0]
                point[ph.final_state_variables[i]] = zu[j][N - 1]
                j += 1
            else:
                point[ph.initial_state_variables[i]] = point[ph.final_state_variables[i]] = const[y]
        j = 0
        for i, k in enumerate(qk):
            if k:
                point[ph.integral_variables[i]] = qu[j]; j += 1
            else:
                point[ph.integral_variables[i]] = 0.5 * sum(q_b[i])
        point[ph.initial_time_variable] = t[0]
        point[ph.final_time_variable] = t[1]
        ox += n_z + len(qt) + len(tt)
        oc += n_y + len(p) + len(g)
    xs.extend(st)
    low_pt = lambda e: _subs_all(e, dict(prob.auxiliary_data)).subs(point, simultaneous=True)
    J = w_J * low_pt(prob.objective_function)
    for r_, e in enumerate(prob.endpoint_constraints):
        c.append(W_ocp[oc + r_] * low_pt(e))
    return xs, J, c


def jacobian_triplets(xs, c, x0):
    """Structural (rows, cols) and numeric values of dc/dx at x0, row-major ascending columns."""
    sub = dict(zip(xs, x0))
    index = {s: i for i, s in enumerate(xs)}
    rows, cols, vals = [], [], []
    for r, e in enumerate(c):
        fs = sorted((index[s] for s in e.free_symbols if s in index))
        for ci in fs:
            d = sym.diff(e, xs[ci])
            if d != 0:
                rows.append(r); cols.append(ci); vals.append(float(d.subs(sub)))
    return np.array(rows), np.array(cols), np.array(vals)


def hessian_triplets(xs, J, c, x0, sigma, lam):
    """Lower-triangular structural (rows, cols) and values of d2(sigma J + lam.c)/dx2 at x0.

    The structure is the union over terms of entries that are not identically zero *with symbolic
    multipliers* (CasADi differentiates sigma*f + lam'g with symbolic sigma, lam)."""
    sub = dict(zip(xs, x0))
    index = {s: i for i, s in enumerate(xs)}
    acc: dict[tuple[int, int], float] = {}
    terms = [(sigma, J)] + [(lam[i], e) for i, e in enumerate(c)]
    for wgt, e in terms:
        fs = sorted((index[s] for s in e.free_symbols if s in index))
        for a_i, ra in enumerate(fs):
            d = sym.diff(e, xs[ra])
            if d == 0:
                continue
            for ca in fs[:a_i + 1]:
                d2 = sym.diff(d, xs[ca])
                if d2 != 0:
                    acc[(ra, ca)] = acc.get((ra, ca), 0.0) + wgt * float(d2.subs(sub))
    keys = sorted(acc)
    return (np.array([k[0] for k in keys]), np.array([k[1] for k in keys]), np.array([acc[k] for k in keys]))
