"""The KKT factorisation cut across ranks (pycollo_amd/kkt_sharded.py), tables only: every rank's local system and the
reduced border system executed with NumPy (oracle/ref_kkt.py) on G~ / H~ values that are NaN wherever the rank's own tile
kernels (or the tail) do not write -- against a general sparse solve of the whole matrix.  No GPU."""
import numpy as np
import pytest
import scipy.sparse.linalg as spla

from oracle.ref_kkt import RefKkt
from pycollo_amd import kkt, kkt_sharded
from pycollo_amd.sharding import ShardPlan
from test_kkt_cpu import kkt_case, reference_matrix

CASES = [("hypersensitive", dict(K=30, order=6), 2), ("hypersensitive", dict(K=64, order=3), 8),
         ("sliding_mass", dict(num_phases=3, K=8, order=4), 3), ("shuttle", dict(K=12, order=4), 2),
         ("free_flying_robot", dict(K=9, order=5), 3), ("time_coupled_transfer", dict(K=9, order=4), 2),
         ("two_phase_transfer", dict(K=6, order=3), 2), ("hypersensitive", dict(K=2, order=3), 2)]


def rank_values(plan, sp, r, G, H):
    """G~ / H~ as rank r has them after its own tiles and the tail: NaN where another rank's tiles write."""
    oG, oH = sp.num_c, sp.num_c + sp.nnz_G
    po = plan.pos_owner
    Gr, Hr = G.copy(), H.copy()
    Gr[(po[oG:oH] >= 0) & (po[oG:oH] != r)] = np.nan
    Hr[(po[oH:] >= 0) & (po[oH:] != r)] = np.nan
    return Gr, Hr


@pytest.mark.parametrize("group", [1, None])
@pytest.mark.parametrize("name,kw,world", CASES)
def test_sharded_elimination_matches_a_general_sparse_solve(built, name, kw, world, group):
    eng, ora, x, lam, ineq, fixed, sc, dvec = kkt_case(name, kw)
    G, H = ora.G(x), ora.H(x, 1.0, lam)
    sp = ShardPlan(eng, world)
    plan = kkt_sharded.ShardedKktPlan(eng, ineq, fixed, sc, sp, group, ends="border")   # (the NumPy oracle eliminates every chain node)
    K = reference_matrix(eng, G, H, ineq, fixed, sc, dvec)
    handles = {r: RefKkt(plan.ranks[r].tables, *rank_values(plan, sp, r, G, H)) for r in range(world)}
    reduced = RefKkt(plan.reduced)
    npos, nneg = kkt_sharded.factor_ranks(plan, handles, reduced, dvec)
    ev = np.linalg.eigvalsh(K.toarray())
    assert (npos, nneg) == (int((ev > 0).sum()), int((ev < 0).sum()))     # the pivot signs still add up to the inertia
    # the same inertia as the unsharded plan's
    R0 = RefKkt(kkt.build_tables(eng, ineq, fixed, sc, group))
    assert R0.factor(G, H, dvec) == (npos, nneg)
    rng = np.random.default_rng(1)
    rhs = rng.normal(size=plan.nu)
    rhs[np.nonzero(fixed)[0]] = 0.0
    xs = kkt_sharded.solve_ranks(plan, handles, reduced, rhs)
    assert np.all(np.isfinite(xs))
    xs = xs + kkt_sharded.solve_ranks(plan, handles, reduced, rhs - K @ xs)
    lu = spla.splu(K)
    xr = lu.solve(rhs)
    xr = xr + lu.solve(rhs - K @ xr)
    assert np.max(np.abs(xs - xr)) <= 1e-9 * np.max(np.abs(xr))
    x0 = R0.solve(rhs)
    x0 = x0 + R0.solve(rhs - K @ x0)
    assert np.max(np.abs(xs - x0)) <= 1e-9 * np.max(np.abs(x0))             # steps of the single-rank factorisation
    eng.close()


def test_a_rank_holds_its_share_and_a_border_that_does_not_grow_with_the_world(built):
    eng, _, _, _, ineq, fixed, sc, _ = kkt_case("hypersensitive", dict(K=256, order=4))
    whole = kkt.build_tables(eng, ineq, fixed, sc)
    nb_local = {}
    for world in (2, 4, 8):
        plan = kkt_sharded.ShardedKktPlan(eng, ineq, fixed, sc, ShardPlan(eng, world), ends="border")
        f = [plan.footprint(r) for r in range(world)]
        assert max(v["local_vals"] for v in f) <= 1.35 * whole.total_vals / world + 4096
        nb_local[world] = max(v["nb_local"] for v in f)
        assert plan.nb_red == whole.nb + (world - 1) * 3                    # y, u and the defect multiplier of a cut node
        # every unknown is reported by exactly one rank
        cover = np.zeros(plan.nu, int)
        for R in plan.ranks:
            cover[R.univ[R.own]] += 1
        assert np.all(cover == 1)
    assert nb_local[8] == nb_local[4] == whole.nb + 6
    # a process builds its own rank's tables only: the same tables
    import dataclasses
    one = kkt_sharded.ShardedKktPlan(eng, ineq, fixed, sc, ShardPlan(eng, 8), only=[3], ends="border")
    assert [R is not None for R in one.ranks] == [r == 3 for r in range(8)]
    for f in dataclasses.fields(one.ranks[3].tables):
        assert np.array_equal(getattr(one.ranks[3].tables, f.name), getattr(plan.ranks[3].tables, f.name)), f.name
    eng.close()


@pytest.mark.parametrize("name,kw,world", CASES)
def test_shared_nodes_as_chain_ends_tables_reduce_to_the_whole_system(built, name, kw, world):
    """``ends="chain"`` (the default): a rank keeps the nodes it shares as the first / last node of its chain segment and
    does not eliminate them.  The tables alone, with dense linear algebra: every rank's matrix (assembled from its own
    tables and its NaN-masked values), its Schur complement onto [exported nodes | its border] scattered into the reduced
    system the way ``ShardedKktPlan.add_border`` scatters the kernels' panels -- the reduced solve and the
    back-substitution must reproduce the whole system's solution, and a rank's border must be the NLP's own."""
    eng, ora, x, lam, ineq, fixed, sc, dvec = kkt_case(name, kw)
    G, H = ora.G(x), ora.H(x, 1.0, lam)
    sp = ShardPlan(eng, world)
    plan = kkt_sharded.ShardedKktPlan(eng, ineq, fixed, sc, sp)
    whole = kkt.build_tables(eng, ineq, fixed, sc)
    K = reference_matrix(eng, G, H, ineq, fixed, sc, dvec)
    rhs = np.random.default_rng(1).normal(size=plan.nu)
    rhs[np.nonzero(fixed)[0]] = 0.0
    B = np.zeros((plan.nb_red, plan.nb_red))
    rb = np.zeros(plan.nb_red)
    parts = []
    for r, R in enumerate(plan.ranks):
        T = R.tables
        assert T.nb == whole.nb                                      # no cut node in a rank's border
        Kr = RefKkt(T).assemble(*rank_values(plan, sp, r, G, H), plan.local_vector(r, dvec)).toarray()
        assert np.all(np.isfinite(Kr))
        base_chain = int(T.leaf_ptr[-1])
        base_border = base_chain + int(T.chain_ptr[-1])
        keep_l = [T.perm[base_chain + int(T.chain_ptr[c]):base_chain + int(T.chain_ptr[c + 1])] for c, _, _ in kkt.export_shapes(T)]
        keep_l = np.concatenate(keep_l + [T.perm[base_border:]]).astype(np.int64)
        keep_red = np.concatenate(list(R.export_red) + [R.border_red]).astype(np.int64)
        elim = np.setdiff1d(np.arange(T.nu), keep_l)
        rl = plan.local_vector(r, rhs)
        rl[T.fixed.astype(bool)] = 0.0
        Kee, Kek = Kr[np.ix_(elim, elim)], Kr[np.ix_(elim, keep_l)]
        X = np.linalg.solve(Kee, np.column_stack([Kek, rl[elim]])) if len(elim) else np.zeros((0, len(keep_l) + 1))
        B[np.ix_(keep_red, keep_red)] += Kr[np.ix_(keep_l, keep_l)] - Kek.T @ X[:, :-1]
        rb[keep_red] += rl[keep_l] - Kek.T @ X[:, -1]
        parts.append((R, elim, keep_l, keep_red, Kee, Kek, rl))
    xb = np.linalg.solve(B, rb)
    xs = np.zeros(plan.nu)
    for R, elim, keep_l, keep_red, Kee, Kek, rl in parts:
        xl = np.zeros(R.tables.nu)
        xl[keep_l] = xb[keep_red]
        if len(elim):
            xl[elim] = np.linalg.solve(Kee, rl[elim] - Kek @ xb[keep_red])
        xs[R.univ[R.own]] = xl[R.own]
    xr = spla.splu(K).solve(rhs)
    assert np.max(np.abs(xs - xr)) <= 1e-7 * np.max(np.abs(xr))      # (dense solves of an ill-conditioned system, unrefined)
    eng.close()
