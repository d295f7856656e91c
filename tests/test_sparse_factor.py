"""The sparse L D L' (round 5, SURVEY.md section 8 / VERDICT round 4 row h): what the reference hands to CHOLMOD for Schur complements that
are sparse or larger than a dense panel (src/solver_interface.c:319-370, 505-541).

 * The oracle's sparse-storage mode is pinned by its dense mode: the same factor BIT FOR BIT on the same matrix (the up-looking recurrence
   over structural nonzeros only adds exact zeros where the dense loop subtracts l * 0), the same iterates on whole solves.
 * The engine's sparse factor (qpalm_amd/csrc/qpalm_sparse.h) against the oracle in sparse mode -- status, iteration count, refactorisation
   count exact, x / y to 1e-9 -- and against the engine's own dense factor on the same problems.
 * At size (gpu): banded / block-diagonal / arrow QPs with n = 20 000 .. 100 000, beyond any dense panel (a 100 000 x 100 000 panel is
   80 GB), against the oracle in sparse mode and against the KKT conditions computed with scipy; device memory proportional to nnz(L).
"""
import numpy as np
import pytest
import scipy.sparse as sp

import oracle.binding as ob
from qpalm_amd.problems import random_qp, sparse_qp
from qpalm_amd.solver import QpalmBatch

ST = dict(eps_abs=1e-6, eps_rel=1e-6, verbose=0)


def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(1.0, np.max(np.abs(b)))) if a.size else 0.0


def oracle_sparse(p, perm=None, **st):
    """the oracle in its sparse-storage mode; perm: the ordering the ENGINE chose for this QP's factor (QpalmBatch.sparse_perm), so that the
    checker factorises the same P H P' entry for entry"""
    o = ob.OracleQP(*p.args(), settings=ob.default_settings(**st))
    o.set_scalar("sparse_mode", 1)
    if perm is not None and not np.array_equal(perm, np.arange(len(perm))):
        o.set_perm(perm)
    o.solve()
    return o


@pytest.mark.parametrize("kind,n", [("banded", 150), ("blocks", 160), ("arrow", 90), ("random", 70)])
def test_oracle_sparse_mode_is_pinned_by_its_dense_mode(kind, n):
    p = random_qp(n, 2 * n, seed=5, density_A=0.03, density_M=0.02) if kind == "random" else sparse_qp(n, kind, seed=3)
    # (a) the same factorisations in both storages: max_rank_update = 0 makes the dense mode refactorise wherever the sparse mode does
    # (any change of the active set or of sigma), max_iter stops the two oracles in the same state: after the factorisation of
    # Q + I / gamma (second iteration), after the first ones of Q + A' Sigma A + I / gamma, and at the end of the solve
    for iters in (2, 3, 5, 100000):
        st = dict(ST, max_iter=iters, max_rank_update=0)
        od = ob.OracleQP(*p.args(), settings=ob.default_settings(**st))
        os_ = ob.OracleQP(*p.args(), settings=ob.default_settings(**st))
        os_.set_scalar("sparse_mode", 2)        # 2: every change refactorises (1 also updates along elimination-tree paths, tested below)
        od.solve(); os_.solve()
        assert od.counter("n_refactor") == os_.counter("n_refactor") and od.counter("n_rank1") == os_.counter("n_rank1") == 0
        Ld, Dd = od.factor()
        Ls, Ds = os_.factor()
        assert np.array_equal(Dd, Ds) and np.array_equal(Ld, Ls), (kind, iters, float(np.max(np.abs(Ld - Ls))))
        assert np.array_equal(od.vec("d"), os_.vec("d")) and np.array_equal(od.vec("x"), os_.vec("x")) and np.array_equal(od.vec("y"), os_.vec("y"))
        assert int(od.info.iter) == int(os_.info.iter) and od.status_val == os_.status_val
    assert od.status_val == 1 and od.counter("n_refactor") >= 2
    # (b) whole solves under the reference's own rule: the dense mode updates its factor where the sparse mode refactorises -- same iterates to rounding
    od = ob.OracleQP(*p.args(), settings=ob.default_settings(**ST)); od.solve()
    os_ = oracle_sparse(p, **ST)
    assert od.status_val == os_.status_val == 1 and int(od.info.iter) == int(os_.info.iter)
    assert rel(os_.x, od.x) <= 1e-10 and rel(os_.y, od.y) <= 1e-9
    assert os_.counter("n_refactor") >= od.counter("n_refactor")
    if kind == "blocks":    # a forest of small trees: rows entering / leaving are rank-1 updates along their paths, as in the dense mode
        assert 0 < os_.counter("n_rank1") <= od.counter("n_rank1")
    if kind == "banded":    # a chain: walking it per row would cost more than refactorising
        assert os_.counter("n_rank1") == 0


def test_oracle_sparse_path_update_equals_the_dense_rank_update():
    """one rank-1 update / downdate of the same factor in both storages: equal to rounding (the dense loop also passes d_j through
    (d_j alpha) / alpha on the columns off the row's elimination-tree path, which the path form -- like CHOLMOD's updown -- leaves alone)"""
    p = sparse_qp(96, "blocks", seed=5)
    st = dict(ST, max_iter=3)          # the third iteration adds the first rows to the factor of Q + I / gamma
    od = ob.OracleQP(*p.args(), settings=ob.default_settings(**st)); od.solve()
    os_ = ob.OracleQP(*p.args(), settings=ob.default_settings(**st)); os_.set_scalar("sparse_mode", 1); os_.solve()
    assert od.counter("n_rank1") == os_.counter("n_rank1") > 0 and os_.counter("n_refactor") == 0
    Ld, Dd = od.factor()
    Ls, Ds = os_.factor()
    assert np.max(np.abs(Ld - Ls)) <= 1e-14 * max(1.0, np.max(np.abs(Ld))) and np.max(np.abs(Dd - Ds)) <= 1e-14 * np.max(np.abs(Dd))
    assert rel(os_.vec("d"), od.vec("d")) <= 1e-13


@pytest.mark.parametrize("ordering", [0, 1])
@pytest.mark.parametrize("kind,n", [("banded", 90), ("blocks", 96), ("arrow", 60), ("random", 50)])
def test_sparse_factor_against_the_oracle_and_the_dense_factor(ctx, kind, n, ordering):
    """ordering 0: the reference's natural ordering; 1: the engine's nested dissection (the oracle then factorises under the same permutation)"""
    p = random_qp(n, 2 * n, seed=7, density_A=0.04, density_M=0.03) if kind == "random" else sparse_qp(n, kind, seed=11)
    res = {}
    perm = None
    for mode in (0, 1):
        ctx.set_option("sparse_factor", mode)
        ctx.set_option("sparse_ordering", ordering)
        try:
            bt = QpalmBatch(ctx, [p], ctx.default_settings(**ST))
            bt.solve()
            x, y = bt.solution()
            res[mode] = (int(bt.info(0).status_val), int(bt.info(0).iter), x[0].copy(), y[0].copy(), int(bt.stats(0).n_refactor), int(bt.stats(0).n_factor_Q))
            if mode:
                nnzL, nbytes = bt.sparse_info(0)
                assert 0 < nnzL <= n * (n - 1) // 2 and nbytes > 0
                perm, levels = bt.sparse_perm(0)
                assert np.array_equal(np.sort(perm), np.arange(p.n)) and 1 <= levels <= p.n
                if ordering == 0:
                    assert np.array_equal(perm, np.arange(p.n))
                elif kind == "banded":
                    assert levels < p.n // 2        # the chain of 90 columns became a tree
            bt.close()
        finally:
            ctx.set_option("sparse_factor", -1)
            ctx.set_option("sparse_ordering", -1)
    o = oracle_sparse(p, perm, **ST)
    st_, it_, x, y, nref, nfq = res[1]
    assert st_ == o.status_val == 1 and it_ == int(o.info.iter)
    assert (nref, nfq) == (o.counter("n_refactor"), o.counter("n_factor_Q"))
    assert rel(x, o.x) <= 1e-9 and rel(y, o.y) <= 1e-9
    assert res[0][0] == 1 and res[0][1] == it_ and rel(x, res[0][2]) <= 1e-9 and rel(y, res[0][3]) <= 1e-9   # the dense factor, same engine


def test_sparse_factor_batch_of_different_patterns(ctx):
    """members of different sizes and patterns share the batch's strides (nnz(L) of the largest)"""
    probs = [sparse_qp(60, "banded", seed=1), sparse_qp(48, "blocks", seed=2), sparse_qp(40, "arrow", seed=3), sparse_qp(33, "banded", seed=4, band=5)]
    ctx.set_option("sparse_factor", 1)
    try:
        for ordering in (0, 1):
            ctx.set_option("sparse_ordering", ordering)
            bt = QpalmBatch(ctx, probs, ctx.default_settings(**ST))
            bt.solve()
            x, y = bt.solution()
            for k, p in enumerate(probs):
                o = oracle_sparse(p, bt.sparse_perm(k)[0], **ST)
                assert int(bt.info(k).status_val) == o.status_val == 1 and int(bt.info(k).iter) == int(o.info.iter), (ordering, k)
                assert int(bt.stats(k).n_refactor) == o.counter("n_refactor") and int(bt.stats(k).n_rank1) == o.counter("n_rank1"), (ordering, k)
                assert rel(x[k][:p.n], o.x) <= 1e-9 and rel(y[k][:p.m], o.y) <= 1e-9, (ordering, k)
            bt.close()
    finally:
        ctx.set_option("sparse_factor", -1)
        ctx.set_option("sparse_ordering", -1)


def test_nested_dissection_with_a_long_row_and_hubs(ctx):
    """the ordering's corner cases: a row of A with more entries than the ordering graph spells out as a clique (> 64: it enters as a chain of
    its columns; the symbolic analysis that follows still sees the clique), and a band plus vertices of very high degree (set aside,
    ordered last) -- a valid permutation either way, and the solve equal to the oracle's under it"""
    rng = np.random.default_rng(5)
    p = sparse_qp(150, "banded", seed=8)
    A = sp.csc_matrix((p.Ax, p.Ai, p.Ap), shape=(p.m, p.n)).tolil()
    cols = rng.choice(p.n, size=70, replace=False)
    A[3, cols] = 0.1 * rng.standard_normal(70)
    A = sp.csc_matrix(A); A.sort_indices()
    p1 = type(p)(p.n, p.m, p.Qp, p.Qi, p.Qx, A.indptr.astype(np.int64), A.indices.astype(np.int64), A.data.astype(np.float64), p.q, p.bmin, p.bmax)
    p2 = sparse_qp(120, "arrow", seed=9)
    ctx.set_option("sparse_factor", 1)
    ctx.set_option("sparse_ordering", 1)
    try:
        for q in (p1, p2):
            bt = QpalmBatch(ctx, [q], ctx.default_settings(**ST))
            perm, levels = bt.sparse_perm(0)
            assert np.array_equal(np.sort(perm), np.arange(q.n))
            if q is p1 and ctx.kind == "emu":     # (the 70-clique's chain of levels takes the emulator a minute and a half: ordering + analysis here, the solve on the hardware)
                bt.close()
                continue
            bt.solve()
            x, y = bt.solution()
            o = oracle_sparse(q, perm, **ST)
            assert int(bt.info(0).status_val) == o.status_val == 1 and int(bt.info(0).iter) == int(o.info.iter)
            assert rel(x[0], o.x) <= 1e-9 and rel(y[0], o.y) <= 1e-9
            bt.close()
        assert perm[-1] == p2.n - 1          # the arrow's shaft (the dense last row / column of Q) goes last
    finally:
        ctx.set_option("sparse_factor", -1)
        ctx.set_option("sparse_ordering", -1)


@pytest.mark.parametrize("kind,n", [("banded", 3000), ("arrow", 2000), ("blocks", 1600)])
def test_nested_dissection_makes_chains_into_trees(ctx, kind, n):
    """the symbolic side alone (no solve): a band's elimination tree under the natural ordering is a chain of n columns; dissected, its
    height is a few dozen columns at a modest price in fill; a block-diagonal pattern is left as it is.  The automatic choice
    (sparse_ordering = -1) takes the dissection for the chain and the natural ordering for the forest."""
    p = sparse_qp(n, kind, seed=2)
    out = {}
    ctx.set_option("sparse_factor", 1)
    try:
        for ordering in (0, 1, -1):
            ctx.set_option("sparse_ordering", ordering)
            bt = QpalmBatch(ctx, [p], ctx.default_settings(**ST))
            perm, levels = bt.sparse_perm(0)
            assert np.array_equal(np.sort(perm), np.arange(p.n))
            out[ordering] = (levels, bt.sparse_info(0)[0], perm)
            bt.close()
    finally:
        ctx.set_option("sparse_factor", -1)
        ctx.set_option("sparse_ordering", -1)
    print(kind, n, {k: v[:2] for k, v in out.items()})
    if kind == "blocks":
        assert out[1][0] == out[0][0] == 8 and out[1][1] == out[0][1]
        assert np.array_equal(out[-1][2], np.arange(p.n))
    else:
        assert out[0][0] == p.n                                  # one column per level
        assert out[1][0] <= 120 and out[1][1] <= 3 * out[0][1] + p.n
        assert out[-1][0] == out[1][0] and np.array_equal(out[-1][2], out[1][2])


def test_sparse_factor_refuses_what_it_does_not_cover(ctx):
    p = sparse_qp(40, "banded", seed=1)
    ctx.set_option("sparse_factor", 1)
    try:
        bt = QpalmBatch(ctx, [p], ctx.default_settings(**ST))
        with pytest.raises(Exception):
            bt.factor(0)                      # reading the factor back as a dense panel works on the dense panel only
        bt.close()
        # KKT mode keeps the dense factor even when the sparse one is asked for (dual termination no longer does: round 6, next test)
        for kw in (dict(factorization_method=0),):
            bt = QpalmBatch(ctx, [p], ctx.default_settings(**dict(ST, **kw)))
            with pytest.raises(Exception):
                bt.sparse_info(0)
            bt.solve()
            assert int(bt.info(0).status_val) == 1
            bt.close()
    finally:
        ctx.set_option("sparse_factor", -1)


@pytest.mark.parametrize("kind,n,ordering", [("blocks", 96, 0), ("banded", 90, 1), ("random", 50, 0)])
def test_sparse_factor_with_dual_termination_and_sigma_path_updates(ctx, kind, n, ordering):
    """Round 6, two things the sparse factor refused or did differently from the reference through round 5:
     * dual termination (qpalm.c:459-468, 545-583): the second resident factor LD_Q = L D L' of Q alone, here on the main factor's pattern in a
       second value array; the dual objective at every outer iteration against the oracle's, and the DUAL_TERMINATED exit at the same iteration;
     * ldlupdate_sigma_changed (solver_interface.c:443-503) as rank-1 path updates where the tree is bushy enough for them to pay, instead of
       a refactorisation at every change of sigma: refactorisation and rank-1 counts equal the sparse-mode oracle's, which applies the same rule."""
    p = random_qp(n, 2 * n, seed=7, density_A=0.06, density_M=0.04) if kind == "random" else sparse_qp(n, kind, seed=4)
    ctx.set_option("sparse_factor", 1)
    ctx.set_option("sparse_ordering", ordering)
    try:
        st = dict(ST, enable_dual_termination=1, dual_objective_limit=1e20)
        bt = QpalmBatch(ctx, [p], ctx.default_settings(**st))
        assert bt.sparse_info(0)[0] > 0                       # the batch really keeps the sparse factor
        bt.solve()
        o = oracle_sparse(p, bt.sparse_perm(0)[0], **st)
        info, s = bt.info(0), bt.stats(0)
        assert int(info.status_val) == o.status_val == 1 and int(info.iter) == int(o.info.iter)
        assert abs(info.dual_objective - o.info.dual_objective) <= 1e-8 * max(1.0, abs(o.info.dual_objective)), (info.dual_objective, o.info.dual_objective)
        assert abs(info.dual_objective - info.objective) <= 1e-4 * max(1.0, abs(info.objective))      # (and it IS the dual objective: no gap at the solution)
        assert rel(bt.solution()[0][0], o.x) <= 1e-9 and rel(bt.solution()[1][0], o.y) <= 1e-9
        assert int(s.n_refactor) == o.counter("n_refactor") and int(s.n_rank1) == o.counter("n_rank1")
        if kind == "blocks":   # bushy tree: the penalties that change are path updates, so fewer factorisations than "every change refactorises" (mode 2)
            o2 = ob.OracleQP(*p.args(), settings=ob.default_settings(**st)); o2.set_scalar("sparse_mode", 2); o2.solve()
            assert int(s.n_sigma_updates) > 0 and int(s.n_refactor) < o2.counter("n_refactor")
        bt.close()
        # the early exit: a limit below the optimal value is crossed by the (increasing) dual objective on the way
        d0 = float(o.info.dual_objective)
        for lim in (d0 - 0.5 * abs(d0) - 1.0, d0 - 0.01 * abs(d0) - 1e-3):       # crossed by the starting point / late in the solve
            st2 = dict(st, dual_objective_limit=lim)
            bt = QpalmBatch(ctx, [p], ctx.default_settings(**st2))
            bt.solve()
            o2 = oracle_sparse(p, bt.sparse_perm(0)[0], **st2)
            assert int(bt.info(0).status_val) == o2.status_val == 2 and int(bt.info(0).iter) == int(o2.info.iter), (lim, bt.info(0).status_val, o2.status_val)   # DUAL_TERMINATED
            assert abs(bt.info(0).dual_objective - o2.info.dual_objective) <= 1e-8 * max(1.0, abs(o2.info.dual_objective))
            bt.close()
    finally:
        ctx.set_option("sparse_factor", -1)
        ctx.set_option("sparse_ordering", -1)


def test_sparse_factor_single_operations_of_the_boundary(ctx):
    """Round 6: ldlcholQAtsigmaA, ldlupdate_entering / ldldowndate_leaving, ldlupdate_sigma_changed and ldlsolveLD_neg_dphi of solver_interface.h
    on a batch that keeps the SPARSE factor (refused through round 5), each against the same operation of the oracle in its sparse-storage mode on
    the same state.  The sparse factor cannot be read back entry by entry, so every step is checked through a solve with a fixed right-hand side --
    and the updated factor against a fresh factorisation of the new active set (an oracle-free property)."""
    import ctypes as C
    n = 96
    p = sparse_qp(n, "blocks", seed=6)
    m = p.m
    ctx.set_option("sparse_factor", 1)
    ctx.set_option("sparse_ordering", 0)
    try:
        bt = QpalmBatch(ctx, [p], ctx.default_settings(**ST))
        assert bt.sparse_info(0)[0] > 0
        o = ob.OracleQP(*p.args(), settings=ob.default_settings(**dict(ST, max_iter=5)))
        o.set_scalar("sparse_mode", 1)
        o.solve()
        bt.iterate(5)
        assert rel(bt.vec("x"), o.vec("x")) <= 1e-9
        L = ob.lib()
        ln = ob.c_int(0)
        rhs = np.random.default_rng(3).standard_normal(n)

        def solve_both():
            o.vec("dphi", copy=False)[:] = rhs
            bt.set_vec("dphi", rhs)
            L.oq_ldlsolveLD_neg_dphi(o.w)
            bt.op("ldlsolveLD_neg_dphi")
            return bt.vec("d").copy(), o.vec("d").copy()
        act = (np.arange(m) % 3 == 0).astype(np.int64)
        pa = L.oq_get_ivec(o.w, b"active", C.byref(ln))
        np.ctypeslib.as_array(pa, shape=(m,))[:] = act
        bt.set_ivec("active", act)
        L.oq_ldlcholQAtsigmaA(o.w)
        bt.op("ldlcholQAtsigmaA")
        dg, do = solve_both()
        assert rel(dg, do) <= 1e-10
        enter, leave = np.where(act == 0)[0][:9], np.where(act == 1)[0][3:10]
        act_new = act.copy(); act_new[enter] = 1; act_new[leave] = 0
        for name, lst in (("enter", enter), ("leave", leave)):
            bt.set_ivec(name, lst)
        np.ctypeslib.as_array(L.oq_get_ivec(o.w, b"active_old", C.byref(ln)), shape=(m,))[:] = act
        np.ctypeslib.as_array(pa, shape=(m,))[:] = act_new
        L.oq_set_entering_leaving_constraints(o.w)
        L.oq_ldlupdate_entering_constraints(o.w)
        L.oq_ldldowndate_leaving_constraints(o.w)
        bt.set_scalar("nb_enter", len(enter)); bt.set_scalar("nb_leave", len(leave))
        bt.op("ldlupdate_entering_constraints")
        bt.op("ldldowndate_leaving_constraints")
        dg, do = solve_both()
        assert rel(dg, do) <= 1e-9
        bt.set_ivec("active", act_new)                       # property: the updated factor = a fresh one of the new active set
        bt.op("ldlcholQAtsigmaA")
        dfresh, _ = solve_both()
        assert rel(dg, dfresh) <= 1e-9
        # ldlupdate_sigma_changed: five rows of the active set with grown penalties (At_scale = sqrt(mult_factor), the list in enter[])
        L.oq_ldlcholQAtsigmaA(o.w)
        changed = np.where(act_new == 1)[0][2:7]
        scale = np.ones(m); scale[changed] = np.sqrt(1.0 + 50.0 * np.random.default_rng(4).random(5))
        o.vec("At_scale", copy=False)[:] = scale
        bt.set_vec("At_scale", scale)
        np.ctypeslib.as_array(L.oq_get_ivec(o.w, b"enter", C.byref(ln)), shape=(m,))[:5] = changed
        bt.set_ivec("enter", changed)
        L.oq_set_scalar(o.w, b"nb_sigma_changed", 5.0)
        bt.set_scalar("nb_sigma_changed", 5)
        L.oq_ldlupdate_sigma_changed(o.w)
        bt.op("ldlupdate_sigma_changed")
        dg, do = solve_both()
        assert rel(dg, do) <= 1e-9 and np.array_equal(bt.vec("At_scale"), o.vec("At_scale"))
        with pytest.raises(Exception):
            bt.factor(0)                                     # (the dense read-back of the factor stays refused)
        bt.close()
    finally:
        ctx.set_option("sparse_factor", -1)
        ctx.set_option("sparse_ordering", -1)


def _kkt_check(p, x, y, tol=1e-5):
    A = sp.csc_matrix((p.Ax, p.Ai, p.Ap), shape=(p.m, p.n))
    Ql = sp.csc_matrix((p.Qx, p.Qi, p.Qp), shape=(p.n, p.n))
    Q = Ql + sp.tril(Ql, -1).T
    ax = A @ x
    prim = np.max(np.maximum(p.bmin - ax, 0) + np.maximum(ax - p.bmax, 0))
    grad = Q @ x + p.q + A.T @ y
    assert prim <= tol * max(1.0, np.max(np.abs(ax))), prim
    assert np.max(np.abs(grad)) <= tol * max(1.0, np.max(np.abs(Q @ x)), np.max(np.abs(p.q))), np.max(np.abs(grad))
    # complementarity: a multiplier pushes only against the bound its row sits on
    assert np.all((y <= 1e-6) | (np.abs(ax - p.bmax) <= 1e-4 * max(1.0, np.max(np.abs(ax)))))
    assert np.all((y >= -1e-6) | (np.abs(ax - p.bmin) <= 1e-4 * max(1.0, np.max(np.abs(ax)))))


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n,ordering", [("blocks", 100000, -1), ("banded", 20000, -1), ("arrow", 20000, -1), ("banded", 20000, 0), ("banded", 100000, -1)])
def test_sparse_factor_at_size(kind, n, ordering):
    """beyond any dense panel (qpg_batch_create refused more than 8192 rows through round 4): selected automatically.  ordering -1: the
    library's choice (nested dissection for the two chains, natural for the forest); 0: the reference's natural ordering on the chain"""
    import time
    from qpalm_amd.solver import Context
    ctx = Context(0)
    ctx.set_option("sparse_ordering", ordering)
    p = sparse_qp(n, kind, seed=21)
    bt = QpalmBatch(ctx, [p], ctx.default_settings(**ST))
    nnzL, nbytes = bt.sparse_info(0)
    perm, levels = bt.sparse_perm(0)
    assert (levels <= 100) if (ordering != 0 or kind == "blocks") else (levels == p.n)
    t0 = time.perf_counter()
    bt.solve()
    dt = time.perf_counter() - t0
    x, y = bt.solution()
    info, s = bt.info(0), bt.stats(0)
    print("sparse factor, %s n = %d m = %d, ordering %d: %d levels, nnz(L) = %d (dense triangle %.3g), device block %.1f MB, %d iterations, %d + %d factorisations, %d path updates, %.2f s" % (
        kind, p.n, p.m, ordering, levels, nnzL, 0.5 * p.n * p.n, nbytes / 2 ** 20, int(info.iter), int(s.n_refactor), int(s.n_factor_Q), int(s.n_rank1), dt))
    assert int(info.status_val) == 1
    assert nnzL <= 40 * p.n and nbytes <= 200 * 8 * (nnzL + 40 * p.n)      # memory proportional to nnz(L) (+ O(n) work vectors), nowhere near n^2
    _kkt_check(p, x[0], y[0])
    t0 = time.perf_counter()
    o = oracle_sparse(p, perm, **ST)
    print("  the oracle (one CPU core, sparse storage, same ordering): %.2f s" % (time.perf_counter() - t0))
    assert o.status_val == 1 and int(info.iter) == int(o.info.iter) and int(s.n_refactor) == o.counter("n_refactor") and int(s.n_rank1) == o.counter("n_rank1")
    assert rel(x[0], o.x) <= 1e-8 and rel(y[0], o.y) <= 1e-8
    bt.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n", [("banded", 2000), ("blocks", 2000)])
def test_sparse_batch_throughput(kind, n):
    """what the sparse mode is for on this hardware: MANY sparse QPs at once, one workgroup each (a single one runs at the latency of its
    tree: test_sparse_factor_at_size).  2048 QPs with n = 2000 (64 distinct ones, repeated); a sample against the oracle under the same
    ordering; the oracle's time on one core printed beside it (profiles/r05)."""
    import time
    from qpalm_amd.solver import Context
    ctx = Context(0)
    ctx.set_option("sparse_factor", 1)
    distinct = [sparse_qp(n, kind, seed=700 + k) for k in range(64)]
    probs = [distinct[k % 64] for k in range(2048)]
    bt = QpalmBatch(ctx, probs, ctx.default_settings(**ST))
    bt.solve()                                      # (warm-up: first launch of the kernel)
    bt.warm_start(None, None)
    t0 = time.perf_counter()
    bt.solve()
    dt = time.perf_counter() - t0
    x, y = bt.solution()
    assert all(int(bt.info(k).status_val) == 1 for k in range(len(probs)))
    t_or = 0.0
    for k in range(0, 64, 8):
        t1 = time.perf_counter()
        o = oracle_sparse(distinct[k], bt.sparse_perm(k)[0], **ST)
        t_or += time.perf_counter() - t1
        for kk in (k, k + 64 * 31):
            assert int(bt.info(kk).iter) == int(o.info.iter) and rel(x[kk], o.x) <= 1e-8 and rel(y[kk], o.y) <= 1e-8, kk
    nnzL, nbytes = bt.sparse_info(0)
    print("sparse batch, %s n = %d: %d QPs in %.3f s = %.0f QP/s (%d levels, nnz(L) = %d, device block %.0f MB); the oracle on one core, same ordering: %.1f QP/s" % (
        kind, n, len(probs), dt, len(probs) / dt, bt.sparse_perm(0)[1], nnzL, nbytes / 2 ** 20, 8 / t_or))
    bt.close()


def write_free_qps(path, p, name="SPARSEQP"):
    """a QP of qpalm_amd.problems as a free-format QPS file: every row an L row with RHS = bmax and RANGES = bmax - bmin, every variable FR
    (the reader then appends no bound rows), QUADOBJ = the lower triangle"""
    A = sp.csc_matrix((p.Ax, p.Ai, p.Ap), shape=(p.m, p.n))
    Q = sp.csc_matrix((p.Qx, p.Qi, p.Qp), shape=(p.n, p.n))
    with open(path, "w") as f:
        f.write("NAME %s\nROWS\n N obj\n" % name)
        for i in range(p.m):
            f.write(" L r%d\n" % i)
        f.write("COLUMNS\n")
        for j in range(p.n):
            if p.q[j] != 0.0:
                f.write(" x%d obj %r\n" % (j, float(p.q[j])))
            for k in range(A.indptr[j], A.indptr[j + 1]):
                f.write(" x%d r%d %r\n" % (j, A.indices[k], float(A.data[k])))
        f.write("RHS\n")
        for i in range(p.m):
            f.write(" rhs r%d %r\n" % (i, float(p.bmax[i])))
        f.write("RANGES\n")
        for i in range(p.m):
            f.write(" rng r%d %r\n" % (i, float(p.bmax[i] - p.bmin[i])))
        f.write("BOUNDS\n")
        for j in range(p.n):
            f.write(" FR bnd x%d\n" % j)
        f.write("QUADOBJ\n")
        for j in range(p.n):
            for k in range(Q.indptr[j], Q.indptr[j + 1]):
                if Q.indices[k] >= j:
                    f.write(" x%d x%d %r\n" % (j, Q.indices[k], float(Q.data[k])))
        f.write("ENDATA\n")


@pytest.mark.gpu
def test_large_sparse_qps_files_through_the_reader(tmp_path):
    """BASELINE.json config 4 on members the dense panel could not take: two generated QPS files with n = 12 000 and 30 000 (the large
    Maros-Meszaros members -- LISWET, CONT-xxx, AUG3DCQP, BOYD -- have n = 10^4 .. 10^5 and > 99 % sparse Schur complements; the set itself
    is not in the image) and a small dense-friendly one, streamed through interfaces/qps' reader (the shipped libqpalm.so) as size-bucketed
    batches; each against the oracle (sparse storage for the large ones) on the data the reader returned."""
    from qpalm_amd.qps import read_qps, solve_qps_files
    from qpalm_amd.solver import Context
    ctx = Context(0)
    ctx.set_option("sparse_ordering", 0)     # the reference's natural ordering (the batches are made inside solve_qps_files: the oracle below is not told a permutation)
    gens = [("banded12k", sparse_qp(12000, "banded", seed=31)), ("blocks30k", sparse_qp(30000, "blocks", seed=32)), ("small", sparse_qp(40, "banded", seed=33))]
    paths = []
    for name, p in gens:
        path = str(tmp_path / (name + ".qps"))
        write_free_qps(path, p, name.upper())
        paths.append(path)
    res = solve_qps_files(ctx, paths, ctx.default_settings(**ST))
    for path, (name, p0) in zip(paths, gens):
        p = read_qps(path)
        assert (p.n, p.m) == (p0.n, p0.m) and np.array_equal(p.Ax, p0.Ax) and np.array_equal(p.Qx, p0.Qx) and np.array_equal(p.bmax, p0.bmax)
        assert np.allclose(p.bmin, p0.bmin, rtol=0, atol=1e-15)      # (bmin comes back as bmax - (bmax - bmin): RANGES on L rows)
        x, y, info = res[path]
        o = ob.OracleQP(*p.args(), c=p.c, settings=ob.default_settings(**ST))
        if p.n > 8192:
            o.set_scalar("sparse_mode", 1)
        o.solve()
        assert int(info.status_val) == o.status_val == 1 and int(info.iter) == int(o.info.iter), (name, int(info.iter), int(o.info.iter))
        assert rel(x, o.x) <= 1e-8 and rel(y, o.y) <= 1e-8, name
        _kkt_check(p, x, y)


def test_sparse_factor_nonconvex_warm_start_updates_and_queue(ctx):
    """the rest of the loop on a sparse batch: an indefinite Q (LOBPCG front-end, proximal penalty 1 / |lambda|, L D L' with negative pivots),
    a warm-started re-solve after update_bounds / update_q, and more QPs than resident factor slots (the values of a slot are reused by
    the next QP of the work queue, the symbolic arrays stay per QP)"""
    ctx.set_option("sparse_factor", 1)
    try:
        # (a) nonconvex
        p = sparse_qp(72, "blocks", seed=41)
        Q = sp.csc_matrix((p.Qx, p.Qi, p.Qp), shape=(p.n, p.n)).tolil()
        for j in range(0, p.n, 4):
            Q[j, j] = Q[j, j] - 3.0 * abs(Q[j, j])
        Q = sp.csc_matrix(Q); Q.sort_indices()
        p2 = type(p)(p.n, p.m, Q.indptr.astype(np.int64), Q.indices.astype(np.int64), Q.data.copy(), p.Ap, p.Ai, p.Ax, p.q, p.bmin, p.bmax)
        st = dict(ST, nonconvex=1)
        bt = QpalmBatch(ctx, [p2], ctx.default_settings(**st))
        bt.solve()
        o = oracle_sparse(p2, **st)
        assert int(bt.stats(0).nonconvex) == o.counter("nonconvex") == 1
        assert int(bt.info(0).status_val) == o.status_val and int(bt.info(0).iter) == int(o.info.iter)
        x, y = bt.solution()
        assert rel(x[0], o.x) <= 1e-8 and rel(y[0], o.y) <= 1e-8
        bt.close()
        # (b) update_bounds / update_q + warm start
        p = sparse_qp(80, "banded", seed=42)
        bt = QpalmBatch(ctx, [p], ctx.default_settings(**ST))
        o = ob.OracleQP(*p.args(), settings=ob.default_settings(**ST)); o.set_scalar("sparse_mode", 1)
        bt.solve(); o.solve()
        x, y = bt.solution()
        bmin2, bmax2, q2 = p.bmin - 0.05, p.bmax + 0.1, p.q * 1.1
        bt.update_bounds(bmin2[None, :], bmax2[None, :]); bt.update_q(q2[None, :]); bt.warm_start(x, y)
        o.update_bounds(bmin2, bmax2); o.update_q(q2); o.warm_start(o.x, o.y)
        bt.solve(); o.solve()
        x, y = bt.solution()
        assert int(bt.info(0).status_val) == o.status_val == 1 and int(bt.info(0).iter) == int(o.info.iter)
        assert rel(x[0], o.x) <= 1e-8 and rel(y[0], o.y) <= 1e-8
        bt.close()
        # (c) more QPs than slots
        probs = [sparse_qp(40 + 4 * (k % 3), ("banded", "blocks", "arrow")[k % 3], seed=50 + k) for k in range(7)]
        ctx.set_option("max_slots", 2)
        bt = QpalmBatch(ctx, probs, ctx.default_settings(**ST))
        bt.solve()
        x, y = bt.solution()
        for k, p in enumerate(probs):
            o = oracle_sparse(p, **ST)
            assert int(bt.info(k).status_val) == o.status_val == 1 and int(bt.info(k).iter) == int(o.info.iter), k
            assert rel(x[k][:p.n], o.x) <= 1e-9 and rel(y[k][:p.m], o.y) <= 1e-9, k
        bt.close()
    finally:
        ctx.set_option("max_slots", 512)
        ctx.set_option("sparse_factor", -1)


def test_sparse_lds_forms_are_bit_identical_to_the_hbm_forms(ctx):
    """Round 6: the factorisation accumulates a column in LDS (positions found by binary search in the column's pattern, the entries of four
    contributing rows / columns requested at once) and the solves keep the right-hand side in LDS: per entry the same operations in the same
    order as the forms that work on vectors in HBM (context option "sparse_lds" = 0), so the iterates are equal BIT FOR BIT -- with rows of A
    and columns of L longer than a group of lanes (a 40-column row of A: a 40-clique), under both orderings, with the LDS form limited to
    columns of at most six entries ("sparse_lds" = 6: the longer columns of the same factorisation take the HBM form; the default limit is two
    passes of a group, 32 entries) and with columns of up to 200 entries in LDS ("sparse_lds" = 200: the 40-clique's columns)."""
    rng = np.random.default_rng(5)
    p = sparse_qp(60, "banded", seed=8)
    A = sp.csc_matrix((p.Ax, p.Ai, p.Ap), shape=(p.m, p.n)).tolil()
    A[3, rng.choice(p.n, size=40, replace=False)] = 0.1 * rng.standard_normal(40)
    A = sp.csc_matrix(A); A.sort_indices()
    p1 = type(p)(p.n, p.m, p.Qp, p.Qi, p.Qx, A.indptr.astype(np.int64), A.indices.astype(np.int64), A.data.astype(np.float64), p.q, p.bmin, p.bmax)
    cases = [(p1, 0, 6), (p1, 1, 6), (sparse_qp(48, "blocks", seed=2), 0, 0), (sparse_qp(40, "arrow", seed=3), 1, 0), (random_qp(40, 80, seed=7, density_A=0.05, density_M=0.04), 0, 0)]
    ctx.set_option("sparse_factor", 1)
    try:
        for q, ordering, iters in cases:
            ctx.set_option("sparse_ordering", ordering)
            res = {}
            for mode in (0, 1, 6, 200):
                ctx.set_option("sparse_lds", mode)
                bt = QpalmBatch(ctx, [q], ctx.default_settings(**dict(ST, enable_dual_termination=1)))
                if iters:
                    bt.iterate(iters)
                    res[mode] = (bt.vec("x").copy(), bt.vec("y").copy(), 0)
                else:
                    bt.solve()
                    x, y = bt.solution()
                    assert int(bt.info(0).status_val) == 1
                    res[mode] = (x[0].copy(), y[0].copy(), int(bt.info(0).iter))
                bt.close()
            for mode in (1, 6, 200):
                assert res[mode][2] == res[0][2] and np.array_equal(res[mode][0], res[0][0]) and np.array_equal(res[mode][1], res[0][1]), (ordering, mode)
    finally:
        ctx.set_option("sparse_factor", -1)
        ctx.set_option("sparse_ordering", -1)
        ctx.set_option("sparse_lds", 1)


def test_sparse_columns_per_wavefront_do_not_change_the_iterates(ctx):
    """Round 6: a wavefront factorises 1, 2, 4 or 8 columns of a level at a time (context option "sparse_gpw"; the default went from four groups of
    16 lanes to eight groups of eight).  Which lanes own a column changes neither the operations of an entry nor their order: same iterates bit for
    bit, same iteration count -- for a band under nested dissection, a forest of blocks and an arrow (columns longer than a group has lanes)."""
    cases = [(sparse_qp(90, "banded", seed=11), 1), (sparse_qp(96, "blocks", seed=11), 0), (sparse_qp(60, "arrow", seed=11), 1)]
    ctx.set_option("sparse_factor", 1)
    try:
        for q, ordering in cases:
            ctx.set_option("sparse_ordering", ordering)
            res = {}
            for gpw in (8, 4, 2, 1):
                ctx.set_option("sparse_gpw", gpw)
                bt = QpalmBatch(ctx, [q], ctx.default_settings(**ST))
                bt.solve()
                x, y = bt.solution()
                assert int(bt.info(0).status_val) == 1
                res[gpw] = (x[0].copy(), y[0].copy(), int(bt.info(0).iter))
                bt.close()
            for gpw in (4, 2, 1):
                assert res[gpw][2] == res[8][2] and np.array_equal(res[gpw][0], res[8][0]) and np.array_equal(res[gpw][1], res[8][1]), (ordering, gpw)
    finally:
        ctx.set_option("sparse_factor", -1)
        ctx.set_option("sparse_ordering", -1)
        ctx.set_option("sparse_gpw", 0)
