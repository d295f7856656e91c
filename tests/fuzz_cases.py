"""Randomised parity cases (TEST INFRASTRUCTURE, used by tools/evidence/fuzz_parity.py and tests/test_fuzz_seeds.py): small random QPs with
random shapes, bound patterns, settings (scaling, proximal, sigma, gamma, dual termination, KKT / Schur, inner_max_iter,
max_iter) and warm starts.  `cases(seed, count, n_lo, n_hi)` yields (index, problem, settings, warm_start) in a fixed order:
case k of a seed is always the same problem (the draws of a case do not depend on any solve)."""
import numpy as np

from qpalm_amd.problems import random_qp


def cases(seed, count, n_lo=2, n_hi=70, force=None):
    """force: dict of settings that overrides the drawn ones (the draws are made all the same, so the stream stays aligned);
    the pseudo-setting q_shift makes the Hessian indefinite (use with nonconvex=1), lp = 1 zeroes Q, q_scale multiplies it"""
    rng = np.random.default_rng(int(seed))
    for it in range(count):
        n = int(rng.integers(n_lo, n_hi))
        m = int(rng.integers(1, max(2, int(1.7 * n_hi))))
        dA = float(rng.choice([0.05, 0.15, 0.4, 1.0])) * min(1.0, 70.0 / n)
        dM = float(rng.choice([0.02, 0.1, 0.5])) * min(1.0, 70.0 / n)
        p = random_qp(n, m, seed=int(rng.integers(1 << 30)), density_A=dA, density_M=dM)
        mode = int(rng.integers(0, 4))          # widen / tighten / equality / infinite bounds
        if mode == 1:
            p.bmax[:] = p.bmin + 0.0
        if mode == 2:
            p.bmin[rng.random(m) < 0.5] = -1e20
            p.bmax[rng.random(m) < 0.5] = 1e20
        if mode == 3:
            p.bmin *= 10
            p.bmax *= 10
        st = dict(eps_abs=float(rng.choice([1e-4, 1e-6, 1e-8])), eps_rel=float(rng.choice([1e-4, 1e-6, 1e-8])), verbose=0,
                  scaling=int(rng.choice([0, 1, 2, 10])), proximal=int(rng.integers(0, 2)), max_iter=int(rng.choice([50, 1000, 10000])),
                  sigma_init=float(rng.choice([2e1, 1.0, 1e3])), theta=float(rng.choice([0.25, 0.5])), delta=float(rng.choice([10, 100])),
                  gamma_init=float(rng.choice([1e1, 1e4, 1e7])), gamma_max=1e7, enable_dual_termination=int(rng.random() < 0.2),
                  factorization_method=int(rng.choice([0, 1, 1, 2])), inner_max_iter=int(rng.choice([5, 100])))
        warm = None
        if rng.random() < 0.3:
            warm = (rng.standard_normal(n), rng.standard_normal(m))
        if force:
            st.update({k: v for k, v in force.items() if k not in ("q_shift", "lp", "q_scale")})
            if force.get("q_scale"):
                p.Qx[:] = p.Qx * float(force["q_scale"])   # pseudo-setting q_scale: a positive definite but tiny Hessian (positive diagonal: not flagged at setup)
            if force.get("lp"):
                p.Qx[:] = 0.0   # pseudo-setting lp = 1: a linear programme -- H = A' Sigma A + I / gamma, pivots down to 1 / gamma_max
            if force.get("q_shift"):
                # indefinite Hessian for the nonconvex front-end: the diagonal of Q lowered by q_shift x its mean (every column of a
                # random_qp Hessian stores its diagonal entry first)
                Qp, Qi = np.asarray(p.Qp), np.asarray(p.Qi)
                dpos = np.array([k for j in range(n) for k in range(int(Qp[j]), int(Qp[j + 1])) if int(Qi[k]) == j], dtype=np.int64)
                if dpos.size:
                    p.Qx[dpos] -= float(force["q_shift"]) * float(np.mean(np.abs(p.Qx[dpos])))
        yield it, p, st, warm, dict(n=n, m=m, dA=dA, dM=dM, mode=mode)


def rel(a, b):
    return (np.max(np.abs(a - b)) / max(1.0, np.max(np.abs(b)))) if a.size else 0.0


def y_bound_from_dx(p, o, dx):
    """What a difference dx of the primal solutions explains in the multipliers, per outer update: y <- y + sigma o (A x - z) in the scaled
    variables (iteration.c:26-35), i.e. for a row that stays active  dy_i = c^-1 E_i^2 sigma_i (A dx)_i  in the unscaled ones (x = D xbar,
    y = c^-1 E ybar, Abar = E A D: scaling.c:34-113) -- with the ORACLE's own final penalties and scaling, relative to max(1, |y|_inf)
    like rel().  Every outer update adds one such term (the caller multiplies by the number of outer iterations)."""
    import scipy.sparse as sp
    if p.m == 0:
        return 0.0
    A = sp.csc_matrix((np.asarray(p.Ax, float), np.asarray(p.Ai), np.asarray(p.Ap)), shape=(p.m, p.n))
    sig = o.vec("sigma")
    E, cinv = np.ones(p.m), 1.0
    if int(o.settings.scaling) > 0:
        E, cinv = o.vec("E"), o.scalar("cinv")
        if not np.all(np.isfinite(E)) or not np.isfinite(cinv) or cinv == 0.0:
            E, cinv = np.ones(p.m), 1.0
    t = np.abs(cinv * E * E * sig * (A @ dx))
    return float(np.max(t) / max(1.0, float(np.max(np.abs(o.y))))) if t.size else 0.0


def run_case(ctx, p, st, warm, oracle_sparse_mode=0):
    """the engine and the oracle on one case -> dict(status, iter (engine, oracle), x / y relative differences)
    oracle_sparse_mode: the oracle's storage of the Schur factor (campaign S: the engine runs with the context option sparse_factor = 1)"""
    import oracle.binding as ob
    from qpalm_amd.solver import QpalmBatch
    bt = QpalmBatch(ctx, [p], ctx.default_settings(**st))
    o = ob.OracleQP(*p.args(), settings=ob.default_settings(**st))
    if oracle_sparse_mode:
        o.set_scalar("sparse_mode", oracle_sparse_mode)
        perm = bt.sparse_perm(0)[0]     # the ordering the engine chose for this factor: the checker factorises the same P H P'
        if not np.array_equal(perm, np.arange(len(perm))):
            o.set_perm(perm)
    if warm is not None:
        bt.warm_start(warm[0][None, :], warm[1][None, :])
        o.warm_start(warm[0], warm[1])
    bt.solve()
    o.solve()
    info = bt.info(0)
    x, y = bt.solution()
    sig = bt.vec("sigma", 0)
    res = dict(status=(int(info.status_val), int(o.status_val)), iter=(int(info.iter), int(o.info.iter)),
               nonfinite=(not (np.all(np.isfinite(x[0])) and np.all(np.isfinite(y[0]))), not (np.all(np.isfinite(o.x)) and np.all(np.isfinite(o.y)))),
               guard=(int(bt.stats(0).n_guard_refactor), int(o.counter("n_guard_refactor"))),
               dx=rel(x[0], o.x), dy=rel(y[0], o.y), ymax=float(np.max(np.abs(o.y))) if o.y.size else 0.0,
               obj=(float(info.objective), float(o.info.objective)), sigma_max=float(np.max(sig)) if sig.size else 0.0,
               iter_out=int(o.info.iter_out), ybound=y_bound_from_dx(p, o, x[0] - o.x))
    bt.close()
    o.cleanup()
    return res


# ---- when the engine and the oracle disagree on (status, iterations): is the count a property of the algorithm on this case? -------------
_VARIANTS = {}


def oracle_variants():
    """the ORACLE's own source compiled three ways (TEST INFRASTRUCTURE): as shipped (no contraction), with fused multiply-adds
    (-O3 -ffp-contract=fast -mfma: what the device's explicit fma() in SpMV dots and the LDL' kernels corresponds to) and -Ofast -mfma
    (re-associated sums as well: the device's reductions are trees).  Built once per process."""
    import os
    import subprocess
    if not _VARIANTS:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        # "pivot" (round 5): the rank-update recurrence in the device code's algebraically equal form (1 / alpha carried, d_new = d + s w^2 / alpha:
        # oracle/qpalm_oracle.c, OQ_PIVOT_ENGINE), with fused multiply-adds -- the one place where the engine's formula is not the oracle's
        for name, flags in (("fma", ["-O3", "-ffp-contract=fast", "-mfma"]), ("ofast", ["-Ofast", "-mfma"]),
                            ("pivot", ["-O3", "-ffp-contract=fast", "-mfma", "-DOQ_PIVOT_ENGINE"]), ("pivot_plain", ["-O2", "-DOQ_PIVOT_ENGINE"])):
            out = os.path.join(os.environ.get("TMPDIR", "/tmp"), "libqpalm_oracle_%s_%d.so" % (name, os.getpid()))
            subprocess.check_call(["gcc", "-std=c99", "-fPIC", "-shared", "-o", out, os.path.join(root, "oracle", "qpalm_oracle.c"), "-lm"] + flags)
            _VARIANTS[name] = out
        import atexit
        atexit.register(lambda: [os.remove(f) for f in _VARIANTS.values() if os.path.exists(f)])
    return _VARIANTS


def oracle_outcomes(p, st, warm, nonfinite=None, solutions=None):
    """{variant: (status, iter)} of the oracle variants on one case; nonfinite (a set, optional) collects the variants whose x or y is not finite,
    solutions (a dict, optional) their (x, y)"""
    import oracle.binding as ob
    out = {}
    for name, lib in oracle_variants().items():
        o = ob.OracleQP(*p.args(), settings=ob.default_settings(**st), libpath=lib)
        if warm is not None:
            o.warm_start(warm[0], warm[1])
        o.solve()
        out[name] = (int(o.status_val), int(o.info.iter))
        if nonfinite is not None and not (np.all(np.isfinite(o.x)) and np.all(np.isfinite(o.y))):
            nonfinite.add(name)
        if solutions is not None:
            solutions[name] = (o.x.copy(), o.y.copy())
        o.cleanup()
    return out


def reference_builds_spread(p, st, warm):
    """(dx, dy): how far the x and y of the reference's own source move when it is compiled with fused multiply-adds or -Ofast (NOT the variants that
    restate the engine's formulas), relative like rel(), on a case all of them solve -- the conditioning of the problem as the reference itself sees it"""
    import oracle.binding as ob
    o = ob.OracleQP(*p.args(), settings=ob.default_settings(**st))
    if warm is not None:
        o.warm_start(warm[0], warm[1])
    o.solve()
    x0, y0, s0 = o.x.copy(), o.y.copy(), int(o.status_val)
    o.cleanup()
    sol = {}
    var = oracle_outcomes(p, st, warm, None, sol)
    dx = dy = 0.0
    for name in ("fma", "ofast"):
        if var[name][0] == s0 and s0 in (1, 2):
            dx, dy = max(dx, rel(sol[name][0], x0)), max(dy, rel(sol[name][1], y0))
    return dx, dy


ENGINE_FORM_VARIANTS = ("pivot", "pivot_plain")   # oracle variants that restate the ENGINE's form of the rank-update recurrence (OQ_PIVOT_ENGINE)


def oracle_perturbed(p, st, warm, count=6):
    """{name: (status, iter)} of the plain oracle on `count` copies of the problem whose values (A, Q, q) are moved by ONE unit in the
    last place, up or down at random (fixed seeds)."""
    import copy
    import oracle.binding as ob
    out = {}
    for k in range(count):
        rng = np.random.default_rng(1000 + k)
        q = copy.deepcopy(p)
        for arr in (q.Ax, q.Qx, q.q):
            up = rng.random(arr.size) < 0.5
            arr[:] = np.where(up, np.nextafter(arr, np.inf), np.nextafter(arr, -np.inf))
        o = ob.OracleQP(*q.args(), settings=ob.default_settings(**st))
        if warm is not None:
            o.warm_start(warm[0], warm[1])
        o.solve()
        out["ulp%d" % k] = (int(o.status_val), int(o.info.iter))
        o.cleanup()
    return out


def judge_case(r, p, st, warm, ytol=1e-8, ctx=None):
    """The sharp form of "parity with the oracle" for one case.  Returns (ok, why, cls): cls is False for a case that matches the oracle outright,
    else the bucket the accepted case is counted in -- "rounding" (the reference's own source changes its outcome under the compiler's flags or a
    one-ulp perturbation of the data, or the trajectory criterion holds), "engine-form" (only the oracle variants that restate the ENGINE's form
    of the rank-update recurrence move), "singular" (non-finite iterates in the engine AND in a reference build: H singular by construction), "conditioning" (equal status and count, x
    or y beyond the tolerance but within twice what the reference's own source moves under -ffp-contract=fast / -Ofast: the solution is determined to kappa eps).
      * (status, iterations) equal to the plain oracle's: x within 1e-8, y within ytol of it (solved cases).
      * otherwise the case must be one whose count ROUNDING decides -- the oracle's own source, compiled with fused multiply-adds or
        -Ofast, does not reproduce the plain oracle's (status, iterations) either; or (noise_decided_branch) the two iteration paths
        are the same up to the first iteration where they branch differently, and that branch is "Newton step or outer step" taken
        on an inner residual that the preceding exact Newton step had already reduced to rounding noise (below 1e-7 of its value
        before the step) in BOTH implementations, the threshold lying in that noise band -- the engine's status must be one an
        oracle variant reaches, and when both solved the objectives agree to 10 x the case's tolerance (x, y need not: such cases
        include degenerate problems with several minimisers, where the path decides which one is returned)."""
    if r["status"][0] == r["status"][1] and r["iter"][0] == r["iter"][1]:
        # y <- y + sigma (Ax - z): the multipliers carry the rounding-level difference of x multiplied by the penalties the solve has
        # reached (the engine's final sigma, up to sigma_max = 1e9) -- seen on the KKT path, whose quasi-definite solves leave dx ~ 1e-10
        # where the Schur path leaves 1e-14: dy / dx = 1e5 .. 1e6 with sigma_max = 1e4 .. 1e5 (round 4's fresh-seed campaign).  So y must
        # agree to max(ytol, 100 sigma_max dx): what x explains, nothing more (capped: never looser than 1e-3).
        # Round 5: the bound is DERIVED, not fitted -- per outer update the multipliers of the active rows move by c^-1 E^2 sigma o (A dx)
        # (y_bound_from_dx: the two x's, the ORACLE's final sigma and scaling), and there are iter_out such updates; 2 x for the rows
        # whose penalty grew on the way.  (Round 4 used 100 x the engine's largest sigma x |dx|_inf, a factor fitted to one campaign.)
        ytol = min(1e-4, max(ytol, 2.0 * max(1, r.get("iter_out", 1)) * r.get("ybound", 0.0)))
        r["ytol_used"] = ytol
        if r.get("nonfinite", (False, False)) == (True, False):
            return False, "same status and count, but the engine's iterate is NOT FINITE and the oracle's is", False
        if r["status"][1] in (1, 2) and not (r["dx"] <= 1e-8 and r["dy"] <= ytol):
            # Round 6 (campaign 721, Hessians of 1e-10): same status, same count, x apart by 1e-8 .. 2e-6 -- and the reference's OWN source moves its x by as
            # much when it is compiled with fused multiply-adds or -Ofast.  The solution is determined to kappa eps, not to 1e-8: accepted in a bucket of its
            # own iff the engine is no further from the plain oracle than TWICE the spread of the reference's builds (x and y each), else it fails as before.
            sx, sy = reference_builds_spread(p, st, warm)
            if r["dx"] <= max(1e-8, 2.0 * sx) and r["dy"] <= max(ytol, 2.0 * sy):
                return True, "ill-conditioned: dx %.3e dy %.3e against %.3e / %.3e between the reference's own builds" % (r["dx"], r["dy"], sx, sy), "conditioning"
            return False, "same count, x / y differ: dx %.3e dy %.3e (y bound %.3e; the reference's own builds spread %.3e / %.3e)" % (r["dx"], r["dy"], ytol, sx, sy), False
        return True, "", False
    nonfin = set()
    var = oracle_outcomes(p, st, warm, nonfin)
    plain = (r["status"][1], r["iter"][1])
    if r.get("nonfinite", (False, False))[0]:
        # Round 6: an engine iterate that is NOT FINITE is never "rounding".  Either the reference's own arithmetic does the same
        # on this case -- the plain oracle or its FMA / -Ofast build (NOT the variants that restate the engine's formulas) ends with a
        # non-finite iterate and the engine's status: H is singular by construction (an LP without the proximal term and fewer active rows
        # than variables: the factorisation divides by a pivot that is zero up to rounding) and there is no behaviour to match -- or it fails.
        ref_nonfin = (nonfin - set(ENGINE_FORM_VARIANTS)) | ({"plain"} if r["nonfinite"][1] else set())
        same_status = [k for k in ref_nonfin if (plain if k == "plain" else var[k])[0] == r["status"][0]]
        if same_status:
            return True, "singular H: engine %s not finite, like the oracle build(s) %s; plain %s, variants %s" % ((r["status"][0], r["iter"][0]), sorted(same_status), plain, var), "singular"
        return False, "engine iterate NOT FINITE (%s) where the plain oracle %s and its FMA / -Ofast builds %s stay finite" % ((r["status"][0], r["iter"][0]), plain, var), False
    if all(v == plain for v in var.values()):
        # no compiler-flag / formula variant flipped: is the count stable under a one-ulp perturbation of the DATA?  (If it is not, no
        # implementation with another -- equally valid -- rounding can be expected to reproduce it.)
        var.update(oracle_perturbed(p, st, warm))
    if all(v == plain for v in var.values()):
        # still the same: look at the two trajectories themselves
        if ctx is None:
            return False, "engine %s vs oracle %s, and every oracle variant agrees with the plain one %s" % ((r["status"][0], r["iter"][0]), plain, var), False
        okt, whyt = noise_decided_branch(ctx, p, st, warm)
        if not okt:
            return False, "engine %s vs oracle %s; oracle variants %s; trajectories: %s" % ((r["status"][0], r["iter"][0]), plain, var, whyt), False
        var["trajectory"] = whyt
    if r["status"][0] not in {plain[0]} | {v[0] for v in var.values()}:
        return False, "engine status %d is reached by no oracle variant (%s, %s)" % (r["status"][0], plain, var), True
    if r["status"][0] == 1 and r["status"][1] == 1:
        tol = 10.0 * max(st["eps_abs"], st["eps_rel"])
        if abs(r["obj"][0] - r["obj"][1]) > tol * max(1.0, abs(r["obj"][1])):
            return False, "rounding-decided case, but the objectives differ: %r" % (r["obj"],), True
    moved = {k for k, v in var.items() if k != "trajectory" and v != plain}
    # the variants that restate the engine's own recurrence are a bucket of their own: they show that a count depends on how
    # that recurrence is rounded, which is weaker than the reference's source changing its count under the compiler's flags
    cls = "engine-form" if (moved and moved <= set(ENGINE_FORM_VARIANTS)) else "rounding"
    return True, "%s: engine %s, oracle %s, variants %s" % ("rounding-decided" if cls == "rounding" else "decided by the form of the rank-update recurrence", (r["status"][0], r["iter"][0]), plain, var), cls


def noise_decided_branch(ctx, p, st, warm, cap=3000):
    """Runs the engine one iteration per launch next to the oracle's per-iteration trace and looks at the FIRST iteration whose kind
    (Newton / outer / forced outer / terminated) differs.  Returns (True, description) when up to there the iterates agree to 1e-6 and
    the differing decision is one taken on rounding noise: the inner dual residual ||dphi||_inf that decides "another Newton step or
    the outer update" (qpalm.c:515-520) is, in both implementations, below 1e-7 of what it was before the last Newton step."""
    import oracle.binding as ob
    from qpalm_amd.solver import QpalmBatch
    bt = QpalmBatch(ctx, [p], ctx.default_settings(**st))
    o = ob.OracleQP(*p.args(), settings=ob.default_settings(**st))
    if warm is not None:
        bt.warm_start(warm[0][None, :], warm[1][None, :])
        o.warm_start(warm[0], warm[1])
    o.enable_trace(cap)
    o.solve()
    t = o.trace()
    bt.begin_solve()
    try:
        prev_big, prev_kind_e = 0.0, -1
        for k in range(min(cap, len(t["kind"]))):
            bt.iterate(1)
            s, info = bt.stats(0), bt.info(0)
            ek, ok_ = int(s.last_kind), int(t["kind"][k])
            e2, o2 = float(info.dua2_res_norm), float(t["dua2_res_norm"][k])
            if ek != ok_:
                if {ek, ok_} <= {0, 1, 2} and max(e2, o2) <= 1e-7 * prev_big:
                    return True, "iteration %d: engine kind %d, oracle kind %d on inner residuals %.3e / %.3e (%.3e before the Newton step)" % (k, ek, ok_, e2, o2, prev_big)
                # Round 5: the same decision after a FULL Newton step on an unchanged active set.  The step before was a Newton step
                # with tau = 1 (the exact line search met no breakpoint) and the side that goes on with Newton steps finds nothing
                # entering or leaving: the iterate stayed on one quadratic piece of phi, on which x + d is its exact minimiser -- the
                # inner residual tested here is ZERO in exact arithmetic, what each implementation sees is the rounding error of its
                # own LDL' solve (on an indefinite matrix without pivoting easily 1e-6 of the residual before the step), whatever
                # its size.  The threshold then sits inside rounding noise by construction, no fitted constant involved.
                if {ek, ok_} <= {0, 1, 2} and k >= 1 and int(t["kind"][k - 1]) == 0 and prev_kind_e == 0 and abs(float(t["tau"][k - 1]) - 1.0) <= 1e-6:
                    newton_side_unchanged = (int(s.nb_enter) + int(s.nb_leave) == 0) if ek == 0 else (int(t["nb_enter"][k]) + int(t["nb_leave"][k]) == 0)
                    if newton_side_unchanged:
                        return True, ("iteration %d: engine kind %d, oracle kind %d after a full Newton step (tau = 1) on an unchanged active set: inner residuals %.3e / %.3e "
                                      "are the rounding error of the two solves (%.3e before the step)" % (k, ek, ok_, e2, o2, prev_big))
                if int(info.status_val) == -3 and ok_ == 0 and k >= 1 and int(t["kind"][k - 1]) == 0 and prev_kind_e == 0:
                    # the engine stops on the primal-infeasibility certificate where the oracle takes a Newton step: the oracle's vectors of
                    # this iteration from a second run that stops right after it (max_iter = k + 1; the Newton step leaves yh, Atyh, dphi alone)
                    o2_ = ob.OracleQP(*p.args(), settings=ob.default_settings(**dict(st, max_iter=k + 1)))
                    if warm is not None:
                        o2_.warm_start(warm[0], warm[1])
                    o2_.solve()
                    try:
                        return _certificate_on_rounding(st, k, float(t["tau"][k - 1]), int(t["nb_enter"][k]) + int(t["nb_leave"][k]),
                                                        _certificate_terms(lambda name: bt.vec(name, 0), p, st), _certificate_terms(o2_.vec, p, st), "engine")
                    finally:
                        o2_.cleanup()
                return False, "iteration %d: engine kind %d (dua2 %.3e), oracle kind %d (dua2 %.3e), before the step %.3e" % (k, ek, e2, ok_, o2, prev_big)
            if int(info.status_val) != -10:
                return False, "same kinds up to termination at iteration %d" % k
            dx = rel(bt.vec("x", 0)[:p.n], t["x"][k])
            if dx > 1e-6:
                return False, "iterates differ (%.2e) at iteration %d before any branch differs" % (dx, k)
            prev_kind_e = ek
            if ek == 0:
                prev_big = max(e2, o2)   # kind 0: a Newton step is taken from this residual
        k = len(t["kind"])
        if k < cap and k >= 1 and int(o.status_val) == -3 and int(t["kind"][k - 1]) == 0 and prev_kind_e == 0:
            # the oracle stopped on the primal-infeasibility certificate after these iterations: does the engine take one more Newton step?
            bt.iterate(1)
            s, info = bt.stats(0), bt.info(0)
            if int(info.status_val) == -10 and int(s.last_kind) == 0:
                return _certificate_on_rounding(st, k, float(t["tau"][k - 1]), int(s.nb_enter) + int(s.nb_leave),
                                                _certificate_terms(lambda name: bt.vec(name, 0), p, st), _certificate_terms(o.vec, p, st), "oracle")
        return False, "no differing branch found"
    finally:
        bt.close()
        o.cleanup()


def _certificate_terms(vec, p, st):
    """what is_primal_infeasible (termination.c:136-182) compares, from one implementation's workspace after the residual pass of an
    iteration: T = |Dinv o (A'yh - A'y)|_inf, thr = eps_prim_inf |E o (yh - y)|_inf, and |Dinv o dphi|_inf"""
    n, m = p.n, p.m
    scaled = int(st.get("scaling", 10)) > 0
    Dinv = vec("Dinv")[:n] if scaled else np.ones(n)
    E = vec("E")[:m] if scaled else np.ones(m)
    T = float(np.max(np.abs(Dinv * (vec("Atyh")[:n] - vec("Aty")[:n]))))
    thr = float(st.get("eps_prim_inf", 1e-5)) * float(np.max(np.abs(E * (vec("yh")[:m] - vec("y")[:m]))))
    return T, thr, float(np.max(np.abs(Dinv * vec("dphi")[:n])))


def _certificate_on_rounding(st, k, tau_prev, nchange, a, b, who_stops):
    """Round 5 (fresh-seed campaign, seed 501 case 280): one implementation stops on the primal-infeasibility certificate at iteration k,
    the other takes one more Newton step and stops an iteration later.  The certificate compares T = |A'(yh - y)| with eps |yh - y|,
    and A'yh = dphi - (Qx + q + (x - x0) / gamma): when the step before was a FULL Newton step (tau = 1) and the active set did not
    change, dphi is ZERO in exact arithmetic -- what each implementation holds in dphi is the rounding error of its own solve, and that
    error sits in T one to one.  So the two T's may differ by the sum of the two |dphi|, and a threshold between them is decided by that
    rounding error: accepted iff the thresholds agree, the threshold lies between the two T's, and |T_a - T_b| <= |dphi_a| + |dphi_b|.
    No fitted constant."""
    (Ta, thra, na), (Tb, thrb, nb) = a, b
    what = "iteration %d: the %s stops on the primal-infeasibility certificate, the other side takes a Newton step: T %.6e / %.6e against %.6e / %.6e, |dphi| %.3e / %.3e, tau before %.9f, %d active-set changes" % (
        k, who_stops, Ta, Tb, thra, thrb, na, nb, tau_prev, nchange)
    if abs(tau_prev - 1.0) > 1e-6 or nchange != 0:
        return False, what + " -- not after a full Newton step on an unchanged active set"
    if abs(thra - thrb) > 1e-6 * max(thra, thrb):
        return False, what + " -- the thresholds differ"
    lo, hi = min(Ta, Tb), max(Ta, Tb)
    if not (lo <= max(thra, thrb) and hi >= min(thra, thrb)):
        return False, what + " -- the threshold is not between the two values"
    if hi - lo > na + nb:
        return False, what + " -- the two values differ by more than the rounding error in dphi"
    return True, what + ": decided by the rounding error of the two solves"
