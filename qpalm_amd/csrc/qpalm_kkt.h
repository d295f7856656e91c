/*
 * qpalm_kkt.h -- the KKT path of newton_set_direction (src/newton.c:22-95) on the dense-panel engine:
 *   qpalm_form_kkt / qpalm_reform_kkt + ladel_factorize*_with_diag      src/solver_interface.c:119-200, newton.c:32-45
 *   kkt_update_entering_constraints (ladel_row_add)                     src/solver_interface.c:202-218
 *   kkt_update_leaving_constraints  (ladel_row_del)                     src/solver_interface.c:220-236
 *   kkt_solve + iterative refinement                                    src/solver_interface.c:238-247, newton.c:55-90
 *
 * K = [[Q + I/gamma, A_a'], [A_a, -Sigma_a^{-1}]] of size n + m, inactive constraints = unit diagonal rows, natural
 * order, held as ONE dense column-major panel per resident workgroup like the Schur factor (LADEL keeps it sparse
 * with an AMD ordering; the quasi-definite system and therefore d are the same).  A row addition / deletion is the
 * bordering step of Davis & Hager (2005): a forward solve on the leading block, a mat-vec with the rows below, and ONE
 * rank-1 sweep over the trailing block (dense_updown with a prestaged vector); the factor is never rebuilt for it.
 * Included by qpalm_iter.h (needs QpArrays / IterShared).
 */
#ifndef QPALM_KKT_H
#define QPALM_KKT_H

/* dense lower triangle of K into the slot (qpalm_form_kkt / qpalm_reform_kkt) */
QPD void kkt_form(const qpg_view &V, const QpArrays &a, int b, double *L, double gamma, int prox) {
  const int n = a.n, m = a.m, np = n + m, ld = V.ld, tid = threadIdx.x;
  int *state = V.kkt_state + (size_t)b * V.m;
  __syncthreads();
  for (int j = 0; j < np; j++) /* zero the lower triangle, column by column (coalesced) */
    for (int i = j + tid; i < np; i += QP_T) L[(size_t)j * ld + i] = 0.0;
  __syncthreads();
  for (int j = tid; j < n; j += QP_T) {
    for (int k = a.Qp()[j]; k < a.Qp()[j + 1]; k++) { const int i = a.Qi()[k]; if (i >= j) L[(size_t)j * ld + i] += a.Qx()[k]; }
    if (prox) L[(size_t)j * ld + j] += 1.0 / gamma;
  }
  for (int k = tid; k < m; k += QP_T) {
    const int p = n + k, e0 = a.Atp()[k], e1 = a.Atp()[k + 1];
    if (a.active()[k]) {
      state[k] = 1;
      for (int e = e0; e < e1; e++) L[(size_t)a.Ati()[e] * ld + p] = a.Atx()[e];
      L[(size_t)p * ld + p] = (e1 > e0) ? -a.sigma_inv()[k] : 1.0;
    } else { state[k] = 0; L[(size_t)p * ld + p] = 1.0; }
  }
  __syncthreads();
}

/* ---- compact (re)factorisation (round 4) --------------------------------------------------------------------------------
 * An inactive constraint is a unit row / column of K: it takes no part in the elimination (LADEL's sparse factor never touches
 * it), but a dense LDL' of the whole (n+m) x (n+m) panel pays (n+m)^3/3 whatever the active set -- mpc-160: a 430-row
 * factorisation for ~60 active constraints, 4.4 of the 5.3 ms per warm-started solve (round 3).  So the solver's
 * "form + factorise" forms the matrix of the variables and the ACTIVE constraints only (same order: constraint rows ascending),
 * factorises that (n + n_a) x (n + n_a) panel in place, and spreads the factor out to the full layout afterwards: rows and
 * columns keep their order, nothing fills in between an inactive row and the rest, so the result IS the factor of the full K
 * (every entry gets the same terms in the same order; the terms it no longer gets are products with exact zeros).  The row
 * additions / deletions and the solves work on the full layout as before.  Used when the staging of QP_NW columns fits the LDS. */
QPD int kkt_form_compact(const qpg_view &V, const QpArrays &a, int b, double *L, double gamma, int prox, IterShared &I, int *list) {
  const int n = a.n, m = a.m, ld = V.ld, tid = threadIdx.x;
  int *state = V.kkt_state + (size_t)b * V.m;
  int na = 0, nd = 0;
  __syncthreads();
  block_compact2(I.S, m, [&](int i) { return a.active()[i] != 0; }, [&](int i) { return 0; }, list, list + m, na, nd); /* ascending */
  const int npc = n + na;
  for (int j = 0; j < npc; j++)
    for (int i = j + tid; i < npc; i += QP_T) L[(size_t)j * ld + i] = 0.0;
  __syncthreads();
  for (int j = tid; j < n; j += QP_T) {
    for (int k = a.Qp()[j]; k < a.Qp()[j + 1]; k++) { const int i = a.Qi()[k]; if (i >= j) L[(size_t)j * ld + i] += a.Qx()[k]; }
    if (prox) L[(size_t)j * ld + j] += 1.0 / gamma;
  }
  for (int k = tid; k < m; k += QP_T) state[k] = a.active()[k] ? 1 : 0;
  for (int c = tid; c < na; c += QP_T) {
    const int k = list[c], p = n + c, e0 = a.Atp()[k], e1 = a.Atp()[k + 1];
    for (int e = e0; e < e1; e++) L[(size_t)a.Ati()[e] * ld + p] = a.Atx()[e];
    L[(size_t)p * ld + p] = (e1 > e0) ? -a.sigma_inv()[k] : 1.0;
  }
  __syncthreads();
  return na;
}
/* factor of the compact panel (order n + na, in the slot's top left corner) -> factor of the full layout, in place: columns from
 * the last to the first, QP_NW at a time through LDS (a column moves to the right and its entries down, never the other way) */
QPD void kkt_expand(const qpg_view &V, const QpArrays &a, double *L, double *Dg, const int na, const int *list, char *lds) {
  const int n = a.n, m = a.m, np = n + m, npc = n + na, ld = V.ld, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  double *buf = (double *)lds + (size_t)wid * np; /* one column per wavefront */
  int *pos = (int *)((double *)lds + (size_t)QP_NW * np); /* pos[k] = compact row of constraint k, -1 = inactive */
  __syncthreads();
  for (int k = tid; k < m; k += QP_T) pos[k] = -1;
  __syncthreads();
  for (int c = tid; c < na; c += QP_T) pos[list[c]] = n + c;
  /* pivots: through the first column buffer */
  __syncthreads();
  double *dbuf = (double *)lds; /* wavefront 0's buffer */
  for (int i = tid; i < npc; i += QP_T) dbuf[i] = Dg[i];
  __syncthreads();
  for (int i = tid; i < np; i += QP_T) {
    const int ic = (i < n) ? i : pos[i - n];
    Dg[i] = (ic >= 0) ? dbuf[ic] : 1.0;
  }
  __syncthreads();
  for (int jb = npc - 1; jb >= 0; jb -= QP_NW) {
    const int jc = jb - wid;
    if (jc >= 0) for (int ic = jc + 1 + lane; ic < npc; ic += 64) buf[ic] = L[(size_t)jc * ld + ic];
    __syncthreads();
    if (jc >= 0) {
      const int jf = (jc < n) ? jc : n + list[jc - n];
      for (int i = jf + 1 + lane; i < np; i += 64) {
        const int ic = (i < n) ? i : pos[i - n];
        L[(size_t)jf * ld + i] = (ic >= 0) ? buf[ic] : 0.0;
      }
    }
    __syncthreads();
  }
  /* the columns of the inactive constraints: unit columns */
  for (int k = wid; k < m; k += QP_NW)
    if (pos[k] < 0) for (int i = n + k + 1 + lane; i < np; i += 64) L[(size_t)(n + k) * ld + i] = 0.0;
  __syncthreads();
}

/* r = b - K sol with b = [-dphi; 0] (newton.c:58-62,81-84); returns max |K sol| and max |r|.
 * The refinement loop of newton.c:64-90 stops on res <= max(1e-10 ref_norm, 1e-12), and with -1/sigma on the diagonal of a
 * quasi-definite K the residual of a good solution is rounding noise: how many refinement passes run -- and with them the
 * iterate -- depends on the order of the sums in mat_vec(kkt, sol).  So they are done in the REFERENCE'S order, one thread per
 * row, no fma (this file is compiled with -ffp-contract=off):
 *   rows 0..n-1 : the symmetric product with Q as cholmod_sdmult / ladel's symmetric mat-vec accumulate it (the entries of the
 *                 columns before the row first, then the column's own accumulator: lob_matvec's order), then + sol/gamma
 *                 (vec_mult_add_scaled, newton.c:59), then the entries of column j of A times the multipliers, constraint by
 *                 constraint in ascending order (the kkt columns n+k are walked after the Q columns);
 *   rows n+k    : the row of A times sol[0..n) in ascending column order, then the diagonal. */
QPD void kkt_residual(const qpg_view &V, const QpArrays &a, int b, IterShared &I, double gamma, int prox, double &norm_Ksol, double &norm_r) {
  const int n = a.n, m = a.m, tid = threadIdx.x;
  const size_t sk = (size_t)V.n + V.m; /* batch strides */
  const double *sol = V.kkt_sol + (size_t)b * sk;
  double *r = V.kkt_rhs + (size_t)b * sk;
  const int *state = V.kkt_state + (size_t)b * V.m;
  __syncthreads();
  const double ginv = 1.0 / gamma;
  double vm[2] = {0.0, 0.0}, vs[1] = {0.0};
  for (int j = tid; j < n; j += QP_T) {
    double up = 0.0, lo = 0.0;
    for (int k = a.Qfp()[j]; k < a.Qfp()[j + 1]; k++) {
      const int c = a.Qfi()[k];
      const double t = a.Qfx()[k] * sol[c];
      if (c < j) up += t; else lo += t;
    }
    double y = up + lo;
    if (prox) y = 1 * y + ginv * sol[j];
    for (int e = a.Ap()[j]; e < a.Ap()[j + 1]; e++) { /* rows of A in ascending order = kkt columns n+k in ascending order */
      const int k = a.Ai()[e];
      if (state[k] == 1) y += a.Ax()[e] * sol[n + k]; /* columns truncated by nz[] (state 0 / 2) do not couple */
    }
    double v = y * -1;
    vm[0] = qmax(vm[0], qabs(v));
    v = 1 * v + (-1) * a.dphi()[j];
    r[j] = v;
    vm[1] = qmax(vm[1], qabs(v));
  }
  for (int k = tid; k < m; k += QP_T) {
    const double xk = sol[n + k];
    const int st = state[k];
    double y;
    if (st == 1) {
      double acc = 0.0;
      const int e0 = a.Atp()[k], e1 = a.Atp()[k + 1];
      for (int e = e0; e < e1; e++) acc += a.Atx()[e] * sol[a.Ati()[e]];
      y = acc + ((e1 > e0) ? -a.sigma_inv()[k] : 1.0) * xk;
    } else if (st == 2) y = -a.sigma_inv()[k] * xk;
    else y = xk;
    const double v = y * -1;
    vm[0] = qmax(vm[0], qabs(v));
    r[n + k] = v;
    vm[1] = qmax(vm[1], qabs(v));
  }
  block_reduce<2, 0>(I.S, vm, vs);
  norm_Ksol = vm[0]; norm_r = vm[1];
}

/* The KKT branch.  action: 1 (re)form + factorise (qpalm_form_kkt / qpalm_reform_kkt + ladel_factorize*), 2 row additions for
 * enter[0 .. ne) then row deletions for leave[0 .. nl) (kkt_update_entering_constraints / kkt_update_leaving_constraints),
 * 3 form only, 4 factorise what the slot holds, 5 spread a compact factor out to the full layout, 0 keep.  flags: 1 kkt_solve (solver_interface.c:238-247), 2 the iterative
 * refinement of newton.c:57-90 on top of it.  One Newton step = (action, nb_enter, nb_leave, 3); the boundary operations of
 * include/qpalm_gfx950.h (qpg_kkt_*) call the pieces one by one. */
#define QP_KKT_SOLVE 1
#define QP_KKT_REFINE 2
template <int RPT>
QPNI void kkt_newton(const qpg_view *Vp, int b_, double *L, double *Dg, double *Wst, IterShared *Ip, char *lds, int action_, int ne_, int nl_, int flags_) {
  const qpg_view &V = *Vp;
  IterShared &I = *Ip;
  const int b = QP_UNIFORM(b_), action = QP_UNIFORM(action_), ne = QP_UNIFORM(ne_), nl = QP_UNIFORM(nl_), flags = QP_UNIFORM(flags_);
  const QpArrays a = qp_arrays(V, b);
  const qpg_settings &st = *V.settings;
  const int n = a.n, m = a.m, np = n + m, ld = V.ld, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, prox = qp_prox(st, I.s);
  const double gamma = I.s.gamma;
  const size_t sk = (size_t)V.n + V.m; /* batch strides */
  double *sol = V.kkt_sol + (size_t)b * sk, *rhs = V.kkt_rhs + (size_t)b * sk, *z = V.kkt_tmp + (size_t)b * sk;
  int *state = V.kkt_state + (size_t)b * V.m;
  /* The solver's "form + factorise" works on the active rows only (when the staging of the spreading step fits the LDS), and the
   * factor STAYS compact -- the solves run on it, half the rows -- until a row addition / deletion or a read of the factor needs
   * the full layout (kkt_expand then; I.s.kkt_na says which layout the slot holds). */
  int *list = V.kkt_list + (size_t)b * V.m;
  const bool can_compact = V.kkt_compact && ((size_t)QP_NW * np * sizeof(double) + (size_t)m * sizeof(int) <= (size_t)V.lds_bytes);
  int kna = QP_UNIFORM(I.s.kkt_na);
  if (action == 1 && can_compact) {
    kna = kkt_form_compact(V, a, b, L, gamma, prox, I, list);
    dense_factor<RPT>(L, Dg, n + kna, ld, lds, I.s.ticks_dbg);
  } else {
    if (kna >= 0 && (action == 2 || action == 5)) { kkt_expand(V, a, L, Dg, kna, list, lds); kna = -1; } /* 5: only make the layout full (a read of the factor) */
    if (action == 1 || action == 3) { kkt_form(V, a, b, L, gamma, prox); kna = -1; }
    if (action == 1 || action == 4) { dense_factor<RPT>(L, Dg, np, ld, lds, I.s.ticks_dbg); kna = -1; }
  }
  __syncthreads();
  if (tid == 0) I.s.kkt_na = kna;
  __syncthreads();
  if (action == 2) {
    for (int e = 0; e < ne + nl; e++) {
      const bool add = e < ne;
      const int k = QP_UNIFORM(add ? a.enter()[e] : a.leave()[e - ne]), p = n + k;
      __syncthreads();
      int up; /* sign of the trailing rank-1 term: L33 D33 L33' + (up ? + : -) w w' */
      if (add) { /* ladel_row_add(LD, sym, n+k, kkt, n+k, -sigma_inv[k]) */
        for (int j = tid; j < np; j += QP_T) z[j] = 0.0;
        __syncthreads();
        for (int q = a.Atp()[k] + tid; q < a.Atp()[k + 1]; q += QP_T) z[a.Ati()[q]] = a.Atx()[q];
        __syncthreads();
        dense_solve(L, Dg, p, ld, z, lds, V.lds_bytes, nullptr, 1); /* L11 z = k12 */
        double vm[1] = {0.0}, vs[1] = {0.0};
        for (int j = tid; j < p; j += QP_T) {
          const double l = z[j] / Dg[j];
          vs[0] += l * z[j];
          L[(size_t)j * ld + p] = l; /* row p of L */
        }
        block_reduce<0, 1>(I.S, vm, vs);
        const double d22 = -a.sigma_inv()[k] - vs[0];
        const double sq = QP_SQRT(qabs(d22));
        /* l32 = -(L31 z) / d22; its scaled copy is the rank-1 vector.  The mat-vec is spread over the whole workgroup: blocks of
         * 64 rows (lane = row: coalesced column segments), the p columns dealt to the wavefronts, their partial sums combined in
         * wavefront order through LDS (one thread per row walking all p columns was a cliff for KKT panels of thousands of rows) */
        double *part = (double *)lds; /* [QP_NW][64] */
        for (int i = tid; i <= p && i < np; i += QP_T) Wst[i] = 0.0;
        for (int i0 = p + 1; i0 < np; i0 += 64) {
          const int i = i0 + lane;
          const int ic = (i < np) ? i : np - 1;
          double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
          int j = wid;
          for (; j + 3 * QP_NW < p; j += 4 * QP_NW) { /* four independent column loads in flight per lane */
            const double l0 = L[(size_t)j * ld + ic], l1 = L[(size_t)(j + QP_NW) * ld + ic], l2 = L[(size_t)(j + 2 * QP_NW) * ld + ic], l3 = L[(size_t)(j + 3 * QP_NW) * ld + ic];
            acc0 = QP_FMA(l0, z[j], acc0); acc1 = QP_FMA(l1, z[j + QP_NW], acc1); acc2 = QP_FMA(l2, z[j + 2 * QP_NW], acc2); acc3 = QP_FMA(l3, z[j + 3 * QP_NW], acc3);
          }
          for (; j < p; j += QP_NW) acc0 = QP_FMA(L[(size_t)j * ld + ic], z[j], acc0);
          __syncthreads(); /* part[] of the previous block has been consumed */
          part[wid * 64 + lane] = (acc0 + acc1) + (acc2 + acc3);
          __syncthreads();
          if (wid == 0 && i < np) {
            double acc = part[lane];
            for (int w = 1; w < QP_NW; w++) acc += part[w * 64 + lane];
            const double l = -acc / d22;
            L[(size_t)p * ld + i] = l;
            Wst[i] = sq * l;
          }
        }
        if (tid == 0) { Dg[p] = d22; state[k] = 1; }
        up = QP_UNIFORM((int)(d22 < 0)); /* - l32 d22 l32' */
      } else { /* ladel_row_del(LD, sym, n+k) */
        const double d22 = Dg[p];
        __syncthreads();
        const double sq = QP_SQRT(qabs(d22));
        for (int i = tid; i < np; i += QP_T) {
          double wv = 0.0;
          if (i > p) { wv = sq * L[(size_t)p * ld + i]; L[(size_t)p * ld + i] = 0.0; }
          Wst[i] = wv;
        }
        for (int j = tid; j < p; j += QP_T) L[(size_t)j * ld + p] = 0.0;
        if (tid == 0) { Dg[p] = 1.0; state[k] = 2; }
        up = QP_UNIFORM((int)(d22 > 0)); /* + l32 d22 l32' */
      }
      __syncthreads();
      /* trailing block: one rank-1 sweep over the columns after p */
      if (p + 1 < np)
        dev_updown<RPT>(V, b, np, L, Dg, Wst, nullptr, up ? 1 : 0, nullptr, up ? 0 : 1, I.S, lds, I.s.ticks_dbg, nullptr, p + 1);
    }
    if (tid == 0) I.s.n_rank1 += ne + nl;
  }
  if (!(flags & QP_KKT_SOLVE)) { __syncthreads(); return; }
  /* kkt_solve (solver_interface.c:238-247).  On the compact factor: the right-hand side / solution of the listed constraints sit behind the
   * variables in a work vector; an inactive constraint's row of K is a unit row, its solution is its right-hand side. */
  __syncthreads();
  auto solve_in_place = [&](double *v) QP_ALWAYS_INLINE { /* v (full layout, n + m) <- K^{-1} v with the factor the slot holds */
    if (kna < 0) { dense_solve(L, Dg, np, ld, v, lds, V.lds_bytes, I.s.ticks_dbg); return; }
    double *w = V.kkt_rhs2 + (size_t)b * sk; /* compact right-hand side */
    __syncthreads();
    for (int j = tid; j < n + kna; j += QP_T) w[j] = (j < n) ? v[j] : v[n + list[j - n]];
    __syncthreads();
    dense_solve(L, Dg, n + kna, ld, w, lds, V.lds_bytes, I.s.ticks_dbg);
    for (int j = tid; j < n + kna; j += QP_T) { if (j < n) v[j] = w[j]; else v[n + list[j - n]] = w[j]; } /* the other constraints keep v = rhs (unit rows) */
    __syncthreads();
  };
  for (int j = tid; j < np; j += QP_T) sol[j] = (j < n) ? a.dphi()[j] * -1 : 0.0;
  __syncthreads();
  solve_in_place(sol);
  for (int j = tid; j < n; j += QP_T) a.d()[j] = sol[j];
  if (!(flags & QP_KKT_REFINE)) { __syncthreads(); return; }
  /* iterative refinement (newton.c:57-90; constants.h:101-103) */
  double nK, res;
  kkt_residual(V, a, b, I, gamma, prox, nK, res);
  double vm[1] = {0.0}, vs[1] = {0.0};
  for (int j = tid; j < n; j += QP_T) vm[0] = qmax(vm[0], qabs(a.dphi()[j]));
  block_reduce<1, 0>(I.S, vm, vs);
  const double ref_norm = qmax(nK, vm[0]);
  int kref = 0;
  while (kref < 3 && QP_UNIFORM((int)(res > qmax(1e-10 * ref_norm, 1e-12)))) {
    kref++;
    __syncthreads();
    for (int j = tid; j < np; j += QP_T) z[j] = rhs[j]; /* the correction solve works in place on a copy */
    __syncthreads();
    solve_in_place(z);
    for (int j = tid; j < np; j += QP_T) {
      const double dz = z[j];
      if (j < n) a.d()[j] = dz + 1 * a.d()[j];
      sol[j] = 1 * dz + 1 * sol[j];
    }
    kkt_residual(V, a, b, I, gamma, prox, nK, res);
  }
  __syncthreads();
}

#endif
