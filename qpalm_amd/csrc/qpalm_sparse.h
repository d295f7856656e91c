/*
 * qpalm_sparse.h -- the factor of Q + A' Sigma_act A (+ I / gamma) as a SPARSE L D L' (round 5; VERDICT round 4, row h).
 *
 * Replaces src/solver_interface.c:319-370, 505-541 (ldlchol / ldlcholQAtsigmaA / ldlsolveLD_neg_dphi: what the reference hands to
 * CHOLMOD's analyze + factorize + solve) for QPs whose Schur complement is sparse or has more rows than a dense panel allows
 * (8192): device memory per QP is proportional to nnz(L), not n^2.
 *
 *  - Symbolic analysis once per QP on the host (qpalm_capi.inc: sparse_analyze): pattern of Q + A'A with ALL rows of A (a superset
 *    of every active set's pattern, so the structure never changes while constraints enter and leave), elimination tree, pattern of
 *    L by columns (strict lower part, rows ascending), the same entries by rows (column + position in the column arrays, columns
 *    ascending) and the level sets of the elimination tree.  Natural ordering: what the reference configures
 *    (solver_interface.c:530-540, c->nmethods = 1, method[0].ordering = CHOLMOD_NATURAL) -- or, context option "sparse_ordering", a
 *    nested dissection (qpalm_capi.inc: sparse_order) under which the factor is that of P H P': the kernels below then work in the
 *    permuted numbering (perm, AtiP, QfiP, first) and the solve permutes its right-hand side in and out.
 *  - Numeric factorisation: left-looking by columns, the columns of one level of the elimination tree in parallel (one wavefront
 *    per column, lanes over the entries), a level ends with a workgroup barrier.  A column is assembled AND updated in a dense work
 *    vector of its wavefront (HBM, n doubles, zero outside of use): A' Sigma A first (active rows t of A(:, j) ascending, as
 *    form_schur), then Q, then 1 / gamma on the diagonal; then  w_i -= l_ik (l_jk d_k)  for the columns k of row j's structure,
 *    ascending; then d_j = w_j, l_ij = w_i / d_j.  Assembly and factorisation are one pass: H never exists in memory.
 *  - Triangular solves in place on the right-hand side in HBM: forward by rows (x_j = b_j - sum_k l_jk x_k, k ascending: the
 *    order of the column-oriented loop of cholmod_solve), levels ascending, one thread per row; D; backward by columns, levels
 *    descending.
 *  - Rows that enter or leave the active set: rank-1 updates / downdates along the row's path of the elimination tree (sp_updown) when
 *    the tree is bushy enough for that to be cheaper than a refactorisation (sp_update_pays), else a refactorisation.  Changes of
 *    sigma refactorise (the reference's own behaviour under FACTORIZE_KKT, iteration.c:135-144).
 * One workgroup per QP like the dense engine: batches of sparse QPs fill the chip; a single large sparse QP runs at the latency of
 * its level chain (a band matrix has n levels).  DESIGN.md section 2.
 */
#ifndef QPALM_SPARSE_H
#define QPALM_SPARSE_H

#ifndef QP_SP_LDS_FACTOR
#define QP_SP_LDS_FACTOR 1 /* (bisection builds: 0 compiles the LDS form of the factorisation / of the solves out) */
#endif
#ifndef QP_SP_LDS_SOLVE
#define QP_SP_LDS_SOLVE 1
#endif
/* measured on the two sparse batch workloads (profiles/r06/sparse_lds/variants.txt, QP/s banded / blocks): the phases as real functions with
 * local copies of the array pointers (QP_SP_LOCAL 1) 13277 / 15972 against 12409 / 14728 without; requesting the entries of the first batch
 * of rows of A (QP_SP_PREFETCH 1) and of columns of L (2) before the accumulators are set up LOSES (10878 / 15141, 11250 / 15349 with batches
 * of two): the registers it holds live spill under the 128-VGPR cap.  Round 5's forms (everything in HBM, inlined): 10324 / 14017. */
#ifndef QP_SP_LOCAL
#define QP_SP_LOCAL 1
#endif
#ifndef QP_SP_PREFETCH
#define QP_SP_PREFETCH 0
#endif
#ifndef QP_SPB
#define QP_SPB 4 /* contributing rows of A / columns of L whose entries the LDS form of sp_factor requests at once */
#endif
struct SpArrays { /* this QP's symbolic arrays + this slot's values and work vectors */
  const int *Lp, *Li, *Rp, *Rk, *Rpos, *levptr, *levcol;
  const int *perm, *AtiP, *QfiP, *first; /* the factor's numbering (P H P'): perm[new] = old; Ati and Qfi renumbered; every row of A's first column */
  int nlev;
  double *Lx, *Dg, *wv, *tmp;
  char *lds;     /* the workgroup's dynamic LDS (nothing else lives there while a factor operation runs) */
  int lds_bytes, lds_cap; /* lds_cap = context option "sparse_lds": 0 = the HBM forms below, 1 = the LDS forms where they fit, >= 2: a test's limit on the entries of a column the LDS form takes */
};
QPD SpArrays sp_arrays(const qpg_view &V, int b, int slot, double *Dg, char *lds) {
  SpArrays s;
  s.lds = lds; s.lds_bytes = V.lds_bytes; s.lds_cap = V.sp_lds;
  s.Lp = V.sp_Lp + (size_t)b * (V.n + 1); s.Li = V.sp_Li + (size_t)b * V.sp_nnzL;
  s.Rp = V.sp_Rp + (size_t)b * (V.n + 1); s.Rk = V.sp_Rk + (size_t)b * V.sp_nnzL; s.Rpos = V.sp_Rpos + (size_t)b * V.sp_nnzL;
  s.levptr = V.sp_levptr + (size_t)b * (V.n + 1); s.levcol = V.sp_levcol + (size_t)b * V.n;
  s.perm = V.sp_perm + (size_t)b * V.n; s.AtiP = V.sp_AtiP + (size_t)b * V.nnzA; s.QfiP = V.sp_QfiP + (size_t)b * V.nnzQf; s.first = V.sp_first + (size_t)b * V.m;
  s.nlev = V.sp_nlev[b];
  s.Lx = V.sp_Lx + (size_t)slot * V.sp_nnzL; s.Dg = Dg; s.wv = V.sp_wv + (size_t)slot * QP_NW * V.sp_gpw * V.n; s.tmp = V.sp_tmp + (size_t)slot * V.n;
  return s;
}

/* levels lev and lev + 1 both hold ONE column: the same wavefront (thread) handles both, in program order, and nobody else works */
QPD bool sp_level_needs_barrier(const SpArrays &S, int lev) {
  if (lev + 1 >= S.nlev) return true;
  return (S.levptr[lev + 1] - S.levptr[lev] > 1) || (S.levptr[lev + 2] - S.levptr[lev + 1] > 1);
}
/* H = Q (+ A' Sigma_act A) (+ I / gamma) assembled column by column and factorised in the same pass (see the header).
 * with_AtSA = false, proximal = false: the second resident factor LD_Q of the dual objective (dev_solve, la == 7), into the value arrays the caller points S at. */
QPNI void sp_factor(const qpg_view &V, int b, const int n, const SpArrays &S_, bool with_AtSA, bool proximal, double gamma) {
  /* a column per GROUP of lanes: the columns of a sparse factor are short (a band: half a dozen entries), so a wavefront takes gpw = 1, 2,
   * 4 or 8 columns of the level at a time (8 lanes each at 8) and a 512-thread workgroup up to 64 -- every step of a column is a chain of
   * dependent HBM round trips, the groups' chains overlap.  Loops run to the wavefront's longest trip count with the other groups
   * masked off, so that the wavefront-level synchronisation points are met by every lane. */
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int gpw = V.sp_gpw, spg = 64 / gpw, gl = lane & (spg - 1), grp = wid * gpw + lane / spg, ngrp = QP_NW * gpw;
  const int *Ap = V.Ap + (size_t)b * (V.n + 1), *Ai = V.Ai + (size_t)b * V.nnzA;
  const int *Atp = V.Atp + (size_t)b * (V.m + 1);
  const double *Atss = V.Atss + (size_t)b * V.nnzA;
  const int *Ainv = V.Ainv + (size_t)b * V.nnzA;
  const int *Qfp = V.Qfp + (size_t)b * (V.n + 1);
  const double *Qfx = V.Qfx + (size_t)b * V.nnzQf;
  const int *active = V.active + (size_t)b * V.m;
#if QP_SP_LOCAL
  const SpArrays S = S_; /* (a real function: the fields of S_ would be re-read from the caller's frame after every store) */
  const int *Li = QP_UNIFORM_PTR(S.Li), *Lp = QP_UNIFORM_PTR(S.Lp), *Rp = QP_UNIFORM_PTR(S.Rp), *Rk = QP_UNIFORM_PTR(S.Rk), *Rpos = QP_UNIFORM_PTR(S.Rpos);
  const int *AtiP = QP_UNIFORM_PTR(S.AtiP), *QfiP = QP_UNIFORM_PTR(S.QfiP);
  double *Lx = QP_UNIFORM_PTR(S.Lx), *Dg = QP_UNIFORM_PTR(S.Dg);
#else
  const SpArrays &S = S_;
  const int *Li = S.Li, *Lp = S.Lp, *Rp = S.Rp, *Rk = S.Rk, *Rpos = S.Rpos, *AtiP = S.AtiP, *QfiP = S.QfiP;
  double *Lx = S.Lx, *Dg = S.Dg;
#endif
  double *w = S.wv + (size_t)grp * n; /* this group's work vector: zero on entry, zero again when the column is done */
  /* LDS form of a column (round 6): the column's accumulators live in the group's share of the dynamic LDS, indexed by POSITION in the
   * column's pattern (0 = the diagonal, 1 + e - e0 = entry e); the pattern's row indices sit next to them and a contribution to row i
   * finds its position by binary search there (rows ascending).  What a column costs is its chain of dependent HBM round trips: in the
   * HBM form every contribution is a read-modify-write of the work vector (index -> value -> store, one after the other because two
   * contributions may hit the same row); here the indices and values of QP_SPB contributing rows / columns are fetched at once and
   * applied in order on LDS.  Per entry the same operations in the same order as the HBM form. */
  const int cap_lds = (QP_SP_LDS_FACTOR && S.lds_cap) ? (((S.lds_bytes / ngrp) / 12) & ~1) : 0; /* entries per group: 8 B accumulator + 4 B row index */
  /* the LDS form pays while the entries of a contributing column fit one or two passes of the group (a position costs a binary search on LDS,
   * per pass; the HBM form's passes are independent read-modify-writes): by default columns of at most 2 spg entries take it, longer ones --
   * near-dense factors -- the HBM form (measured: campaign S at n = 257..420 with near-dense L ran 5 x slower with every column in LDS).
   * "sparse_lds" >= 2 sets the limit (tests; cap_lds stays the stride) */
  const int cap_want = (S.lds_cap >= 2) ? S.lds_cap : 2 * spg;
  const int cap = (cap_want < cap_lds) ? cap_want : cap_lds;
  double QP_LDS_AS *wl = QP_LDS_ARG(double, S.lds) + (size_t)grp * cap_lds;
  int QP_LDS_AS *ridx = (int QP_LDS_AS *)(QP_LDS_ARG(double, S.lds) + (size_t)ngrp * cap_lds) + (size_t)grp * cap_lds;
  const int gbase = lane & ~(spg - 1);
  __syncthreads();
  for (int lev = 0; lev < S.nlev; lev++) {
    const int c0 = S.levptr[lev], c1 = S.levptr[lev + 1];
    for (int cw = c0 + wid * gpw; cw < c1; cw += ngrp) { /* (the same for the lanes of a wavefront) */
      const int c = cw + lane / spg;
      const bool on = c < c1;
      const int j = on ? S.levcol[c] : 0;
      const int jo = on ? S.perm[j] : 0; /* column j of P H P' is column perm[j] of H; row indices through AtiP / QfiP */
      const int e0 = on ? S.Lp[j] : 0, e1 = on ? S.Lp[j + 1] : 0;
      if (cap > 0 && wave_imax(e1 - e0 + 1) <= cap) {
        const int len = e1 - e0;
        auto posof = [&](const int i) QP_ALWAYS_INLINE { /* position of row i (>= j, in the pattern) among the accumulators */
          if (i == j) return 0;
          int lo = 0, hi = len;
          while (lo < hi) { const int mid = (lo + hi) >> 1; if (ridx[mid] < i) lo = mid + 1; else hi = mid; }
          return lo + 1;
        };
        /* ---- everything the column reads that does not depend on its own sums is requested up front, level by level of the index
         * chains (pointers -> first chunk of indices -> what those point at -> the entries of the first QP_SPB rows of A and columns
         * of L), so that the column pays ~7 dependent round trips instead of one or two per contributing row / column ---- */
        const bool wA = on && with_AtSA;
        const int p0 = wA ? Ap[jo] : 0, p1 = wA ? Ap[jo + 1] : 0;
        const int k0 = on ? Qfp[jo] : 0, k1 = on ? Qfp[jo + 1] : 0;
        const int r0 = on ? Rp[j] : 0, r1 = on ? Rp[j + 1] : 0;
        const int f_li = (e0 + gl < e1) ? Li[e0 + gl] : 0;
        int f_t = -1, f_ainv = 0, f_qi = -1, f_k = -1, f_tpos = 0;
        double f_qx = 0.0;
        if (p0 + gl < p1) { f_t = Ai[p0 + gl]; f_ainv = Ainv[p0 + gl]; }
        if (k0 + gl < k1) { f_qi = QfiP[k0 + gl]; f_qx = Qfx[k0 + gl]; }
        if (r0 + gl < r1) { f_k = Rk[r0 + gl]; f_tpos = Rpos[r0 + gl]; }
        int f_tq0 = 0, f_tq1 = 0, f_tk1 = 0;
        double f_tv = 0.0, f_tl = 0.0, f_td = 0.0;
        if (f_t >= 0) { const int act = active[f_t]; const double v = Atss[f_ainv]; const int q0 = Atp[f_t], q1 = Atp[f_t + 1]; if (act) { f_tv = v; f_tq0 = q0; f_tq1 = q1; } }
        if (f_k >= 0) { f_tl = Lx[f_tpos]; f_td = Dg[f_k]; f_tk1 = Lp[f_k + 1]; }
        const int np = wave_imax(p1 - p0), nr = wave_imax(r1 - r0);
        /* a batch of QP_SPB rows of A: their entries, one per lane, requested together (load), applied in order (apply) */
        int a_iv[QP_SPB], a_q0[QP_SPB], a_q1[QP_SPB];
        double a_av[QP_SPB], a_vj[QP_SPB];
        bool a_over = false;
        auto A_load = [&](const int s0, const int cnt, const double tv, const int tq0, const int tq1) QP_ALWAYS_INLINE {
          int longest = 0;
#pragma unroll
          for (int u = 0; u < QP_SPB; u++) {
            const int src = gbase + ((s0 + u) & (spg - 1));
            const bool has = s0 + u < cnt;
            a_vj[u] = __shfl(tv, src);
            const int sq0 = __shfl(tq0, src), sq1 = __shfl(tq1, src); /* (every lane takes part in a shuffle) */
            a_q0[u] = has ? sq0 : 0; a_q1[u] = has ? sq1 : 0;
            const int q = a_q0[u] + gl;
            const bool valid = q < a_q1[u];
            a_iv[u] = valid ? AtiP[q] : -1;
            a_av[u] = valid ? Atss[q] : 0.0;
            longest = (a_q1[u] - a_q0[u] > longest) ? (a_q1[u] - a_q0[u]) : longest;
          }
          a_over = wave_imax(longest) > spg; /* a row of A with more entries than the group has lanes: the rest in a loop */
        };
        auto A_apply = [&]() QP_ALWAYS_INLINE {
#pragma unroll
          for (int u = 0; u < QP_SPB; u++) {
            if (a_iv[u] >= j) { const int p = posof(a_iv[u]); wl[p] += a_av[u] * a_vj[u]; }
            if (a_over)
              for (int q = a_q0[u] + gl + spg; q < a_q1[u]; q += spg) {
                const int i = AtiP[q];
                if (i >= j) { const int p = posof(i); wl[p] += Atss[q] * a_vj[u]; }
              }
            QP_WAVE_SYNC();
          }
        };
        /* a batch of QP_SPB columns k of row j's structure: their entries below row j */
        int l_iv[QP_SPB], l_ev[QP_SPB], l_k1[QP_SPB];
        double l_lx[QP_SPB], l_mk[QP_SPB], l_lj[QP_SPB];
        bool l_has[QP_SPB], l_over = false;
        auto L_load = [&](const int rb, const int s0, const int cnt, const double tl, const double td, const int tpos, const int tk1) QP_ALWAYS_INLINE {
          int longest = 0;
#pragma unroll
          for (int u = 0; u < QP_SPB; u++) {
            const int src = gbase + ((s0 + u) & (spg - 1));
            l_has[u] = (s0 + u < cnt) && (r0 + rb + s0 + u < r1);
            l_lj[u] = __shfl(tl, src);
            l_mk[u] = l_lj[u] * __shfl(td, src);
            const int pos = __shfl(tpos, src);
            const int sk1 = __shfl(tk1, src);
            l_k1[u] = l_has[u] ? sk1 : 0;
            l_ev[u] = pos + 1 + gl;
            const bool valid = l_has[u] && l_ev[u] < l_k1[u];
            l_iv[u] = valid ? Li[l_ev[u]] : -1;
            l_lx[u] = valid ? Lx[l_ev[u]] : 0.0;
            const int below = l_has[u] ? (l_k1[u] - pos - 1) : 0;
            longest = (below > longest) ? below : longest;
          }
          l_over = wave_imax(longest) > spg;
        };
        auto L_apply = [&]() QP_ALWAYS_INLINE {
#pragma unroll
          for (int u = 0; u < QP_SPB; u++) {
            if (l_has[u] && gl == 0) wl[0] = QP_FMA(-l_lj[u], l_mk[u], wl[0]);
            if (l_iv[u] >= 0) { const int p = posof(l_iv[u]); wl[p] = QP_FMA(-l_lx[u], l_mk[u], wl[p]); }
            if (l_over)
              for (int e = l_ev[u] + spg; e < l_k1[u]; e += spg) {
                const int p = posof(Li[e]);
                wl[p] = QP_FMA(-Lx[e], l_mk[u], wl[p]);
              }
            QP_WAVE_SYNC();
          }
        };
        const int cntA0 = (np < spg) ? np : spg, cntL0 = (nr < spg) ? nr : spg;
        if (QP_SP_PREFETCH >= 1 && np > 0) A_load(0, cntA0, f_tv, f_tq0, f_tq1);
        if (QP_SP_PREFETCH >= 2 && nr > 0) L_load(0, 0, cntL0, f_tl, f_td, f_tpos, f_tk1);
        /* ---- the pattern's row indices and zeroed accumulators ---- */
        for (int e = e0 + gl; e < e1; e += spg) { ridx[e - e0] = (e == e0 + gl) ? f_li : Li[e]; wl[e - e0 + 1] = 0.0; }
        if (on && gl == 0) wl[0] = 0.0;
        QP_WAVE_SYNC();
        /* ---- A' Sigma A, column j: active rows t ascending ---- */
        for (int pb = 0; pb < np; pb += spg) {
          int tq0 = f_tq0, tq1 = f_tq1;
          double tv = f_tv;
          if (pb > 0) {
            const int pm = p0 + pb + gl;
            tq0 = 0; tq1 = 0; tv = 0.0;
            if (pm < p1) {
              const int t = Ai[pm];
              if (active[t]) { tv = Atss[Ainv[pm]]; tq0 = Atp[t]; tq1 = Atp[t + 1]; }
            }
          }
          const int cnt = (np - pb < spg) ? (np - pb) : spg;
          for (int s0 = 0; s0 < cnt; s0 += QP_SPB) {
            if (QP_SP_PREFETCH < 1 || pb > 0 || s0 > 0) A_load(s0, cnt, tv, tq0, tq1);
            A_apply();
          }
        }
        /* ---- + Q(:, j), + 1 / gamma ---- */
        if (f_qi >= j) { const int p = posof(f_qi); wl[p] = f_qx + wl[p]; }
        for (int k = k0 + gl + spg; k < k1; k += spg) {
          const int i = QfiP[k];
          if (i >= j) { const int p = posof(i); wl[p] = Qfx[k] + wl[p]; }
        }
        QP_WAVE_SYNC();
        if (on && proximal && gl == 0) wl[0] += 1.0 / gamma;
        QP_WAVE_SYNC();
        /* ---- left-looking updates: every column k < j with l_jk != 0, ascending ---- */
        for (int rb = 0; rb < nr; rb += spg) {
          int tpos = f_tpos, tk1 = f_tk1;
          double tl = f_tl, td = f_td;
          if (rb > 0) {
            const int rm = r0 + rb + gl;
            tpos = 0; tk1 = 0; tl = 0.0; td = 0.0;
            if (rm < r1) { const int k = Rk[rm]; tpos = Rpos[rm]; tl = Lx[tpos]; td = Dg[k]; tk1 = Lp[k + 1]; }
          }
          const int cnt = (nr - rb < spg) ? (nr - rb) : spg;
          for (int s0 = 0; s0 < cnt; s0 += QP_SPB) {
            if (QP_SP_PREFETCH < 2 || rb > 0 || s0 > 0) L_load(rb, s0, cnt, tl, td, tpos, tk1);
            L_apply();
          }
        }
        /* ---- pivot and column ---- */
        const double dj = on ? wl[0] : 1.0;
        for (int e = e0 + gl; e < e1; e += spg) Lx[e] = wl[e - e0 + 1] / dj;
        if (on && gl == 0) Dg[j] = dj;
        QP_WAVE_SYNC();
        continue;
      }
      /* ---- A' Sigma A, column j: active rows t ascending, lanes over the entries of row t (distinct columns i: no conflicts) ---- */
      if (with_AtSA) {
        /* what a row costs is its chain of dependent round trips (row index -> active flag, value, row pointers -> entries), so the
         * group's lanes fetch that chain for spg rows of the column AT ONCE and the rows are then taken in order from the lanes */
        const int p0 = on ? Ap[jo] : 0, p1 = on ? Ap[jo + 1] : 0;
        const int np = wave_imax(p1 - p0);
        for (int pb = 0; pb < np; pb += spg) {
          const int pm = p0 + pb + gl;
          int tq0 = 0, tq1 = 0;
          double tv = 0.0;
          if (pm < p1) {
            const int t = Ai[pm];
            if (active[t]) { tv = Atss[Ainv[pm]]; tq0 = Atp[t]; tq1 = Atp[t + 1]; }
          }
          const int cnt = (np - pb < spg) ? (np - pb) : spg;
          for (int sidx = 0; sidx < cnt; sidx++) {
            const int src = (lane & ~(spg - 1)) + sidx;
            const double vj = __shfl(tv, src);
            const int q0 = __shfl(tq0, src), q1 = __shfl(tq1, src);
            const int nq = wave_imax(q1 - q0);
            if (nq == 0) continue; /* (the same for every lane of the wavefront: no row of this round is active) */
            for (int qq = gl; qq < nq; qq += spg) {
              const int q = q0 + qq;
              if (q < q1) {
                const int i = S.AtiP[q];
                if (i >= j) w[i] += Atss[q] * vj;
              }
            }
            QP_WAVE_SYNC();
          }
        }
      }
      /* ---- + Q(:, j) (both triangles are stored: the lower one of the permuted matrix is picked here), + 1 / gamma ---- */
      {
        const int k0 = on ? Qfp[jo] : 0, k1 = on ? Qfp[jo + 1] : 0;
        for (int k = k0 + gl; k < k1; k += spg) {
          const int i = S.QfiP[k];
          if (i >= j) w[i] = Qfx[k] + w[i];
        }
      }
      QP_WAVE_SYNC();
      if (on && proximal && gl == 0) w[j] += 1.0 / gamma;
      QP_WAVE_SYNC();
      /* ---- left-looking updates: every column k < j with l_jk != 0, ascending.  Column k was finished in an earlier level, so l_jk, d_k
       * and the column's extent can be fetched for spg of them at once (one lane each), then applied in order ---- */
      {
        const int r0 = on ? S.Rp[j] : 0, r1 = on ? S.Rp[j + 1] : 0;
        const int nr = wave_imax(r1 - r0);
        for (int rb = 0; rb < nr; rb += spg) {
          const int rm = r0 + rb + gl;
          int tpos = 0, tk1 = 0;
          double tl = 0.0, td = 0.0;
          if (rm < r1) { const int k = S.Rk[rm]; tpos = S.Rpos[rm]; tl = S.Lx[tpos]; td = S.Dg[k]; tk1 = S.Lp[k + 1]; }
          const int cnt = (nr - rb < spg) ? (nr - rb) : spg;
          for (int sidx = 0; sidx < cnt; sidx++) {
            const int src = (lane & ~(spg - 1)) + sidx;
            const double ljk = __shfl(tl, src);
            const double mk = ljk * __shfl(td, src);
            const int pos = __shfl(tpos, src), k1 = __shfl(tk1, src);
            const bool has = r0 + rb + sidx < r1;
            if (has && gl == 0) w[j] = QP_FMA(-ljk, mk, w[j]);
            for (int e = pos + 1 + gl; e < k1; e += spg) { /* rows below j of column k: they all belong to column j's pattern (k1 = 0 where the group has no such column) */
              const int i = S.Li[e];
              w[i] = QP_FMA(-S.Lx[e], mk, w[i]);
            }
            QP_WAVE_SYNC();
          }
        }
      }
      /* ---- pivot and column; the work vector goes back to zero ---- */
      const double dj = on ? w[j] : 1.0;
      QP_WAVE_SYNC();
      for (int e = e0 + gl; e < e1; e += spg) {
        const int i = S.Li[e];
        S.Lx[e] = w[i] / dj;
        w[i] = 0.0;
      }
      if (on && gl == 0) { S.Dg[j] = dj; w[j] = 0.0; }
      QP_WAVE_SYNC();
    }
    /* a run of one-column levels (a chain of the tree) is wavefront 0's alone: no workgroup barrier inside the run */
    if (sp_level_needs_barrier(S, lev)) __syncthreads();
  }
  __syncthreads();
}

/* max_j (C_jj + sum_{i != j} |C_ij|),  C = A' Sigma_act A  (gershgorin_max of nonconvex.c:185-210, used by boost_gamma): full
 * columns in the work vectors, one wavefront per column */
QPN double sp_gershgorin(const qpg_view &V, int b, const int n, const SpArrays &S, QpShared &Sh) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int *Ap = V.Ap + (size_t)b * (V.n + 1), *Ai = V.Ai + (size_t)b * V.nnzA;
  const int *Atp = V.Atp + (size_t)b * (V.m + 1), *Ati = V.Ati + (size_t)b * V.nnzA;
  const double *Atss = V.Atss + (size_t)b * V.nnzA;
  const int *Ainv = V.Ainv + (size_t)b * V.nnzA;
  const int *active = V.active + (size_t)b * V.m;
  double *w = S.wv + (size_t)wid * n;
  double gmax = -1e300;
  __syncthreads();
  for (int j = wid; j < n; j += QP_NW) {
    for (int p = Ap[j]; p < Ap[j + 1]; p++) {
      const int t = Ai[p];
      if (!active[t]) continue;
      const double vj = Atss[Ainv[p]];
      for (int q = Atp[t] + lane; q < Atp[t + 1]; q += 64) w[Ati[q]] += Atss[q] * vj;
      QP_WAVE_SYNC();
    }
    const double cjj = w[j];
    QP_WAVE_SYNC();
    if (lane == 0) w[j] = 0.0;
    QP_WAVE_SYNC();
    double rad = 0.0;
    for (int p = Ap[j]; p < Ap[j + 1]; p++) { /* every touched entry is summed once: it is cleared when it is first met */
      const int t = Ai[p];
      if (!active[t]) continue;
      for (int q = Atp[t] + lane; q < Atp[t + 1]; q += 64) { const int i = Ati[q]; rad += qabs(w[i]); w[i] = 0.0; }
      QP_WAVE_SYNC();
    }
    rad = wave_sum(rad);
    const double ub = cjj + rad;
    gmax = (ub > gmax) ? ub : gmax;
  }
  double vm[1] = {gmax}, vs[1] = {0.0};
  block_reduce<1, 0>(Sh, vm, vs);
  return vm[0];
}

/* Rank-1 updates (rows entering the active set) and downdates (rows leaving it) of the sparse factor, one row of A at a time:
 * CHOLMOD's updown semantics (solver_interface.c:407-441) on a FIXED pattern -- L's pattern was computed for all rows of A, so an
 * entering row never adds structure.  The entries of a row of A form a clique of H, hence lie on ONE path of the elimination tree:
 * the update walks that path from the row's first column to the root (parent(j) = first row index of column j's pattern), wavefront 0
 * alone, lanes over the entries of the column; the work vector is consumed (zeroed) on the way.  Per column Davis & Hager's method C1 in
 * the oracle's form (oq_dense_ldl_rank1: a = alpha +- w_j^2 / d_j, d_j <- d_j a / alpha, gamma = -+ w_j / (d_j a)), per entry
 * w_i <- w_i - w_j l_ij, l_ij <- l_ij - gamma w_i.  Columns off the path are not touched.  Cost: one chain step (a few dependent HBM
 * round trips) per path column -- cheap on bushy trees (block structure), hopeless on a chain (band matrix): sp_update_pays decides. */
/* (sp_update_pays: qpalm_iter.h, in front of dev_update_sigma_pre, which applies the same rule to changed penalties) */
QPNI void sp_updown(const qpg_view &V, int b, const int n, const SpArrays &S_, const int *up, int n_up, const int *dn, int n_dn) {
#if QP_SP_LOCAL
  const SpArrays S = S_;
#else
  const SpArrays &S = S_;
#endif
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int *Atp = V.Atp + (size_t)b * (V.m + 1);
  const double *Atss = V.Atss + (size_t)b * V.nnzA;
  const int *Lp = QP_UNIFORM_PTR(S.Lp), *Li = QP_UNIFORM_PTR(S.Li), *AtiP = QP_UNIFORM_PTR(S.AtiP);
  double *Lx = QP_UNIFORM_PTR(S.Lx), *Dg = QP_UNIFORM_PTR(S.Dg);
  /* the work vector in LDS where n doubles fit (round 6): w_j and the w_i of a path column are LDS accesses, so a column costs two dependent
   * HBM round trips (its extent and pivot; its entries) instead of four -- the next column of the path comes from lane 0's first entry */
  const bool in_lds = QP_SP_LDS_SOLVE && S.lds_cap && (size_t)n * sizeof(double) <= (size_t)S.lds_bytes;
  auto walk = [&](auto w) QP_ALWAYS_INLINE {
    for (int c = 0; c < n_up + n_dn; c++) {
      const bool update = c < n_up;
      const int t = update ? up[c] : dn[c - n_up];
      const int q0 = Atp[t], q1 = Atp[t + 1];
      if (q1 <= q0) continue;
      for (int q = q0 + lane; q < q1; q += 64) w[AtiP[q]] = Atss[q];
      QP_WAVE_SYNC();
      double alpha = 1.0;
      int j = S.first[t]; /* the row's first column in the factor's numbering: the path starts there */
      while (j >= 0) {
        const int e0 = Lp[j], e1 = Lp[j + 1];
        const double wj = w[j];
        double dj = Dg[j], a, gam;
        QP_WAVE_SYNC(); /* every lane has read w_j and d_j before lane 0 overwrites them below */
        if (update) { a = alpha + (wj * wj) / dj; dj *= a; gam = -wj / dj; }
        else        { a = alpha - (wj * wj) / dj; dj *= a; gam =  wj / dj; }
        dj /= alpha;
        alpha = a;
        int ifirst = -1; /* lane 0: the first row of the column's pattern = the parent = the next column of the path */
        for (int e = e0 + lane; e < e1; e += 64) {
          const int i = Li[e];
          const double lx = Lx[e];
          const double wi = w[i] - wj * lx;
          w[i] = wi;
          Lx[e] = lx - gam * wi;
          if (e == e0) ifirst = i;
        }
        if (lane == 0) { Dg[j] = dj; w[j] = 0.0; }
        QP_WAVE_SYNC();
        j = __shfl(ifirst, 0);
      }
    }
  };
  __syncthreads();
  if (in_lds) {
    double QP_LDS_AS *w = QP_LDS_ARG(double, S.lds);
    for (int i = threadIdx.x; i < n; i += QP_T) w[i] = 0.0; /* (the walks leave it zero again) */
    __syncthreads();
    if (wid == 0) walk(w);
  } else {
    if (wid == 0) walk(S.wv); /* wavefront 0's work vector in HBM: zero outside of use */
  }
  __syncthreads();
}

/* x <- (L D L')^-1 x.  The permuted right-hand side lives in LDS where it fits (xs: the gathers x[k] of a row / column are LDS reads and
 * the entries of L are requested QP_SPU at a time; the subtractions stay in the row's / column's order), else in HBM (S.tmp). */
#ifndef QP_SPU
#define QP_SPU 4
#endif
template <class XP> QPD void sp_solve_levels(const int n, const SpArrays &S, XP x) {
  for (int lev = 0; lev < S.nlev; lev++) { /* forward: a row needs the rows of its structure, which sit in earlier levels */
    for (int c = S.levptr[lev] + (int)threadIdx.x; c < S.levptr[lev + 1]; c += QP_T) {
      const int j = S.levcol[c];
      double v = x[j];
      int r = S.Rp[j];
      const int r1 = S.Rp[j + 1];
      for (; r + QP_SPU <= r1; r += QP_SPU) {
        int pp[QP_SPU], kk[QP_SPU];
        double lv[QP_SPU];
#pragma unroll
        for (int u = 0; u < QP_SPU; u++) { pp[u] = S.Rpos[r + u]; kk[u] = S.Rk[r + u]; }
#pragma unroll
        for (int u = 0; u < QP_SPU; u++) lv[u] = S.Lx[pp[u]];
#pragma unroll
        for (int u = 0; u < QP_SPU; u++) v -= lv[u] * x[kk[u]];
      }
      for (; r < r1; r++) v -= S.Lx[S.Rpos[r]] * x[S.Rk[r]];
      x[j] = v;
    }
    if (sp_level_needs_barrier(S, lev)) __syncthreads();
  }
  __syncthreads();
  for (int j = threadIdx.x; j < n; j += QP_T) x[j] = x[j] / S.Dg[j];
  __syncthreads();
  for (int lev = S.nlev - 1; lev >= 0; lev--) { /* backward: a column needs the rows of its pattern (ancestors: later levels) */
    for (int c = S.levptr[lev] + (int)threadIdx.x; c < S.levptr[lev + 1]; c += QP_T) {
      const int j = S.levcol[c];
      double v = x[j];
      int e = S.Lp[j];
      const int e1 = S.Lp[j + 1];
      for (; e + QP_SPU <= e1; e += QP_SPU) {
        int ii[QP_SPU];
        double lv[QP_SPU];
#pragma unroll
        for (int u = 0; u < QP_SPU; u++) { ii[u] = S.Li[e + u]; lv[u] = S.Lx[e + u]; }
#pragma unroll
        for (int u = 0; u < QP_SPU; u++) v -= lv[u] * x[ii[u]];
      }
      for (; e < e1; e++) v -= S.Lx[e] * x[S.Li[e]];
      x[j] = v;
    }
    if (lev == 0 || sp_level_needs_barrier(S, lev - 1)) __syncthreads();
  }
  __syncthreads();
}
QPNI void sp_solve(const int n, const SpArrays &S_, double *xo) {
#if QP_SP_LOCAL
  const SpArrays S = S_;
#else
  const SpArrays &S = S_;
#endif
  const bool in_lds = QP_SP_LDS_SOLVE && S.lds_cap && (size_t)n * sizeof(double) <= (size_t)S.lds_bytes;
  __syncthreads();
  if (in_lds) {
    double QP_LDS_AS *x = QP_LDS_ARG(double, S.lds);
    for (int j = threadIdx.x; j < n; j += QP_T) x[j] = xo[S.perm[j]]; /* P b; the result goes back as P' x */
    __syncthreads();
    sp_solve_levels(n, S, x);
    for (int j = threadIdx.x; j < n; j += QP_T) xo[S.perm[j]] = x[j];
  } else {
    double *x = S.tmp;
    for (int j = threadIdx.x; j < n; j += QP_T) x[j] = xo[S.perm[j]];
    __syncthreads();
    sp_solve_levels(n, S, x);
    for (int j = threadIdx.x; j < n; j += QP_T) xo[S.perm[j]] = x[j];
  }
  __syncthreads();
}

#endif
