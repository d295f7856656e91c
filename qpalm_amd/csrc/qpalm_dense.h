/*
 * qpalm_dense.h -- dense-panel LDL^T kernels (device functions, one workgroup per factor).
 *
 * The Schur matrix Q + I/gamma + A_a' S_a A_a of the reference's CHOLMOD path is factorised with
 * natural ordering (src/solver_interface.c:530-540); at the benchmark configuration its factor is
 * 77-97 % dense, so the factor is kept as ONE dense supernodal panel: column-major
 * n x n fp64 in HBM (leading dimension ld, strict lower part = L, unit diagonal implicit) plus the
 * pivot vector D.  All walks below are down columns => consecutive lanes touch consecutive
 * addresses.
 *
 *   form_schur      replaces cholmod_aat + cholmod_add           (solver_interface.c:389-392)
 *   dense_factor    replaces cholmod_analyze + factorize_p        (solver_interface.c:347-356)
 *   dense_updown    replaces cholmod_updown (multi-rank, +/-)     (solver_interface.c:415-421,433-439,496)
 *   dense_solve     replaces cholmod_solve(CHOLMOD_LDLt)          (solver_interface.c:516)
 */
#ifndef QPALM_DENSE_H
#define QPALM_DENSE_H

#ifdef QPALM_EMU
#define QP_WAVE_SYNC() emu_wave_sync()
#define QP_SCHED_BARRIER() do { } while (0)
#else
/* keeps the scheduler from hoisting a whole unrolled loop's LDS reads to the top (register blow-up) */
#define QP_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
#define QP_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#endif


#ifdef QPALM_EMU
QPD double qp_readlane(double v, int src) { return emu_exchange(v, src); }
#else
QPD double qp_readlane(double v, int src) { /* src is wave-uniform: v_readlane_b32 x2, result lives in SGPRs */
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
#endif

/* ---------------------------------------------------------------------------------------------
 * form_schur: H(:,j) for j = 0..n-1, lower triangle, written into the factor slot.
 *   H_ij = Q_ij + sum_{t active} F_it F_jt (+ 1/gamma on the diagonal), F = At_sqrt_sigma.
 * One wavefront assembles one column in an LDS column buffer: the active rows t of A(:,j) are
 * walked in ascending order (the order cholmod_aat accumulates in), lanes spread over the entries
 * of F(:,t) => conflict-free LDS adds, deterministic sums.
 * GERSH = true: no Q, full columns, returns max_j (C_jj + sum_{i!=j} |C_ij|)  (nonconvex.c:185-210).
 * ------------------------------------------------------------------------------------------- */
template <bool GERSH>
QPN double form_schur(const qpg_view &V, int b, double *Lslot, bool with_AtSA, bool proximal, double gamma,
                      QpShared &S, char *lds) {
  const int n = V.n, ld = V.ld;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int *Ap = V.Ap + (size_t)b * (n + 1), *Ai = V.Ai + (size_t)b * V.nnzA;
  const int *Atp = V.Atp + (size_t)b * (V.m + 1), *Ati = V.Ati + (size_t)b * V.nnzA;
  const double *Atss = V.Atss + (size_t)b * V.nnzA;
  const int *Ainv = V.Ainv + (size_t)b * V.nnzA;
  const int *Qp = V.Qp + (size_t)b * (n + 1), *Qi = V.Qi + (size_t)b * V.nnzQ;
  const double *Qx = V.Qx + (size_t)b * V.nnzQ;
  const int *active = V.active + (size_t)b * V.m;
  int ncb = V.lds_bytes / (8 * n);
  if (ncb > QP_NW) ncb = QP_NW;
  if (ncb < 1) ncb = 1; /* host guarantees lds_bytes >= 8n */
  double *buf = (double *)lds + (size_t)wid * n;
  double gmax = -1e300;
  __syncthreads();
  for (int j0 = 0; j0 < n; j0 += ncb) {
    const int j = j0 + wid;
    if (wid < ncb && j < n) {
      const int lo = GERSH ? 0 : j;
      for (int i = lo + lane; i < n; i += 64) buf[i] = 0.0;
      QP_WAVE_SYNC();
      if (with_AtSA) {
        for (int p = Ap[j]; p < Ap[j + 1]; p++) {
          const int t = Ai[p];
          if (!active[t]) continue;
          const int k0 = Atp[t], k1 = Atp[t + 1];
          const double vj = Atss[Ainv[p]]; /* F_jt: the entry of column t of A' that sits in row j */
          for (int kb = k0; kb < k1; kb += 64) {
            const int k = kb + lane;
            if (k < k1) {
              const int i = Ati[k];
              if (i >= lo) buf[i] += Atss[k] * vj;
            }
          }
          QP_WAVE_SYNC();
        }
      }
      if (!GERSH) {
        for (int k = Qp[j] + lane; k < Qp[j + 1]; k += 64) {
          const int i = Qi[k];
          if (i >= j) buf[i] = Qx[k] + buf[i];
        }
        QP_WAVE_SYNC();
        if (proximal && lane == 0) buf[j] += 1.0 / gamma;
        QP_WAVE_SYNC();
        for (int i = j + lane; i < n; i += 64) Lslot[(size_t)j * ld + i] = buf[i];
      } else {
        double rad = 0.0;
        for (int i = lane; i < n; i += 64) if (i != j) rad += qabs(buf[i]);
        rad = wave_sum(rad);
        const double ub = buf[j] + rad;
        gmax = (ub > gmax) ? ub : gmax;
      }
      QP_WAVE_SYNC();
    }
  }
  if (GERSH) {
    double vm[1] = {gmax}, vs[1] = {0.0};
    block_reduce<1, 0>(S, vm, vs);
    return vm[0];
  }
  __syncthreads();
  return 0.0;
}

/* ---------------------------------------------------------------------------------------------
 * dense_factor: in-place LDL^T of the lower triangle held in the slot (no pivoting, negative
 * pivots accepted like CHOLMOD's simplicial LDL^T).  Left-looking over block columns of 32.
 *
 *  (1) panel update  P = H(J:n, Jb) - L(J:n, 0:J) D L(Jb, 0:J)'  on the matrix cores
 *      (v_mfma_f64_16x16x4_f64): wavefront w owns the 16-row tiles w, w+NW, ... and both 16-column
 *      tiles of the block; operands are read straight from the column-major panel (lane l reads
 *      element [k + (l>>4)][base + (l&15)], i.e. four 128-byte column segments per fragment), the
 *      panel-column fragment is the A operand so that accumulator registers run down rows and the
 *      write-back is coalesced.
 *  (2) the 32 x 32 diagonal block is factorised by wavefront 0 (lane = row, registers + shuffles);
 *  (3) rows below the block are finished one row per thread with the block's L, D from LDS.
 * ------------------------------------------------------------------------------------------- */
#define QP_FNB 32
struct FactorLds {
  double Ld[QP_FNB][QP_FNB + 1];
  double dv[QP_FNB];
  double colbuf[QP_FNB];
};

/* Panel update of block column J for a wavefront that owns NTJ consecutive-strided 16-row tiles
 * (tile index J/16 + wid + NW*t): branch-free k loop, fragments of step k+4 are in flight while the
 * 2*NTJ MFMAs of step k execute. */
template <int NTJ>
QPD void factor_panel_update(double *L, const double *Dg, int n, int ld, int J) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int tile0 = J / 16 + wid;
  qp_double4 acc[NTJ][2];
  int rowc[NTJ];
#pragma unroll
  for (int t = 0; t < NTJ; t++) {
    const int row = (tile0 + t * QP_NW) * 16 + l15;
    rowc[t] = (row < n) ? row : (n - 1);
#pragma unroll
    for (int ct = 0; ct < 2; ct++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int col = J + ct * 16 + l4 + 4 * r;
        acc[t][ct][r] = (row < n && col < n && row >= col) ? L[(size_t)col * ld + row] : 0.0;
      }
  }
  const int rowp0 = (J + l15 < n) ? (J + l15) : (n - 1), rowp1 = (J + 16 + l15 < n) ? (J + 16 + l15) : (n - 1);
  if (J > 0) {
    double pa0, pa1, bv[NTJ];
    {
      const double *colk = L + (size_t)l4 * ld;
      const double dk = Dg[l4];
      pa0 = -(colk[rowp0] * dk); pa1 = -(colk[rowp1] * dk);
#pragma unroll
      for (int t = 0; t < NTJ; t++) bv[t] = colk[rowc[t]];
    }
#pragma unroll 1
    for (int k = 0; k < J; k += 4) {
      const int kn = (k + 4 < J) ? (k + 4) : k; /* last step re-reads its own fragments */
      const double *colk = L + (size_t)(kn + l4) * ld;
      const double dk = Dg[kn + l4];
      const double na0 = colk[rowp0], na1 = colk[rowp1];
      double nb[NTJ];
#pragma unroll
      for (int t = 0; t < NTJ; t++) nb[t] = colk[rowc[t]];
#pragma unroll
      for (int t = 0; t < NTJ; t++) {
        acc[t][0] = QP_MFMA_F64(pa0, bv[t], acc[t][0]);
        acc[t][1] = QP_MFMA_F64(pa1, bv[t], acc[t][1]);
      }
      pa0 = -(na0 * dk); pa1 = -(na1 * dk);
#pragma unroll
      for (int t = 0; t < NTJ; t++) bv[t] = nb[t];
    }
  }
#pragma unroll
  for (int t = 0; t < NTJ; t++) {
    const int row = (tile0 + t * QP_NW) * 16 + l15;
    if (row < n) {
#pragma unroll
      for (int ct = 0; ct < 2; ct++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int col = J + ct * 16 + l4 + 4 * r;
          if (col < n && row >= col) L[(size_t)col * ld + row] = acc[t][ct][r];
        }
    }
  }
}

template <int RPT>
QPN void dense_factor(double *L, double *Dg, int n, int ld, char *lds, int64_t *tdbg) {
  FactorLds &F = *(FactorLds *)lds;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int NB = QP_FNB;
  constexpr int NT = (RPT * QP_T / 16 + QP_NW - 1) / QP_NW; /* 16-row tiles per wavefront, at most */
  __syncthreads();
  long long tq0 = QP_CLOCK();
  for (int J = 0; J < n; J += NB) {
    const int jb = (n - J < NB) ? (n - J) : NB;
    /* ---- (1) panel update on the matrix cores ------------------------------------------------ */
    {
      const int ntiles = (n - J + 15) / 16;
      const int ntj = (ntiles + QP_NW - 1) / QP_NW; /* same for every wavefront */
      if (ntj <= 1) factor_panel_update<1>(L, Dg, n, ld, J);
      else if (ntj == 2) factor_panel_update<(NT >= 2 ? 2 : 1)>(L, Dg, n, ld, J);
      else if (ntj == 3) factor_panel_update<(NT >= 3 ? 3 : 1)>(L, Dg, n, ld, J);
      else if (ntj == 4) factor_panel_update<(NT >= 4 ? 4 : 1)>(L, Dg, n, ld, J);
      else if (ntj <= 6) factor_panel_update<(NT >= 6 ? 6 : 1)>(L, Dg, n, ld, J);
      else if (ntj <= 8) factor_panel_update<(NT >= 8 ? 8 : 1)>(L, Dg, n, ld, J);
      else if (ntj <= 12) factor_panel_update<(NT >= 12 ? 12 : 1)>(L, Dg, n, ld, J);
      else factor_panel_update<(NT >= 16 ? 16 : 1)>(L, Dg, n, ld, J);
    }
    __syncthreads();
    if (tid == 0) { const long long tq1 = QP_CLOCK(); tdbg[4] += tq1 - tq0; tq0 = tq1; }
    /* ---- (2) diagonal block: right-looking, lane = row (registers), columns exchanged via LDS - */
    if (wid == 0) {
      double p[QP_FNB];
#pragma unroll
      for (int c = 0; c < NB; c++) p[c] = (lane < jb && c <= lane) ? L[(size_t)(J + c) * ld + (J + lane)] : 0.0;
#pragma unroll
      for (int c = 0; c < NB; c++) {
        if (c < jb) {
          if (lane < NB) F.colbuf[lane] = p[c]; /* un-normalised column c */
          QP_WAVE_SYNC();
          const double dc = F.colbuf[c];
          const double lic = p[c] / dc;
#pragma unroll
          for (int c2 = c + 1; c2 < NB; c2++) p[c2] = QP_FMA(-lic, F.colbuf[c2], p[c2]);
          if (lane > c) p[c] = lic;
          QP_WAVE_SYNC();
        }
      }
      if (lane < jb) {
#pragma unroll
        for (int c = 0; c < NB; c++) {
          if (c < lane) { F.Ld[lane][c] = p[c]; L[(size_t)(J + c) * ld + (J + lane)] = p[c]; }
          if (c == lane) { F.dv[c] = p[c]; Dg[J + c] = p[c]; }
        }
      }
    }
    __syncthreads();
    if (tid == 0) { const long long tq1 = QP_CLOCK(); tdbg[5] += tq1 - tq0; tq0 = tq1; }
    /* ---- (3) rows below the block: l_ic = (p_ic - sum_{c1<c} u_ic1 l_c,c1) / d_c ---------------- */
#pragma unroll 1
    for (int i = J + jb + tid; i < n; i += QP_T) {
      double u[QP_FNB];
#pragma unroll
      for (int c = 0; c < NB; c++) u[c] = (c < jb) ? L[(size_t)(J + c) * ld + i] : 0.0;
#pragma unroll
      for (int c = 0; c < NB; c++) {
        if (c < jb) {
          double v = u[c];
#pragma unroll
          for (int c1 = 0; c1 < c; c1++) v = QP_FMA(-u[c1], F.Ld[c][c1], v);
          u[c] = v; /* un-normalised l*d */
        }
      }
#pragma unroll
      for (int c = 0; c < NB; c++)
        if (c < jb) L[(size_t)(J + c) * ld + i] = u[c] / F.dv[c];
    }
    __syncthreads();
    if (tid == 0) { const long long tq1 = QP_CLOCK(); tdbg[6] += tq1 - tq0; tq0 = tq1; }
  }
  __syncthreads();
}

/* ---------------------------------------------------------------------------------------------
 * dense_solve: x <- (L D L')^{-1} x.  The right-hand side lives in LDS (xs); L is streamed from
 * HBM exactly twice (forward + backward), every read a coalesced column segment.
 * ------------------------------------------------------------------------------------------- */
#define QP_SNB 32
struct SolveLds {
  double tile[QP_SNB][QP_SNB + 1];
  double part[QP_SNB];
};

QPN void dense_solve(const double *L, const double *Dg, int n, int ld, double *xg, char *lds, int lds_bytes) {
  SolveLds &T = *(SolveLds *)lds;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int NB = QP_SNB;
  double *xs = ((size_t)sizeof(SolveLds) + (size_t)n * 8 <= (size_t)lds_bytes) ? (double *)(lds + sizeof(SolveLds)) : xg;
  __syncthreads();
  if (xs != xg) for (int i = tid; i < n; i += QP_T) xs[i] = xg[i];
  /* forward: L y = b */
  for (int J = 0; J < n; J += NB) {
    const int jb = (n - J < NB) ? (n - J) : NB;
    for (int e = tid; e < jb * jb; e += QP_T) {
      const int c = e / jb, r = e % jb;
      if (r > c) T.tile[r][c] = L[(size_t)(J + c) * ld + (J + r)];
    }
    __syncthreads();
    if (wid == 0) { /* lane = row of the block; its row of L in registers, pivots broadcast by readlane */
      double v = (lane < jb) ? xs[J + lane] : 0.0;
      double trow[QP_SNB];
#pragma unroll
      for (int c = 0; c < NB; c++) trow[c] = (lane < jb && c < lane) ? T.tile[lane][c] : 0.0;
#pragma unroll
      for (int c = 0; c < NB; c++) {
        const double yc = qp_readlane(v, c);
        v = QP_FMA(-trow[c], yc, v);
      }
      if (lane < jb) xs[J + lane] = v;
    }
    __syncthreads();
    for (int i = J + jb + tid; i < n; i += QP_T) {
      double acc = xs[i];
      for (int g = 0; g < jb; g += 8) { /* 8 independent column loads in flight per row */
        double lv[8];
#pragma unroll
        for (int cc = 0; cc < 8; cc++) lv[cc] = (g + cc < jb) ? L[(size_t)(J + g + cc) * ld + i] : 0.0;
#pragma unroll
        for (int cc = 0; cc < 8; cc++) if (g + cc < jb) acc = QP_FMA(-lv[cc], xs[J + g + cc], acc);
      }
      xs[i] = acc;
    }
  }
  __syncthreads();
  for (int i = tid; i < n; i += QP_T) xs[i] = xs[i] / Dg[i];
  __syncthreads();
  /* backward: L' x = z */
  const int Jlast = ((n - 1) / NB) * NB;
  for (int J = Jlast; J >= 0; J -= NB) {
    const int jb = (n - J < NB) ? (n - J) : NB;
    for (int c = wid; c < jb; c += QP_NW) {
      double s = 0.0;
      const double *col = L + (size_t)(J + c) * ld;
#pragma unroll 4
      for (int i = J + jb + lane; i < n; i += 64) s = QP_FMA(col[i], xs[i], s);
      s = wave_sum(s);
      if (lane == 0) T.part[c] = s;
    }
    for (int e = tid; e < jb * jb; e += QP_T) {
      const int c = e / jb, r = e % jb;
      if (r > c) T.tile[r][c] = L[(size_t)(J + c) * ld + (J + r)];
    }
    __syncthreads();
    if (wid == 0) { /* lane = column of the block: holds L(J+c, J+lane) for c > lane */
      double v = (lane < jb) ? (xs[J + lane] - T.part[lane]) : 0.0;
      double tcol[QP_SNB];
#pragma unroll
      for (int c = 0; c < NB; c++) tcol[c] = (c < jb && lane < c) ? T.tile[c][lane] : 0.0;
#pragma unroll
      for (int c = NB - 1; c >= 0; c--) {
        const double xc = qp_readlane(v, c);
        v = QP_FMA(-tcol[c], xc, v);
      }
      if (lane < jb) xs[J + lane] = v;
    }
    __syncthreads();
  }
  if (xs != xg) for (int i = tid; i < n; i += QP_T) xg[i] = xs[i];
  __syncthreads();
}

/* ---------------------------------------------------------------------------------------------
 * dense_updown: L D L' <- L D L' + sum_r s_r w_r w_r'   (s_r = +1 update, -1 downdate), the columns
 * w_r being columns cols[r] of At_sqrt_sigma.  Up to K ranks are applied per sweep over the panel,
 * so the panel is read and written once per K ranks (16 B per entry and sweep).
 *
 * Per column j and rank r (Davis & Hager method C1), with alpha_r carried along the columns:
 *     p = s w_j^2/alpha ; d_new = d + p ; gamma = -s w_j/(alpha d_new) ; alpha <- alpha d_new/d
 *     for i > j:  w_i -= w_j l_ij ;  l_ij -= gamma w_i
 * Thread t owns rows t, t+QP_T, ... and keeps their K running w values in registers for the whole
 * sweep.  Per block column of 32: wavefront 0 runs the recurrence on the block's own rows (lane =
 * row, everything in registers; rank-indexed scalars are produced with lane = rank, one reciprocal
 * per column on the critical path, DPP row scans) and publishes the (w_j, gamma) table through LDS;
 * then every thread applies the table to its rows below the block with L streamed from HBM.
 * ------------------------------------------------------------------------------------------- */
#define QP_UNB 32
template <int K>
struct UpdownLds {
  double Ld[QP_UNB][QP_UNB + 1];
  double Wd[QP_UNB][K + 1];
  double cwg[QP_UNB][K][2]; /* (w_j, gamma) per column and rank, read as one 16-byte broadcast */
  double Wt[K];
  double dd[QP_UNB];
};

#ifdef QPALM_EMU
template <int N> QPD double qp_row_shr(double v) { /* DPP row_shr:N within rows of 16 lanes, zero fill */
  const int lane = threadIdx.x & 63;
  const bool ok = (lane & 15) >= N;
  const double r = emu_exchange(v, ok ? lane - N : lane);
  return ok ? r : 0.0;
}
#else
template <int N> QPD double qp_row_shr(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x110 + N, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x110 + N, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
#endif

template <int RPT, int K>
QPN void dense_updown(const qpg_view &V, int b, double *L, double *Dg, double *Wst, const int *cols, int n_up,
                      const int *cols_dn, int n_dn, QpShared &S, char *lds, int64_t *tdbg) {
  static_assert(K <= 16, "rank block must fit one DPP row");
  UpdownLds<K> &U = *(UpdownLds<K> *)lds;
  const int n = V.n, ld = V.ld, NB = QP_UNB;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int *Atp = V.Atp + (size_t)b * (V.m + 1), *Ati = V.Ati + (size_t)b * V.nnzA;
  const double *Atss = V.Atss + (size_t)b * V.nnzA;
  const int nr = n_up + n_dn;
  for (int r0 = 0; r0 < nr; r0 += K) {
    const int kk = (nr - r0 < K) ? (nr - r0) : K;
    __syncthreads();
    long long tq0 = QP_CLOCK();
    for (int e = tid; e < kk * n; e += QP_T) Wst[e] = 0.0;
    __syncthreads();
    int jmin = n;
    for (int r = wid; r < kk; r += QP_NW) {
      const int g = r0 + r;
      const int t = (g < n_up) ? cols[g] : cols_dn[g - n_up];
      for (int k = Atp[t] + lane; k < Atp[t + 1]; k += 64) {
        const int i = Ati[k];
        Wst[(size_t)r * n + i] = Atss[k];
        jmin = (i < jmin) ? i : jmin;
      }
    }
    jmin = block_imin(S, jmin);
    double w[RPT][K];
#pragma unroll
    for (int rr = 0; rr < RPT; rr++) {
      const int i = tid + rr * QP_T;
#pragma unroll
      for (int r = 0; r < K; r++) w[rr][r] = (i < n && r < kk) ? Wst[(size_t)r * n + i] : 0.0;
    }
    double alpha = 1.0, ialpha = 1.0; /* lane r of wavefront 0 carries alpha_r and 1/alpha_r */
    const int grank = r0 + lane;
    const double sg = (lane < kk) ? ((grank < n_up) ? 1.0 : -1.0) : 0.0;
    const int J0 = (jmin / NB) * NB;
    if (tid == 0) { const long long tq1 = QP_CLOCK(); tdbg[0] += tq1 - tq0; tq0 = tq1; }
    for (int J = J0; J < n; J += NB) {
      const int jb = (n - J < NB) ? (n - J) : NB;
#pragma unroll
      for (int rr = 0; rr < RPT; rr++) {
        const int i = tid + rr * QP_T;
        if (i >= J && i < J + jb) {
#pragma unroll
          for (int r = 0; r < K; r++) U.Wd[i - J][r] = w[rr][r];
        }
      }
      for (int e = tid; e < jb * jb; e += QP_T) {
        const int c1 = e / jb, c = e % jb;
        if (c > c1) U.Ld[c][c1] = L[(size_t)(J + c1) * ld + (J + c)];
      }
      if (tid < jb) U.dd[tid] = Dg[J + tid];
      if (jb < NB) for (int e = tid; e < (NB - jb) * K; e += QP_T) { U.cwg[jb + e / K][e % K][0] = 0.0; U.cwg[jb + e / K][e % K][1] = 0.0; }
      __syncthreads();
      if (wid == 0) {
        double wrow[K];
#pragma unroll
        for (int r = 0; r < K; r++) wrow[r] = (lane < jb) ? U.Wd[lane][r] : 0.0;
#pragma unroll
        for (int c1 = 0; c1 < NB; c1++) {
          if (c1 < jb) {
            const double lcur = (lane > c1 && lane < jb) ? U.Ld[lane][c1] : 0.0; /* issued early, used after the scalars */
            if (lane == c1) {
#pragma unroll
              for (int r = 0; r < K; r++) U.Wt[r] = wrow[r];
            }
            QP_WAVE_SYNC();
            /* rank-indexed scalars: lane = rank (lanes >= kk carry w = 0 => gamma = 0: exact no-ops) */
            const double wv = (lane < kk) ? U.Wt[lane] : 0.0;
            const double d0 = U.dd[c1];
            const double p = sg * wv * wv * ialpha;
            double incl = p;
            if (!(V.dbg_flags & 4)) {
            if (K > 1) incl += qp_row_shr<1>(incl);
            if (K > 2) incl += qp_row_shr<2>(incl);
            if (K > 4) incl += qp_row_shr<4>(incl);
            if (K > 8) incl += qp_row_shr<8>(incl);
            }
            const double excl = qp_row_shr<1>(incl);
            const double dnew = d0 + incl, dprev = d0 + excl;
            const int dbgf = V.dbg_flags;
            const double rdn = (dbgf & 1) ? dnew : 1.0 / dnew, rdp = (dbgf & 1) ? dprev : 1.0 / dprev;
            const double gam = -sg * wv * ialpha * rdn;
            if (lane < K) { U.cwg[c1][lane][0] = -wv; U.cwg[c1][lane][1] = -gam; } /* stored negated: plain FMAs below */
            alpha = alpha * dnew * rdp;
            ialpha = ialpha * dprev * rdn;
            if (lane == kk - 1) U.dd[c1] = dnew;
            QP_WAVE_SYNC();
            /* rows of the block: lane = row; all K coefficient pairs are fetched first, then the
             * l chain is one FMA per rank: l <- (1 + g w_j) l - g w_r ; w_r <- w_r - w_j l_r */
            if (!(V.dbg_flags & 2)) {
              double l = lcur;
#pragma unroll
              for (int rb = 0; rb < K; rb += 8) {
                double cw[8], cg[8];
#pragma unroll
                for (int r = 0; r < 8; r++) { cw[r] = (rb + r < K) ? U.cwg[c1][rb + r][0] : 0.0; cg[r] = (rb + r < K) ? U.cwg[c1][rb + r][1] : 0.0; }
#pragma unroll
                for (int r = 0; r < 8; r++) {
                  if (rb + r < K) {
                    const double ca = QP_FMA(cg[r], cw[r], 1.0);       /* 1 + gamma w_j */
                    const double t = cg[r] * wrow[rb + r];             /* -gamma w_r */
                    const double wn = QP_FMA(cw[r], l, wrow[rb + r]);  /* w_r - w_j l */
                    l = QP_FMA(ca, l, t);
                    if (lane > c1) wrow[rb + r] = wn;
                  }
                }
                QP_SCHED_BARRIER();
              }
              if (lane > c1 && lane < jb) U.Ld[lane][c1] = l;
            }
            QP_SCHED_BARRIER();
          }
        }
      }
      __syncthreads();
      if (tid == 0) { const long long tq1 = QP_CLOCK(); tdbg[1] += tq1 - tq0; tq0 = tq1; }
      for (int e = tid; e < jb * jb; e += QP_T) {
        const int c1 = e / jb, c = e % jb;
        if (c > c1) L[(size_t)(J + c1) * ld + (J + c)] = U.Ld[c][c1];
      }
      if (tid < jb) Dg[J + tid] = U.dd[tid];
      /* rows below the block: one column per iteration.  Branch-free body: rows that are not
       * below the block read/write a private dummy cell (column stride 0), the column 8 ahead is
       * prefetched into a register queue, and the (-w_j, -gamma) pairs are fetched half a column
       * ahead of their use (two 8-rank buffers), so neither HBM nor LDS latency is exposed. */
      bool any = false;
#pragma unroll
      for (int rr = 0; rr < RPT; rr++) { const int i = tid + rr * QP_T; any = any || (i >= J + jb && i < n); }
      if (any) {
        double *dummy = Wst + (size_t)QPG_KMAX * n;
        double *rowp[RPT];
        size_t cstride[RPT];
        double q[RPT][8];
#pragma unroll
        for (int rr = 0; rr < RPT; rr++) {
          const int i = tid + rr * QP_T;
          const bool ok = (i >= J + jb && i < n);
          rowp[rr] = ok ? (L + (size_t)J * ld + i) : (dummy + tid + rr * QP_T);
          cstride[rr] = ok ? (size_t)ld : 0;
#pragma unroll
          for (int cc = 0; cc < 8; cc++) q[rr][cc] = rowp[rr][(size_t)((cc < jb) ? cc : jb - 1) * cstride[rr]];
        }
        constexpr int KH = (K + 1) / 2;
        double ca[KH][2], cb[KH][2];
#pragma unroll
        for (int r = 0; r < KH; r++) { ca[r][0] = U.cwg[0][r][0]; ca[r][1] = U.cwg[0][r][1]; }
#pragma unroll 1
        for (int c1 = 0; c1 < jb; c1++) {
          double l[RPT];
          const int cpre = (c1 + 8 < jb) ? c1 + 8 : jb - 1;
#pragma unroll
          for (int rr = 0; rr < RPT; rr++) {
            l[rr] = q[rr][0];
#pragma unroll
            for (int cc = 0; cc < 7; cc++) q[rr][cc] = q[rr][cc + 1];
            q[rr][7] = rowp[rr][(size_t)cpre * cstride[rr]];
          }
#pragma unroll
          for (int r = 0; r < K - KH; r++) { cb[r][0] = U.cwg[c1][KH + r][0]; cb[r][1] = U.cwg[c1][KH + r][1]; }
          QP_SCHED_BARRIER();
#pragma unroll
          for (int r = 0; r < KH; r++) {
#pragma unroll
            for (int rr = 0; rr < RPT; rr++) {
              w[rr][r] = QP_FMA(ca[r][0], l[rr], w[rr][r]);
              l[rr] = QP_FMA(ca[r][1], w[rr][r], l[rr]);
            }
          }
          QP_SCHED_BARRIER();
          const int cn = (c1 + 1 < NB) ? c1 + 1 : c1;
#pragma unroll
          for (int r = 0; r < KH; r++) { ca[r][0] = U.cwg[cn][r][0]; ca[r][1] = U.cwg[cn][r][1]; }
          QP_SCHED_BARRIER();
#pragma unroll
          for (int r = 0; r < K - KH; r++) {
#pragma unroll
            for (int rr = 0; rr < RPT; rr++) {
              w[rr][KH + r] = QP_FMA(cb[r][0], l[rr], w[rr][KH + r]);
              l[rr] = QP_FMA(cb[r][1], w[rr][KH + r], l[rr]);
            }
          }
#pragma unroll
          for (int rr = 0; rr < RPT; rr++) rowp[rr][(size_t)c1 * cstride[rr]] = l[rr];
          QP_SCHED_BARRIER();
        }
      }
      __syncthreads();
      if (tid == 0) { const long long tq1 = QP_CLOCK(); tdbg[2] += tq1 - tq0; tq0 = tq1; }
    }
  }
  __syncthreads();
}

#endif
