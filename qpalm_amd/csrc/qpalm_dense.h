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
#else
#define QP_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
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
          /* find F_jt */
          double vj = 0.0;
          for (int kb = k0; kb < k1; kb += 64) {
            const int k = kb + lane;
            const int hit = (k < k1) && (Ati[k] == j);
            const unsigned long long bal = __ballot(hit);
            const double cand = __shfl((k < k1) ? Atss[k] : 0.0, bal ? (__ffsll(bal) - 1) : 0);
            if (bal) vj = cand;
          }
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
 * pivots accepted like CHOLMOD's simplicial LDL^T).  Left-looking over block columns of NB:
 * thread t owns rows t, t+QP_T, ... (RPT of them) and keeps its NB-wide panel row in registers;
 * the (D L')-tile of the block rows is staged through LDS and broadcast.
 * ------------------------------------------------------------------------------------------- */
#define QP_FNB 16
#define QP_FKC 32
struct FactorLds {
  double Bt[QP_FKC][QP_FNB];
  double Ld[QP_FNB][QP_FNB + 1];
  double dv[QP_FNB];
};

template <int RPT>
QPN void dense_factor(double *L, double *Dg, int n, int ld, char *lds) {
  FactorLds &F = *(FactorLds *)lds;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int NB = QP_FNB;
  __syncthreads();
  for (int J = 0; J < n; J += NB) {
    const int jb = (n - J < NB) ? (n - J) : NB;
    double acc[RPT][QP_FNB];
#pragma unroll
    for (int r = 0; r < RPT; r++) {
      const int i = tid + r * QP_T;
#pragma unroll
      for (int c = 0; c < NB; c++) acc[r][c] = (i < n && c < jb && i >= J + c) ? L[(size_t)(J + c) * ld + i] : 0.0;
    }
    for (int k0 = 0; k0 < J; k0 += QP_FKC) {
      const int kc = (J - k0 < QP_FKC) ? (J - k0) : QP_FKC;
      __syncthreads();
      for (int e = tid; e < QP_FKC * NB; e += QP_T) {
        const int kk = e / NB, c = e % NB;
        F.Bt[kk][c] = (kk < kc && c < jb) ? L[(size_t)(k0 + kk) * ld + (J + c)] * Dg[k0 + kk] : 0.0;
      }
      __syncthreads();
      for (int kk = 0; kk < kc; kk++) {
        double a[RPT];
#pragma unroll
        for (int r = 0; r < RPT; r++) {
          const int i = tid + r * QP_T;
          a[r] = (i >= J && i < n) ? L[(size_t)(k0 + kk) * ld + i] : 0.0;
        }
#pragma unroll
        for (int c = 0; c < NB; c++) {
          const double bv = F.Bt[kk][c];
#pragma unroll
          for (int r = 0; r < RPT; r++) acc[r][c] = QP_FMA(-a[r], bv, acc[r][c]);
        }
      }
    }
    __syncthreads();
    /* rows of the diagonal block -> LDS */
#pragma unroll
    for (int r = 0; r < RPT; r++) {
      const int i = tid + r * QP_T;
      if (i >= J && i < J + jb) {
#pragma unroll
        for (int c = 0; c < NB; c++) F.Ld[i - J][c] = acc[r][c];
      }
    }
    __syncthreads();
    if (wid == 0) { /* unblocked LDL^T of the jb x jb block; lane = row, row kept in registers */
      double p[QP_FNB];
#pragma unroll
      for (int c = 0; c < NB; c++) p[c] = (lane < jb && c <= lane) ? F.Ld[lane][c] : 0.0;
#pragma unroll
      for (int c = 0; c < NB; c++) {
        if (c < jb) {
          const double dc = __shfl(p[c], c);
          const double lic = p[c] / dc;
#pragma unroll
          for (int c2 = c + 1; c2 < NB; c2++) {
            const double u = __shfl(p[c], c2);
            p[c2] = QP_FMA(-lic, u, p[c2]);
          }
          if (lane > c) p[c] = lic;
        }
      }
      QP_WAVE_SYNC();
      if (lane < jb) {
#pragma unroll
        for (int c = 0; c < NB; c++) {
          if (c < lane) { F.Ld[lane][c] = p[c]; L[(size_t)(J + c) * ld + (J + lane)] = p[c]; }
          if (c == lane) { F.dv[c] = p[c]; Dg[J + c] = p[c]; }
        }
      }
    }
    __syncthreads();
    /* panel rows below the block */
#pragma unroll
    for (int r = 0; r < RPT; r++) {
      const int i = tid + r * QP_T;
      if (i >= J + jb && i < n) {
#pragma unroll
        for (int c = 0; c < NB; c++) {
          if (c < jb) {
            double v = acc[r][c];
#pragma unroll
            for (int c1 = 0; c1 < c; c1++) v = QP_FMA(-acc[r][c1], F.Ld[c][c1], v);
            acc[r][c] = v; /* un-normalised l*d */
          }
        }
#pragma unroll
        for (int c = 0; c < NB; c++)
          if (c < jb) L[(size_t)(J + c) * ld + i] = acc[r][c] / F.dv[c];
      }
    }
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
    if (wid == 0) {
      double v = (lane < jb) ? xs[J + lane] : 0.0;
      for (int c = 0; c < jb; c++) {
        const double yc = __shfl(v, c);
        if (lane > c && lane < jb) v = QP_FMA(-T.tile[lane][c], yc, v);
      }
      if (lane < jb) xs[J + lane] = v;
    }
    __syncthreads();
    for (int i = J + jb + tid; i < n; i += QP_T) {
      double acc = xs[i];
      for (int c = 0; c < jb; c++) acc = QP_FMA(-L[(size_t)(J + c) * ld + i], xs[J + c], acc);
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
      for (int i = J + jb + lane; i < n; i += 64) s = QP_FMA(col[i], xs[i], s);
      s = wave_sum(s);
      if (lane == 0) T.part[c] = s;
    }
    for (int e = tid; e < jb * jb; e += QP_T) {
      const int c = e / jb, r = e % jb;
      if (r > c) T.tile[r][c] = L[(size_t)(J + c) * ld + (J + r)];
    }
    __syncthreads();
    if (wid == 0) {
      double v = (lane < jb) ? (xs[J + lane] - T.part[lane]) : 0.0;
      for (int c = jb - 1; c >= 0; c--) {
        const double xc = __shfl(v, c);
        if (lane < c) v = QP_FMA(-T.tile[c][lane], xc, v);
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
 * w_r being columns cols[r] of At_sqrt_sigma.  Up to K ranks are applied per sweep over the panel.
 *
 * Per column j and rank r (Davis & Hager C1, in the division-free-chain form):
 *     p = s w_j^2/alpha ; d_new = d + p ; gamma = -s w_j/(alpha d_new) ; alpha <- alpha d_new/d
 *     for i > j:  w_i -= w_j l_ij ;  l_ij -= gamma w_i
 * Thread t owns rows t, t+QP_T, ... and keeps their K running w values in registers for the whole
 * sweep; per block column the (w_j, gamma) table of the block is produced by wavefront 0 from the
 * block's own rows and broadcast through LDS.
 * ------------------------------------------------------------------------------------------- */
#define QP_UNB 16
struct UpdownLds {
  double Wd[QP_UNB][QPG_KMAX];
  double Ld[QP_UNB][QP_UNB + 1];
  double dd[QP_UNB];
  double cw[QP_UNB][QPG_KMAX];
  double cg[QP_UNB][QPG_KMAX];
};

template <int RPT, int K>
QPN void dense_updown(const qpg_view &V, int b, double *L, double *Dg, double *Wst, const int *cols, int n_up,
                      const int *cols_dn, int n_dn, QpShared &S, char *lds) {
  UpdownLds &U = *(UpdownLds *)lds;
  const int n = V.n, ld = V.ld, NB = QP_UNB;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int *Atp = V.Atp + (size_t)b * (V.m + 1), *Ati = V.Ati + (size_t)b * V.nnzA;
  const double *Atss = V.Atss + (size_t)b * V.nnzA;
  const int nr = n_up + n_dn;
  for (int r0 = 0; r0 < nr; r0 += K) {
    const int kk = (nr - r0 < K) ? (nr - r0) : K;
    __syncthreads();
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
    double alpha = 1.0; /* lane r of wavefront 0 carries alpha_r */
    const int grank = r0 + lane;
    const double sg = (lane < kk) ? ((grank < n_up) ? 1.0 : -1.0) : 0.0;
    const int J0 = (jmin / NB) * NB;
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
      __syncthreads();
      if (wid == 0) {
        for (int c1 = 0; c1 < jb; c1++) {
          /* A-step: lane = rank */
          const double wv = (lane < kk) ? U.Wd[c1][lane] : 0.0;
          const double p = (lane < kk) ? sg * wv * wv / alpha : 0.0;
          double incl = p;
#pragma unroll
          for (int o = 1; o < K; o <<= 1) { const double u = __shfl_up(incl, o); if (lane >= o) incl += u; }
          double excl = __shfl_up(incl, 1);
          if (lane == 0) excl = 0.0;
          const double d0 = U.dd[c1];
          const double dnew = d0 + incl, dprev = d0 + excl;
          if (lane < kk) {
            U.cw[c1][lane] = wv;
            U.cg[c1][lane] = -sg * wv / (alpha * dnew);
            alpha = alpha * dnew / dprev;
          }
          QP_WAVE_SYNC();
          if (lane == kk - 1) U.dd[c1] = dnew;
          /* B-step: lane = row of the diagonal block */
          if (lane > c1 && lane < jb) {
            double l = U.Ld[lane][c1];
#pragma unroll
            for (int r = 0; r < K; r++) {
              if (r < kk) {
                double wr = U.Wd[lane][r];
                wr = QP_FMA(-U.cw[c1][r], l, wr);
                l = QP_FMA(-U.cg[c1][r], wr, l);
                U.Wd[lane][r] = wr;
              }
            }
            U.Ld[lane][c1] = l;
          }
          QP_WAVE_SYNC();
        }
      }
      __syncthreads();
      for (int e = tid; e < jb * jb; e += QP_T) {
        const int c1 = e / jb, c = e % jb;
        if (c > c1) L[(size_t)(J + c1) * ld + (J + c)] = U.Ld[c][c1];
      }
      if (tid < jb) Dg[J + tid] = U.dd[tid];
#pragma unroll
      for (int rr = 0; rr < RPT; rr++) {
        const int i = tid + rr * QP_T;
        if (i >= J + jb && i < n) {
          for (int c1 = 0; c1 < jb; c1++) {
            double l = L[(size_t)(J + c1) * ld + i];
#pragma unroll
            for (int r = 0; r < K; r++) {
              if (r < kk) {
                w[rr][r] = QP_FMA(-U.cw[c1][r], l, w[rr][r]);
                l = QP_FMA(-U.cg[c1][r], w[rr][r], l);
              }
            }
            L[(size_t)(J + c1) * ld + i] = l;
          }
        }
      }
      __syncthreads();
    }
  }
  __syncthreads();
}

#endif
