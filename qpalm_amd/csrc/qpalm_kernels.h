/*
 * qpalm_kernels.h -- __global__ entry points.  One workgroup of QP_T threads per QP everywhere.
 */
#ifndef QPALM_KERNELS_H
#define QPALM_KERNELS_H

#include "qpalm_device.h"

/* two workgroups of 8 wavefronts per CU = 4 wavefronts per SIMD => at most 128 VGPRs */
#ifndef QP_WAVES_PER_SIMD
#define QP_WAVES_PER_SIMD 4
#endif
#ifdef QPALM_EMU
#define QP_OCCUPANCY
#else
#define QP_OCCUPANCY __attribute__((amdgpu_waves_per_eu(QP_WAVES_PER_SIMD, QP_WAVES_PER_SIMD)))
#endif
#define QP_OP_MATVEC_A 1
#define QP_OP_MATVEC_Q 2
#define QP_OP_MATTVEC_A 3
#define QP_OP_LDLCHOL 4
#define QP_OP_LDLCHOL_QATSA 5
#define QP_OP_UPDATE_ENTER 6
#define QP_OP_DOWNDATE_LEAVE 7
#define QP_OP_UPDATE_SIGMA 8
#define QP_OP_SOLVE 9
#define QP_OP_RESIDUALS 10
#define QP_OP_ACTIVE 11
#define QP_OP_LINESEARCH 12
#define QP_OP_FACTOR_LOADED 13
#define QP_OP_KKT_FORM 14
#define QP_OP_KKT_FACTOR 15
#define QP_OP_KKT_ENTER 16
#define QP_OP_KKT_LEAVE 17
#define QP_OP_KKT_SOLVE 18
#define QP_OP_KKT_EXPAND 19 /* a compact KKT factor spread out to the full (n+m) layout (before the host reads it) */

/* qpalm_setup's device part: Ruiz scaling (scaling.c:34-113) and derived copies.
 * mode 0: fresh setup (nscale iterations); mode 1: qpalm_update_settings with more scaling
 * iterations (qpalm.c:753-772): the new factors are composed with the stored ones. */
__global__ __launch_bounds__(QP_T) void k_setup(qpg_view V, int nscale, int mode) {
  __shared__ IterShared I;
  for (int b = blockIdx.x; b < V.B; b += gridDim.x) {
    const QpArrays a = qp_arrays(V, b);
    __syncthreads();
    if (threadIdx.x == 0) I.s = V.sc[b];
    __syncthreads();
    if (nscale > 0) {
      double c_temp = 1.0;
      if (mode == 1) {
        const int had = I.s.has_scaling;
        c_temp = had ? I.s.sc_c : 1.0;
        for (int j = threadIdx.x; j < a.n; j += QP_T) a.dphi_prev()[j] = had ? a.D()[j] : 1.0;
        for (int i = threadIdx.x; i < a.m; i += QP_T) a.ls_delta()[i] = had ? a.E()[i] : 1.0;
        __syncthreads();
      }
      if (threadIdx.x == 0) I.s.has_scaling = 1;
      __syncthreads();
      dev_scale_data(V, a, b, nscale, I);
      if (mode == 1) {
        for (int j = threadIdx.x; j < a.n; j += QP_T) { const double dj = a.D()[j] * a.dphi_prev()[j]; a.D()[j] = dj; a.Dinv()[j] = 1.0 / dj; }
        for (int i = threadIdx.x; i < a.m; i += QP_T) { const double ei = a.E()[i] * a.ls_delta()[i]; a.E()[i] = ei; a.Einv()[i] = 1.0 / ei; }
        __syncthreads();
        if (threadIdx.x == 0) { I.s.sc_c *= c_temp; I.s.sc_cinv = 1 / I.s.sc_c; I.s.kkt_first = 1; /* qpalm.c:774 */ }
        __syncthreads();
      }
    }
    dev_fill_derived(V, a, b);
    if (threadIdx.x == 0) V.sc[b] = I.s;
    __syncthreads();
  }
}


/* =============================================================================================
 * set_settings_nonconvex + lobpcg (src/nonconvex.c:29-183), one workgroup per QP, at setup time (qpalm.c:293-296).
 * LOBPCG stops on ||A x - lambda x||_inf < 1e-5, so its iteration count -- and with it gamma = 1/|lambda| and every
 * later iterate -- depends on the last bits of its dot products.  This is a setup-time routine, so the sums are done
 * in the REFERENCE'S ORDER: vec_prod in groups of four (lin_alg.c:72-86; the groups in parallel, their total by one
 * thread), the symmetric mat-vec row by row in cholmod_sdmult's accumulation order, element-wise updates without fma.
 * The 2 x 2 / 3 x 3 (generalised) eigenproblems of the compressed pencil replace LAPACKE_dsyev / dsygv by a Cholesky
 * reduction + cyclic Jacobi in registers of thread 0.  The start vector is the C library's unseeded rand() sequence
 * (B9), restated on the host (glibc TYPE_3, seed 1) and uploaded into d.
 * =========================================================================================== */
struct LobpcgShared { double g[QP_T]; double y[3]; double lambda; double scal; };

QPD double lob_dot(LobpcgShared &S, const double *u, const double *v, int n) { /* vec_prod, lin_alg.c:72-86 */
  const int ng = n / 4, tid = threadIdx.x;
  double prod = 0.0; /* thread 0's running total: the groups are added in index order, QP_T of them per round */
  for (int k0 = 0; k0 < ng; k0 += QP_T) {
    __syncthreads();
    const int k = k0 + tid;
    if (k < ng) { const int i = 4 * k; S.g[tid] = (u[i] * v[i] + u[i + 1] * v[i + 1] + u[i + 2] * v[i + 2] + u[i + 3] * v[i + 3]); }
    __syncthreads();
    if (tid == 0) { const int cnt = (ng - k0 < QP_T) ? (ng - k0) : QP_T; for (int q = 0; q < cnt; q++) prod += S.g[q]; }
  }
  __syncthreads();
  if (tid == 0) {
    for (int i = 4 * ng; i < n; i++) prod += u[i] * v[i];
    S.scal = prod;
  }
  __syncthreads();
  return S.scal;
}
QPD double lob_norm2(LobpcgShared &S, const double *u, int n) { return QP_SQRT(lob_dot(S, u, u, n)); }
QPD void lob_matvec(const QpArrays &a, const double *x, double *y) { /* cholmod_sdmult on the lower triangle (stype -1) */
  __syncthreads();
  for (int r = threadIdx.x; r < a.n; r += QP_T) {
    double up = 0.0, lo = 0.0; /* Y[r] collects the columns before r; the column's own accumulator starts at zero */
    for (int k = a.Qfp()[r]; k < a.Qfp()[r + 1]; k++) {
      const int c = a.Qfi()[k];
      const double t = a.Qfx()[k] * x[c];
      if (c < r) up += t; else lo += t;
    }
    y[r] = up + lo;
  }
  __syncthreads();
}
/* smallest eigenpair of the symmetric-definite pencil (B, C) of order dim (2 or 3): C = G G', Jacobi on G^{-1} B G^{-T} */
QPD double lob_small_eig(int dim, double B[3][3], double Cm[3][3], double y[3]) {
  double G[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, M[3][3], T[3][3], Vv[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int j = 0; j < dim; j++) {
    double s = Cm[j][j];
    for (int k = 0; k < j; k++) s -= G[j][k] * G[j][k];
    G[j][j] = QP_SQRT(s);
    for (int i = j + 1; i < dim; i++) { double t = Cm[i][j]; for (int k = 0; k < j; k++) t -= G[i][k] * G[j][k]; G[i][j] = t / G[j][j]; }
  }
  for (int c = 0; c < dim; c++)
    for (int i = 0; i < dim; i++) { double t = B[i][c]; for (int k = 0; k < i; k++) t -= G[i][k] * T[k][c]; T[i][c] = t / G[i][i]; }
  for (int r = 0; r < dim; r++)
    for (int i = 0; i < dim; i++) { double t = T[r][i]; for (int k = 0; k < i; k++) t -= G[i][k] * M[r][k]; M[r][i] = t / G[i][i]; }
  for (int i = 0; i < dim; i++) for (int j = 0; j < i; j++) { const double av = 0.5 * (M[i][j] + M[j][i]); M[i][j] = av; M[j][i] = av; }
  for (int sweep = 0; sweep < 30; sweep++) {
    double off = 0;
    for (int i = 0; i < dim; i++) for (int j = 0; j < i; j++) off += M[i][j] * M[i][j];
    if (off == 0.0) break;
    for (int p = 0; p < dim; p++)
      for (int q = p + 1; q < dim; q++) {
        if (M[p][q] == 0.0) continue;
        const double theta = (M[q][q] - M[p][p]) / (2.0 * M[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (qabs(theta) + QP_SQRT(theta * theta + 1.0));
        const double cs = 1.0 / QP_SQRT(t * t + 1.0), sn = t * cs;
        for (int k = 0; k < dim; k++) { const double av = M[k][p], bv = M[k][q]; M[k][p] = cs * av - sn * bv; M[k][q] = sn * av + cs * bv; }
        for (int k = 0; k < dim; k++) { const double av = M[p][k], bv = M[q][k]; M[p][k] = cs * av - sn * bv; M[q][k] = sn * av + cs * bv; }
        for (int k = 0; k < dim; k++) { const double av = Vv[k][p], bv = Vv[k][q]; Vv[k][p] = cs * av - sn * bv; Vv[k][q] = sn * av + cs * bv; }
      }
  }
  int kmin = 0;
  for (int k = 1; k < dim; k++) if (M[k][k] < M[kmin][kmin]) kmin = k;
  for (int i = dim - 1; i >= 0; i--) {
    double t = Vv[i][kmin];
    for (int k = i + 1; k < dim; k++) t -= G[k][i] * y[k];
    y[i] = t / G[i][i];
  }
  return M[kmin][kmin];
}

__global__ __launch_bounds__(QP_T) void k_lobpcg(qpg_view V) {
  __shared__ LobpcgShared S;
  __shared__ IterShared I;
  const int tid = threadIdx.x;
  for (int b = blockIdx.x; b < V.B; b += gridDim.x) {
    const QpArrays a = qp_arrays(V, b);
    const int n = a.n;
    double *x = a.d(), *Ax = a.Qd(), *w = a.dphi(), *Aw = a.Atyh(), *p = a.temp_n(), *Ap = a.delta_x();
    __syncthreads();
    if (tid == 0) I.s = V.sc[b];
    __syncthreads();
    /* x <- rand()/RAND_MAX normalised (the host wrote the raw sequence into d) */
    { const double nx = lob_norm2(S, x, n); const double sc = 1.0 / nx; for (int i = tid; i < n; i += QP_T) x[i] *= sc; }
    lob_matvec(a, x, Ax);
    double lambda = lob_dot(S, x, Ax, n);
    for (int i = tid; i < n; i += QP_T) w[i] = Ax[i] + (-lambda) * x[i];
    { const double xw = lob_dot(S, x, w, n); for (int i = tid; i < n; i += QP_T) w[i] = w[i] + (-xw) * x[i]; }
    { const double nw = lob_norm2(S, w, n); const double sc = 1.0 / nw; for (int i = tid; i < n; i += QP_T) w[i] *= sc; }
    lob_matvec(a, w, Aw);
    double xAw = lob_dot(S, Aw, x, n), wAw = lob_dot(S, Aw, w, n);
    if (tid == 0) {
      double Bm[3][3] = {{lambda, xAw, 0}, {xAw, wAw, 0}, {0, 0, 0}}, Cm[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}, yv[3] = {0, 0, 0};
      S.lambda = lob_small_eig(2, Bm, Cm, yv);
      S.y[0] = yv[0]; S.y[1] = yv[1]; S.y[2] = yv[2];
    }
    __syncthreads();
    lambda = S.lambda;
    { const double y0 = S.y[0], y1 = S.y[1];
      for (int i = tid; i < n; i += QP_T) { p[i] = w[i] * y1; Ap[i] = Aw[i] * y1; x[i] = p[i] + y0 * x[i]; Ax[i] = Ap[i] + y0 * Ax[i]; } }
    int iters = 0;
    for (int it = 0; it < 1000; it++) {
      __syncthreads();
      double vm[1] = {0.0}, vs[1] = {0.0};
      for (int i = tid; i < n; i += QP_T) { const double wi = Ax[i] + (-lambda) * x[i]; w[i] = wi; vm[0] = qmax(vm[0], qabs(wi)); }
      block_reduce<1, 0>(I.S, vm, vs);
      if (vm[0] < 1e-5) { /* LOBPCG_TOL */
        const double norm_w = lob_norm2(S, w, n);
        lambda -= QP_SQRT(2.0) * norm_w + 1e-6;
        if (n <= 3) lambda -= 1e-6;
        break;
      }
      iters++;
      { const double xw = lob_dot(S, x, w, n); for (int i = tid; i < n; i += QP_T) w[i] = w[i] + (-xw) * x[i]; }
      { const double nw = lob_norm2(S, w, n); const double sc = 1.0 / nw; for (int i = tid; i < n; i += QP_T) w[i] *= sc; }
      lob_matvec(a, w, Aw);
      xAw = lob_dot(S, Ax, w, n);
      wAw = lob_dot(S, w, Aw, n);
      { const double pn = lob_norm2(S, p, n); const double pinv = 1.0 / pn; for (int i = tid; i < n; i += QP_T) { p[i] *= pinv; Ap[i] *= pinv; } }
      const double xAp = lob_dot(S, Ax, p, n), wAp = lob_dot(S, Aw, p, n), pAp = lob_dot(S, Ap, p, n);
      const double xp = lob_dot(S, x, p, n), wp = lob_dot(S, w, p, n);
      if (tid == 0) {
        double Bm[3][3] = {{lambda, xAw, xAp}, {xAw, wAw, wAp}, {xAp, wAp, pAp}}, Cm[3][3] = {{1, 0, xp}, {0, 1, wp}, {xp, wp, 1.0}}, yv[3];
        S.lambda = lob_small_eig(3, Bm, Cm, yv);
        S.y[0] = yv[0]; S.y[1] = yv[1]; S.y[2] = yv[2];
      }
      __syncthreads();
      lambda = S.lambda;
      const double y0 = S.y[0], y1 = S.y[1], y2 = S.y[2];
      for (int i = tid; i < n; i += QP_T) {
        const double pi = y2 * p[i] + y1 * w[i], api = y2 * Ap[i] + y1 * Aw[i];
        p[i] = pi; Ap[i] = api;
        x[i] = y0 * x[i] + 1 * pi; Ax[i] = y0 * Ax[i] + 1 * api;
      }
    }
    __syncthreads();
    if (tid == 0) { /* set_settings_nonconvex */
      I.s.lobpcg_lambda = lambda; I.s.lobpcg_iter = iters;
      if (lambda < 0) { I.s.nc_flag = 1; I.s.nc_gamma = 1 / qabs(lambda); I.s.gamma = I.s.nc_gamma; I.s.gamma_maxed = 1; }
      else I.s.nc_flag = 0;
      V.sc[b] = I.s;
    }
    __syncthreads();
    /* the vectors used as scratch go back to what qpalm_setup leaves in them (calloc'ed zeros) */
    for (int i = tid; i < n; i += QP_T) { x[i] = 0.0; Ax[i] = 0.0; w[i] = 0.0; Aw[i] = 0.0; p[i] = 0.0; Ap[i] = 0.0; }
    __syncthreads();
  }
}

__global__ __launch_bounds__(QP_T) void k_warm_start(qpg_view V, int has_x, int has_y) {
  __shared__ IterShared I;
  for (int b = blockIdx.x; b < V.B; b += gridDim.x) {
    const QpArrays a = qp_arrays(V, b);
    __syncthreads();
    if (threadIdx.x == 0) I.s = V.sc[b];
    __syncthreads();
    dev_warm_start(V, a, b, has_x, has_y, I);
    if (threadIdx.x == 0) {
      /* a new warm start begins a new solve (qpalm_solve re-initialises its locals) */
      I.s.done = 0; I.s.in_solve = 0;
      V.sc[b] = I.s;
    }
    __syncthreads();
  }
}

/* Which wavefront of this workgroup runs the serial chains of the update sweep.  The workgroups that share a compute
 * unit stay there for the whole launch (persistent kernel), and the hardware starts every workgroup's wavefront 0 on the
 * same SIMD: with "wavefront 0" as the panel wave of both, the two serial chains of a CU would take turns on one SIMD's
 * issue port while the other SIMDs wait for their tables.  So the k-th workgroup to arrive on a CU (atomic counter per
 * (XCC, SE, SH, CU) id, zeroed by the host before the launch) picks a wavefront by the SIMD it sits on. */
QPD void qp_place_panel_wave(const qpg_view &V, QpShared &S) {
  const int wid = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) S.hw_simd[wid] = QP_HW_SIMD();
  __syncthreads();
  if (threadIdx.x == 0) {
    const int key = QP_HW_CU_KEY() & (QPG_CU_KEYS - 1);
    const int arrival = atomicAdd(V.queue + 64 + key, 1);
    int pw = 0;
    /* The hardware deals the wavefronts of a workgroup to the SIMDs in the order 0, 2, 1, 3, 0, 2, 1, 3 (rotated by where
     * it starts; measured, tools/scratch/placement.py), so the wavefront after the panel wave -- the helper wave of the update sweep
     * when that variant is built, second in priority -- sits two SIMDs further.  Panel waves of the two workgroups of a CU
     * on SIMDs 0 and 1 keep all four serial wavefronts on SIMDs of their own. */
    const int target = (QP_NW >= 8) ? (arrival & 1) : (arrival & 3);
    for (int w = QP_NW - 1; w >= 0; w--) if (S.hw_simd[w] == target) pw = w;
    S.panel_wave = V.place_panel_wave ? pw : 0;
    S.placement = key << 16 | (arrival & 255) << 8 | S.panel_wave << 4 | S.hw_simd[0];
    /* Row ownership follows the SIMDs (place_panel_wave = 2): the sweep's wavefront w owns rows [w RPT 64, (w+1) RPT 64), which retire
     * in that order.  The panel waves of the two workgroups of a CU run on SIMDs 0 and 1: the wavefronts that share those SIMDs with
     * them get the ranks right after the panel wave, i.e. the rows that retire first, so that in the second half of every sweep the
     * serial chains have their SIMDs to themselves (measured alone vs loaded: 51 vs 57 ms per QP of panel-wave time).  Results do
     * not depend on who owns which rows. */
    if (V.place_panel_wave >= 2 && QP_NW >= 8) {
      int r = 0;
      S.wave_rank[S.panel_wave] = r++;
      const int s0 = S.hw_simd[S.panel_wave];
      for (int w = 0; w < QP_NW; w++) if (w != S.panel_wave && S.hw_simd[w] == s0) S.wave_rank[w] = r++;
      for (int w = 0; w < QP_NW; w++) if (S.hw_simd[w] == (s0 ^ 1)) S.wave_rank[w] = r++;
      for (int w = 0; w < QP_NW; w++) if ((S.hw_simd[w] | 1) != (s0 | 1)) S.wave_rank[w] = r++;
    } else {
      for (int w = 0; w < QP_NW; w++) S.wave_rank[w] = (w - S.panel_wave) & (QP_NW - 1);
    }
  }
  __syncthreads();
}

/* Longest-processing-time-first order of the work queue.  A launch of B > slots QPs ends with a tail in which part of the chip idles while the last
 * QPs finish (3 % of the default benchmark: 16 rounds of 512 QPs whose solves take 55 .. 150 ms); started in descending order of cost the tail is
 * made of the cheapest QPs.  The cost estimate is the kernel time of the member's previous solve (qpg_scalars.ticks_total: still in place when the
 * next solve is launched) -- what a receding-horizon sequence or a repeated parametric solve has; before the first solve every cost is zero and
 * the order is the index order.  rank = number of members that go first (ties by index), by counting: B^2 compares out of LDS, ~10 us at B = 8192.
 * Results do not depend on the order (tests/test_full_size.py: bit-identical through the queue). */
__global__ __launch_bounds__(QP_T) void k_queue_order(qpg_view V) {
  __shared__ long long cost[QP_T];
  const int b = blockIdx.x * QP_T + threadIdx.x;
  const long long mine = (b < V.B) ? V.sc[b].ticks_total : 0;
  int rank = 0;
  for (int c0 = 0; c0 < V.B; c0 += QP_T) {
    __syncthreads();
    cost[threadIdx.x] = (c0 + threadIdx.x < V.B) ? V.sc[c0 + threadIdx.x].ticks_total : -1;
    __syncthreads();
    const int lim = (V.B - c0 < QP_T) ? (V.B - c0) : QP_T;
    for (int k = 0; k < lim; k++) rank += (cost[k] > mine || (cost[k] == mine && c0 + k < b)) ? 1 : 0;
  }
  if (b < V.B) V.order[rank] = b;
}

/* The persistent solver: workgroup `blockIdx.x` owns factor slot `blockIdx.x` and pulls QPs either
 * statically (b = blockIdx.x, resumable, needs B <= grid) or from an atomic work queue (dynamic & 1).  dynamic & 2: a
 * fresh qpalm_solve -- QPs whose previous solve has finished start over. */
template <int RPT>
__global__ __launch_bounds__(QP_T) QP_OCCUPANCY void k_solve(qpg_view V, int budget, int dynamic) {
  __shared__ IterShared I;
  char *lds = QP_DYN_LDS();
  qp_place_panel_wave(V, I.S);
  /* one call site of dev_solve (= one copy of the iteration loop in the kernel): static round-robin
   * or the atomic work queue only differ in how the next QP index is obtained */
  int b = blockIdx.x - gridDim.x;
  while (true) {
    if (dynamic & 1) {
      __syncthreads();
      if (threadIdx.x == 0) I.S.ibc[0] = atomicAdd(V.queue, 1);
      __syncthreads();
      b = QP_UNIFORM(I.S.ibc[0]);
      if (b < V.B && V.order != nullptr) b = QP_UNIFORM(V.order[b]);
    } else b += gridDim.x;
    if (b >= V.B) break;
    dev_solve<RPT>(V, b, blockIdx.x, budget, dynamic & 2, I, lds);
  }
}

/* qpalm_update_bounds, device part (qpalm.c:819-826): the host wrote the raw bounds */
/* The raw bounds are staged in the line-search scratch (ls_key: [b][0..m) = bmin, [b][m..2m) = bmax; idle between solves).
 * Validation (bmin <= bmax, qpalm.c:806-817) happens here too: bad[b] = 1 leaves the QP's bounds untouched. */
__global__ __launch_bounds__(QP_T) void k_update_bounds(qpg_view V, int has_bmin, int has_bmax, int *bad) {
  __shared__ int s_bad;
  for (int b = blockIdx.x; b < V.B; b += gridDim.x) {
    const QpArrays a = qp_arrays(V, b);
    const double *smin = V.ls_key + (size_t)b * V.ls_stride, *smax = smin + V.m; /* batch stride m: rows of the host arrays */
    int mine = 0;
    if (has_bmin && has_bmax)
      for (int i = threadIdx.x; i < a.m; i += QP_T) mine |= (smin[i] > smax[i]) ? 1 : 0;
    __syncthreads();
    if (threadIdx.x == 0) s_bad = 0;
    __syncthreads();
    if (mine) s_bad = 1;
    __syncthreads();
    const int isbad = s_bad;
    if (threadIdx.x == 0) bad[b] = isbad;
    if (isbad) continue;
    const bool sc = V.sc[b].has_scaling != 0;
    for (int i = threadIdx.x; i < a.m; i += QP_T) {
      if (has_bmin) a.bmin()[i] = sc ? a.E()[i] * smin[i] : smin[i];
      if (has_bmax) a.bmax()[i] = sc ? a.E()[i] * smax[i] : smax[i];
    }
  }
}

/* qpalm_update_q, device part (qpalm.c:829-871): the host wrote the raw q */
__global__ __launch_bounds__(QP_T) void k_update_q(qpg_view V) {
  __shared__ IterShared I;
  const qpg_settings &st = *V.settings;
  for (int b = blockIdx.x; b < V.B; b += gridDim.x) {
    const QpArrays a = qp_arrays(V, b);
    __syncthreads();
    if (threadIdx.x == 0) I.s = V.sc[b];
    __syncthreads();
    if (I.s.has_scaling) {
      const double c_old = I.s.sc_c, cinv_old = I.s.sc_cinv, mg = -1 / I.s.gamma;
      double vm[1] = {0.0}, vs[1] = {0.0};
      for (int j = threadIdx.x; j < a.n; j += QP_T) {
        const double qj = a.D()[j] * a.q()[j];
        a.q()[j] = qj;
        double Qxj = a.Qxv()[j];
        if (qp_prox(st, I.s)) Qxj = Qxj + mg * a.x()[j];
        a.Qxv()[j] = Qxj;
        const double t = qj + cinv_old * Qxj;
        vm[0] = qmax(vm[0], qabs(t));
      }
      block_reduce<1, 0>(I.S, vm, vs);
      const double c = 1 / qmax(1.0, vm[0]);
      const double ratio = c / c_old;
      const double ginit = qp_gamma_init(st, I.s);
      for (int j = threadIdx.x; j < a.n; j += QP_T) {
        a.q()[j] *= c;
        double Qxj = a.Qxv()[j] * ratio;
        if (qp_prox(st, I.s)) Qxj = Qxj + (1 / ginit) * a.x()[j];
        a.Qxv()[j] = Qxj;
      }
      const int nzQ = a.Qp()[a.n], nzQf = a.Qfp()[a.n];
      for (int k = threadIdx.x; k < nzQ; k += QP_T) a.Qx()[k] *= ratio;
      __syncthreads();
      for (int k = threadIdx.x; k < nzQf; k += QP_T) a.Qfx()[k] = a.Qx()[a.Qfperm()[k]];
      if (threadIdx.x == 0) {
        I.s.sc_c = c; I.s.sc_cinv = 1 / c;
        if (qp_prox(st, I.s)) I.s.gamma = ginit;
        V.sc[b] = I.s;
      }
    }
    __syncthreads();
  }
}

/* compute_residuals alone (iteration.c:24-48), for the boundary surface */
QPN void dev_compute_residuals(const qpg_view &V, const QpArrays &a, IterShared &I) {
  const qpg_settings &st = *V.settings;
  const int n = a.n, m = a.m, tid = threadIdx.x;
  for (int i = tid; i < m; i += QP_T) {
    const double yv = a.y()[i], ax = a.Axv()[i];
    double t = yv * a.sigma_inv()[i];
    const double axys = ax + 1 * t;
    const double zz = qmax(a.bmin()[i], qmin(axys, a.bmax()[i]));
    const double pr = ax + (-1) * zz;
    t = pr * a.sigma()[i];
    a.Axys()[i] = axys; a.z()[i] = zz; a.pri_res()[i] = pr; a.yh()[i] = yv + 1 * t;
  }
  __syncthreads();
  spmv_rows<16>(n, a.Ap(), a.Ai(), a.Ax(), a.yh(), [&](int r, double s) { a.Atyh()[r] = s; });
  __syncthreads();
  const double mginv = -1 / I.s.gamma;
  for (int j = tid; j < n; j += QP_T) {
    double dfv = a.Qxv()[j] + 1 * a.q()[j];
    if (qp_prox(st, I.s)) dfv = dfv + mginv * a.x0()[j];
    a.df()[j] = dfv;
    a.dphi()[j] = dfv + 1 * a.Atyh()[j];
  }
  __syncthreads();
}

/* single-QP boundary operations (include/solver_interface.h) on device-resident state */
template <int RPT>
__global__ __launch_bounds__(QP_T) QP_OCCUPANCY void k_op(qpg_view V, int b, int op) {
  __shared__ IterShared I;
  char *lds = QP_DYN_LDS();
  const qpg_settings &st = *V.settings;
  const QpArrays a = qp_arrays(V, b);
  const int n = a.n, m = a.m, tid = threadIdx.x, slot = b;
  double *L = V.L + (size_t)slot * V.ld * V.nfac, *Dg = V.Dg + (size_t)slot * V.nfac, *Wst = V.Wst + (size_t)slot * V.wst_stride;
  if (tid == 0) { I.s = V.sc[b]; I.S.panel_wave = 0; I.S.placement = 0; for (int w = 0; w < QP_NW; w++) I.S.wave_rank[w] = w; }
  __syncthreads();
  switch (op) {
    case QP_OP_MATVEC_A: spmv_rows<8>(m, a.Atp(), a.Ati(), a.Atx(), V.op_in, [&](int r, double s) { V.op_out[r] = s; }); break;
    case QP_OP_MATVEC_Q: spmv_rows<8>(n, a.Qfp(), a.Qfi(), a.Qfx(), V.op_in, [&](int r, double s) { V.op_out[r] = s; }); break;
    case QP_OP_MATTVEC_A: spmv_rows<16>(n, a.Ap(), a.Ai(), a.Ax(), V.op_in, [&](int r, double s) { V.op_out[r] = s; }); break;
    case QP_OP_LDLCHOL:
      form_schur(V, b, n, L, false, false, qp_prox(st, I.s) != 0, I.s.gamma, I.S, lds);
      dev_factor<RPT>(V, n, L, Dg, lds, I.s.ticks_dbg);
      break;
    case QP_OP_FACTOR_LOADED: /* the host wrote a symmetric matrix (lower triangle) into the slot */
      if (qp_prox(st, I.s)) for (int j = tid; j < n; j += QP_T) L[(size_t)j * V.ld + j] += 1.0 / I.s.gamma;
      __syncthreads();
      dev_factor<RPT>(V, n, L, Dg, lds, I.s.ticks_dbg);
      break;
    case QP_OP_LDLCHOL_QATSA:
      form_schur(V, b, n, L, false, true, qp_prox(st, I.s) != 0, I.s.gamma, I.S, lds);
      dev_factor<RPT>(V, n, L, Dg, lds, I.s.ticks_dbg);
      break;
    case QP_OP_UPDATE_ENTER: dev_updown<RPT>(V, b, n, L, Dg, Wst, a.enter(), I.s.nb_enter, a.leave(), 0, I.S, lds, I.s.ticks_dbg); break;
    case QP_OP_DOWNDATE_LEAVE: dev_updown<RPT>(V, b, n, L, Dg, Wst, a.enter(), 0, a.leave(), I.s.nb_leave, I.S, lds, I.s.ticks_dbg); break;
    case QP_OP_UPDATE_SIGMA: { /* solver_interface.c:443-503; At_scale and the changed list (enter) were set by the caller */
      const int nchg = I.s.nb_sigma_changed;
      dev_ldlupdate_sigma_scale(a, nchg);
      dev_updown<RPT>(V, b, n, L, Dg, Wst, a.enter(), nchg, a.leave(), 0, I.S, lds, I.s.ticks_dbg);
      dev_update_sigma_post(a, I, nchg);
      break;
    }
    case QP_OP_SOLVE:
      for (int j = tid; j < n; j += QP_T) a.d()[j] = a.dphi()[j] * -1;
      __syncthreads();
      dense_solve(L, Dg, n, V.ld, a.d(), lds, V.lds_bytes);
      break;
    case QP_OP_RESIDUALS: dev_compute_residuals(V, a, I); break;
    case QP_OP_ACTIVE: dev_active_sets(a, I); break;
    case QP_OP_LINESEARCH: { const double tau = dev_linesearch(V, a, I, lds); if (tid == 0) I.s.tau = tau; break; }
    /* the KKT operations of solver_interface.h:82-126 on the (n+m) x (n+m) panel (batches created with FACTORIZE_KKT) */
    case QP_OP_KKT_FORM: if (V.kkt) kkt_newton<RPT>(&V, b, L, Dg, Wst, &I, lds, 3, 0, 0, 0); break;
    case QP_OP_KKT_FACTOR: if (V.kkt) kkt_newton<RPT>(&V, b, L, Dg, Wst, &I, lds, 4, 0, 0, 0); break;
    case QP_OP_KKT_ENTER: if (V.kkt) kkt_newton<RPT>(&V, b, L, Dg, Wst, &I, lds, 2, I.s.nb_enter, 0, 0); break;
    case QP_OP_KKT_LEAVE: if (V.kkt) kkt_newton<RPT>(&V, b, L, Dg, Wst, &I, lds, 2, 0, I.s.nb_leave, 0); break;
    case QP_OP_KKT_SOLVE: if (V.kkt) kkt_newton<RPT>(&V, b, L, Dg, Wst, &I, lds, 0, 0, 0, QP_KKT_SOLVE); break;
    case QP_OP_KKT_EXPAND: if (V.kkt) kkt_newton<RPT>(&V, b, L, Dg, Wst, &I, lds, 5, 0, 0, 0); break;
    default: break;
  }
  __syncthreads();
  if (tid == 0) V.sc[b] = I.s;
}

/* ... the same operations on a batch that keeps the SPARSE factor (round 6; through round 5 they were refused): ldlcholQAtsigmaA / ldlchol of Q (+ I / gamma),
 * the updates for entering, leaving and sigma-changed rows as path updates (always: the single operation has no "refactorise instead" policy), the solve.
 * One instance (the sparse factor has no rows-per-thread variants). */
__global__ __launch_bounds__(QP_T) QP_OCCUPANCY void k_op_sparse(qpg_view V, int b, int op) {
  __shared__ IterShared I;
  const qpg_settings &st = *V.settings;
  const QpArrays a = qp_arrays(V, b);
  const int n = a.n, tid = threadIdx.x, slot = b;
  if (tid == 0) { I.s = V.sc[b]; I.S.panel_wave = 0; I.S.placement = 0; for (int w = 0; w < QP_NW; w++) I.S.wave_rank[w] = w; }
  __syncthreads();
  const SpArrays SP = sp_arrays(V, b, slot, V.Dg + (size_t)slot * V.nfac, QP_DYN_LDS());
  switch (op) {
    case QP_OP_LDLCHOL: sp_factor(V, b, n, SP, false, qp_prox(st, I.s) != 0, I.s.gamma); break;
    case QP_OP_LDLCHOL_QATSA: sp_factor(V, b, n, SP, true, qp_prox(st, I.s) != 0, I.s.gamma); break;
    case QP_OP_UPDATE_ENTER: sp_updown(V, b, n, SP, a.enter(), I.s.nb_enter, a.leave(), 0); break;
    case QP_OP_DOWNDATE_LEAVE: sp_updown(V, b, n, SP, a.enter(), 0, a.leave(), I.s.nb_leave); break;
    case QP_OP_UPDATE_SIGMA: {
      const int nchg = I.s.nb_sigma_changed;
      dev_ldlupdate_sigma_scale(a, nchg);
      sp_updown(V, b, n, SP, a.enter(), nchg, a.leave(), 0);
      dev_update_sigma_post(a, I, nchg);
      break;
    }
    case QP_OP_SOLVE:
      for (int j = tid; j < n; j += QP_T) a.d()[j] = a.dphi()[j] * -1;
      __syncthreads();
      sp_solve(n, SP, a.d());
      break;
    default: break;
  }
  __syncthreads();
  if (tid == 0) V.sc[b] = I.s;
}

/* y = Op x for a caller-owned operator given by its compressed rows (boundary mat_vec) */
__global__ __launch_bounds__(QP_T) void k_spmv_generic(int nrows, const int *ptr, const int *idx, const double *val, const double *x, double *y) {
  spmv_rows<8>(nrows, ptr, idx, val, x, [&](int r, double s) { y[r] = s; });
}

/* Every QP of the batch: d = -(L D L')^{-1} dphi with its current factor, `reps` times.  This is
 * the stand-alone LDL^T-solve kernel the roofline line of bench.py measures (8.03 MB of
 * algorithmic traffic per QP and repetition at n = 1000). */
__global__ __launch_bounds__(QP_T) QP_OCCUPANCY void k_ldlsolve_all(qpg_view V, int reps) {
  char *lds = QP_DYN_LDS();
  for (int b = blockIdx.x; b < V.B && b < V.nslots; b += gridDim.x) {
    const QpArrays a = qp_arrays(V, b);
    const int n = a.n;
    double *L = V.L + (size_t)b * V.ld * V.nfac, *Dg = V.Dg + (size_t)b * V.nfac;
    for (int r = 0; r < reps; r++) {
      for (int j = threadIdx.x; j < n; j += QP_T) a.d()[j] = a.dphi()[j] * -1;
      __syncthreads();
      dense_solve(L, Dg, n, V.ld, a.d(), lds, V.lds_bytes);
    }
  }
}

/* ---- coop mode: the linear algebra of ONE QP spread over a grid of workgroups, one block column per launch (the kernel boundary is
 * the synchronisation; chained by coop_solve in qpalm_capi.inc while the QP's iteration is suspended at its linear-algebra site).
 * which = 0: the factor L / Dfac of the Newton system, 1: LD_Q of the dual objective.  Pieces: qpalm_dense.h (co_*). ---- */
QPD double *co_slot_L(const qpg_view &V, int slot, int which) { return (which ? V.LQ : V.L) + (size_t)slot * V.ld * V.nfac; }
QPD double *co_slot_D(const qpg_view &V, int slot, int which) { return (which ? V.DgQ : V.Dg) + (size_t)slot * V.nfac; }
__global__ __launch_bounds__(QP_T) QP_OCCUPANCY void k_co_form(qpg_view V, int b, int slot) {
  __shared__ QpShared S;
  char *lds = QP_DYN_LDS();
  const QpArrays a = qp_arrays(V, b);
  const qpg_scalars &sc = V.sc[b];
  const qpg_settings &st = *V.settings;
  const int la = sc.pend_la, prox = (st.proximal != 0 || sc.nc_flag != 0) ? 1 : 0;
  form_schur(V, b, a.n, co_slot_L(V, slot, la == 7), false, la == 1, (la == 1 || la == 3) && prox, sc.pend_gam, S, lds, (int)blockIdx.x, (int)gridDim.x);
}
/* the first diagonal block (J = 0) */
__global__ __launch_bounds__(QP_T) QP_OCCUPANCY void k_co_factor_diag(qpg_view V, int b, int slot, int which, int J) {
  char *lds = QP_DYN_LDS();
  co_factor_diag(co_slot_L(V, slot, which), co_slot_D(V, slot, which), qp_arrays(V, b).n, V.ld, lds, J);
}
/* trailing update with the 32 columns of block J; workgroup 0 takes the item that holds the next diagonal block first and
 * factorises that block right after it, in the same launch (nobody else reads it before the next launch) */
__global__ __launch_bounds__(QP_T) QP_OCCUPANCY void k_co_factor_trailing(qpg_view V, int b, int slot, int which, int J) {
  char *lds = QP_DYN_LDS();
  const int n = qp_arrays(V, b).n;
  co_factor_trailing(co_slot_L(V, slot, which), co_slot_D(V, slot, which), n, V.ld, lds, J, (int)blockIdx.x, (int)gridDim.x);
  if (blockIdx.x == 0) co_factor_diag(co_slot_L(V, slot, which), co_slot_D(V, slot, which), n, V.ld, lds, J + QP_FNB);
}
__global__ __launch_bounds__(QP_T) QP_OCCUPANCY void k_co_factor_rows(qpg_view V, int b, int slot, int which, int J) {
  char *lds = QP_DYN_LDS();
  co_factor_rows(co_slot_L(V, slot, which), co_slot_D(V, slot, which), qp_arrays(V, b).n, V.ld, lds, J, (int)blockIdx.x, (int)gridDim.x);
}
/* ldlsolveLD_neg_dphi on d (the suspended iteration wrote -dphi there): phase 0 forward block J (solved blocks collect in temp_n),
 * 1 division by D back into d, 2 backward block J (solved blocks collect in temp_n), 3 d <- temp_n */
__global__ __launch_bounds__(QP_T) QP_OCCUPANCY void k_co_solve(qpg_view V, int b, int slot, int phase, int J) {
  char *lds = QP_DYN_LDS();
  const QpArrays a = qp_arrays(V, b);
  const int n = a.n;
  double *L = co_slot_L(V, slot, 0), *Dg = co_slot_D(V, slot, 0);
  if (phase == 0) co_solve_forward(L, n, V.ld, a.d(), a.temp_n(), lds, J, (int)blockIdx.x, (int)gridDim.x);
  else if (phase == 1) { for (int i = blockIdx.x * QP_T + threadIdx.x; i < n; i += QP_T * gridDim.x) a.d()[i] = a.temp_n()[i] / Dg[i]; }
  else if (phase == 2) co_solve_backward(L, n, V.ld, a.d(), a.temp_n(), lds, J, (int)blockIdx.x, (int)gridDim.x);
  else { for (int i = blockIdx.x * QP_T + threadIdx.x; i < n; i += QP_T * gridDim.x) a.d()[i] = a.temp_n()[i]; }
}

/* ldlupdate_entering_constraints / ldldowndate_leaving_constraints / ldlupdate_sigma_changed (solver_interface.c:407-503) of a
 * suspended iteration, sweep r0 / 16 over ranks r0 .. r0 + 15 of the list (entering rows first, then leaving rows; the counts are
 * read from the QP's scalars, so that the launch chain of a sweep is the same every time and can be replayed as a graph):
 * phase 0 clears the running vectors (all workgroups), 1 scatters the rows (one workgroup), 2 = block column J (J >= n: write
 * back the last stage) */
__global__ __launch_bounds__(QP_T) QP_OCCUPANCY void k_co_updown(qpg_view V, int b, int slot, int phase, int J, int r0) {
  __shared__ QpShared S;
  char *lds = QP_DYN_LDS();
  const QpArrays a = qp_arrays(V, b);
  const int n = a.n;
  const int la = V.sc[b].pend_la, n_up = (la == 2) ? V.sc[b].nb_enter : V.sc[b].pend_nchange, n_dn = (la == 2) ? V.sc[b].nb_leave : 0;
  if (r0 >= n_up + n_dn) return;
  const int kk = (n_up + n_dn - r0 < QPG_KMAX) ? (n_up + n_dn - r0) : QPG_KMAX;
  double *L = co_slot_L(V, slot, 0), *Dg = co_slot_D(V, slot, 0), *Wst = V.Wst + (size_t)slot * V.wst_stride;
  double *hst = Wst + (size_t)QPG_KMAX * V.nfac + QPG_DUMMY;
  if (phase < 2) {
    co_updown_init<QPG_KMAX>(V.Atp + (size_t)b * (V.m + 1), V.Ati + (size_t)b * V.nnzA, V.Atss + (size_t)b * V.nnzA, n, Wst, hst, a.enter(), n_up,
                             a.leave(), r0, kk, phase, S, (int)blockIdx.x, (int)gridDim.x);
    if (phase == 1 && blockIdx.x == 0 && threadIdx.x == 0) { /* the sweep's work counters, as dense_updown keeps them (QPGStats.n_sweeps, sweep_entries) */
      const int J0 = ((int)hst[CO_UD_JMIN] / QP_UNB) * QP_UNB;
      V.sc[b].ticks_dbg[QPG_CNT_SWEEPS] += 1;
      V.sc[b].ticks_dbg[QPG_CNT_SWEEP_ENTRIES] += (long long)(n - J0) * (n - J0 - 1) / 2 + (n - J0);
    }
  }
  else co_updown_block<QPG_KMAX>(n, V.ld, L, Dg, Wst, hst, J, r0, kk, n_up, lds, (int)blockIdx.x, (int)gridDim.x, qp_pivot_mode(V, b));
  if (phase == 1 && blockIdx.x == 0 && threadIdx.x == 0 && V.co_flags) V.co_flags[(size_t)b * 4] = (int)hst[CO_UD_JMIN] / QP_UNB; /* the persistent sweep's counter: blocks published so far */
}
/* ... and the whole sweep in ONE launch after phases 0 and 1 (co_updown_persist: one workgroup per 128 rows, at most 64 of them on the
 * chip, so no occupancy attribute: the row's 32 entries and 16 running values stay in registers next to the recurrence's) */
__global__ __launch_bounds__(QP_T) void k_co_sweep(qpg_view V, int b, int slot, int r0) {
  __shared__ QpShared S;
  char *lds = QP_DYN_LDS();
  const QpArrays a = qp_arrays(V, b);
  const int n = a.n;
  const int la = V.sc[b].pend_la, n_up = (la == 2) ? V.sc[b].nb_enter : V.sc[b].pend_nchange, n_dn = (la == 2) ? V.sc[b].nb_leave : 0;
  if (r0 >= n_up + n_dn) return;
  const int kk = (n_up + n_dn - r0 < QPG_KMAX) ? (n_up + n_dn - r0) : QPG_KMAX;
  double *L = co_slot_L(V, slot, 0), *Dg = co_slot_D(V, slot, 0), *Wst = V.Wst + (size_t)slot * V.wst_stride;
  const double *hst = Wst + (size_t)QPG_KMAX * V.nfac + QPG_DUMMY;
  co_updown_persist<QPG_KMAX>(n, V.ld, L, Dg, Wst, hst, V.co_tab + (size_t)b * V.co_tab_stride, V.co_flags + (size_t)b * 4, r0, kk, n_up, lds, (int)blockIdx.x, S, qp_pivot_mode(V, b));
}

/* Diagnostic (tools/evidence/sweep_probe.py): every resident workgroup factorises Q + I/gamma of its QP and then applies `reps` times a
 * rank-`nranks` update followed by the downdate with the same rows of A (constraints 0 .. nranks-1), so that variants of the
 * update sweep can be timed under the contention of a full chip without running the solver around them.  The phase timers of
 * the sweeps (QPGStats.ms_dbg) are left in the QP's scalars.  Not part of the solver path. */
template <int RPT>
__global__ __launch_bounds__(QP_T) QP_OCCUPANCY void k_sweep_probe(qpg_view V, int reps, int nranks) {
  __shared__ IterShared I;
  char *lds = QP_DYN_LDS();
  qp_place_panel_wave(V, I.S);
  const qpg_settings &st = *V.settings;
  for (int b = blockIdx.x; b < V.B && b < V.nslots; b += gridDim.x) {
    const QpArrays a = qp_arrays(V, b);
    const int n = a.n, tid = threadIdx.x, slot = blockIdx.x;
    double *L = V.L + (size_t)slot * V.ld * V.nfac, *Dg = V.Dg + (size_t)slot * V.nfac, *Wst = V.Wst + (size_t)slot * V.wst_stride;
    __syncthreads();
    if (tid == 0) { I.s = V.sc[b]; for (int k = 0; k < QPG_NDBG; k++) I.s.ticks_dbg[k] = 0; }
    __syncthreads();
    form_schur(V, b, n, L, false, false, true, qp_gamma_init(st, I.s), I.S, lds);
    dev_factor<RPT>(V, n, L, Dg, lds, I.s.ticks_dbg);
    for (int i = tid; i < nranks && i < a.m; i += QP_T) { a.enter()[i] = i; a.leave()[i] = i; }
    __syncthreads();
    const int nr = (nranks < a.m) ? nranks : a.m;
    const long long t0 = QP_CLOCK();
    for (int r = 0; r < reps; r++) {
      dev_updown<RPT>(V, b, n, L, Dg, Wst, a.enter(), nr, a.leave(), 0, I.S, lds, I.s.ticks_dbg);
      dev_updown<RPT>(V, b, n, L, Dg, Wst, a.enter(), 0, a.leave(), nr, I.S, lds, I.s.ticks_dbg);
    }
    __syncthreads();
    if (tid == 0) { I.s.ticks_update = QP_CLOCK() - t0; I.s.n_sweeps = (int)I.s.ticks_dbg[QPG_CNT_SWEEPS]; V.sc[b] = I.s; }
    __syncthreads();
  }
}

/* HBM yardsticks quoted by bench.py next to the 8 TB/s spec figure ("measure the attainable
 * ceiling on the box"): a copy (read + write; every workgroup streams contiguous 32 KB pieces, eight 16-byte loads in
 * flight per lane -- the best of the forms in tools/evidence/copybench.hip, ~5.4 TB/s) and a read-only stream (~6.5 TB/s). */
__global__ __launch_bounds__(256) void k_hbm_copy(const double *__restrict__ src, double *__restrict__ dst, size_t npairs) {
#ifdef QPALM_EMU
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npairs; i += (size_t)gridDim.x * blockDim.x) { dst[2 * i] = src[2 * i]; dst[2 * i + 1] = src[2 * i + 1]; }
#else
  typedef double copy_d2 __attribute__((ext_vector_type(2)));
  const copy_d2 *s2 = (const copy_d2 *)src;
  copy_d2 *d2 = (copy_d2 *)dst;
  constexpr int U = 8;
  const size_t per = (size_t)U * 256;
  for (size_t base = (size_t)blockIdx.x * per; base < npairs; base += (size_t)gridDim.x * per) {
    copy_d2 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) { const size_t i = base + u * 256 + threadIdx.x; v[u] = (i < npairs) ? s2[i] : copy_d2{0, 0}; }
#pragma unroll
    for (int u = 0; u < U; u++) { const size_t i = base + u * 256 + threadIdx.x; if (i < npairs) d2[i] = v[u]; }
  }
#endif
}
__global__ __launch_bounds__(256) void k_hbm_read(const double *__restrict__ src, double *__restrict__ out, size_t npairs) {
  double acc = 0.0;
#ifdef QPALM_EMU
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npairs; i += (size_t)gridDim.x * blockDim.x) acc += src[2 * i] + src[2 * i + 1];
#else
  typedef double copy_d2 __attribute__((ext_vector_type(2)));
  const copy_d2 *s2 = (const copy_d2 *)src;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npairs; i += (size_t)gridDim.x * blockDim.x) { const copy_d2 v = s2[i]; acc += v.x + v.y; }
#endif
  if (acc == 1.234567) out[0] = acc; /* keeps the loads alive */
}

#endif
