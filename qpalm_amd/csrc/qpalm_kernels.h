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

/* The persistent solver: workgroup `blockIdx.x` owns factor slot `blockIdx.x` and pulls QPs either
 * statically (b = blockIdx.x, resumable, needs B <= grid) or from an atomic work queue. */
template <int RPT>
__global__ __launch_bounds__(QP_T) QP_OCCUPANCY void k_solve(qpg_view V, int budget, int dynamic) {
  __shared__ IterShared I;
  char *lds = QP_DYN_LDS();
  /* one call site of dev_solve (= one copy of the iteration loop in the kernel): static round-robin
   * or the atomic work queue only differ in how the next QP index is obtained */
  int b = blockIdx.x - gridDim.x;
  while (true) {
    if (dynamic) {
      __syncthreads();
      if (threadIdx.x == 0) I.S.ibc[0] = atomicAdd(V.queue, 1);
      __syncthreads();
      b = QP_UNIFORM(I.S.ibc[0]);
    } else b += gridDim.x;
    if (b >= V.B) break;
    dev_solve<RPT>(V, b, blockIdx.x, budget, I, lds);
  }
}

/* qpalm_update_bounds, device part (qpalm.c:819-826): the host wrote the raw bounds */
__global__ __launch_bounds__(QP_T) void k_update_bounds(qpg_view V, int has_bmin, int has_bmax, const int *skip) {
  for (int b = blockIdx.x; b < V.B; b += gridDim.x) {
    if (skip && skip[b]) continue;
    const QpArrays a = qp_arrays(V, b);
    if (V.sc[b].has_scaling)
      for (int i = threadIdx.x; i < a.m; i += QP_T) {
        if (has_bmin) a.bmin()[i] = a.E()[i] * a.bmin()[i];
        if (has_bmax) a.bmax()[i] = a.E()[i] * a.bmax()[i];
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
        if (st.proximal) Qxj = Qxj + mg * a.x()[j];
        a.Qxv()[j] = Qxj;
        const double t = qj + cinv_old * Qxj;
        vm[0] = qmax(vm[0], qabs(t));
      }
      block_reduce<1, 0>(I.S, vm, vs);
      const double c = 1 / qmax(1.0, vm[0]);
      const double ratio = c / c_old;
      const double ginit = st.gamma_init;
      for (int j = threadIdx.x; j < a.n; j += QP_T) {
        a.q()[j] *= c;
        double Qxj = a.Qxv()[j] * ratio;
        if (st.proximal) Qxj = Qxj + (1 / ginit) * a.x()[j];
        a.Qxv()[j] = Qxj;
      }
      const int nzQ = a.Qp()[a.n], nzQf = a.Qfp()[a.n];
      for (int k = threadIdx.x; k < nzQ; k += QP_T) a.Qx()[k] *= ratio;
      __syncthreads();
      for (int k = threadIdx.x; k < nzQf; k += QP_T) a.Qfx()[k] = a.Qx()[a.Qfperm()[k]];
      if (threadIdx.x == 0) {
        I.s.sc_c = c; I.s.sc_cinv = 1 / c;
        if (st.proximal) I.s.gamma = ginit;
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
    if (st.proximal) dfv = dfv + mginv * a.x0()[j];
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
  if (tid == 0) I.s = V.sc[b];
  __syncthreads();
  switch (op) {
    case QP_OP_MATVEC_A: spmv_rows<8>(m, a.Atp(), a.Ati(), a.Atx(), V.op_in, [&](int r, double s) { V.op_out[r] = s; }); break;
    case QP_OP_MATVEC_Q: spmv_rows<8>(n, a.Qfp(), a.Qfi(), a.Qfx(), V.op_in, [&](int r, double s) { V.op_out[r] = s; }); break;
    case QP_OP_MATTVEC_A: spmv_rows<16>(n, a.Ap(), a.Ai(), a.Ax(), V.op_in, [&](int r, double s) { V.op_out[r] = s; }); break;
    case QP_OP_LDLCHOL:
      form_schur(V, b, n, L, false, false, st.proximal != 0, I.s.gamma, I.S, lds);
      dev_factor<RPT>(V, n, L, Dg, lds, I.s.ticks_dbg);
      break;
    case QP_OP_FACTOR_LOADED: /* the host wrote a symmetric matrix (lower triangle) into the slot */
      if (st.proximal) for (int j = tid; j < n; j += QP_T) L[(size_t)j * V.ld + j] += 1.0 / I.s.gamma;
      __syncthreads();
      dev_factor<RPT>(V, n, L, Dg, lds, I.s.ticks_dbg);
      break;
    case QP_OP_LDLCHOL_QATSA:
      form_schur(V, b, n, L, false, true, st.proximal != 0, I.s.gamma, I.S, lds);
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

/* Plain HBM copy (16 bytes per lane and step, grid-stride): the attainable-bandwidth yardstick that bench.py quotes
 * next to the 8 TB/s spec figure ("measure the attainable ceiling on the box"). */
__global__ __launch_bounds__(256) void k_hbm_copy(const double *__restrict__ src, double *__restrict__ dst, size_t npairs) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
#ifdef QPALM_EMU
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npairs; i += stride) { dst[2 * i] = src[2 * i]; dst[2 * i + 1] = src[2 * i + 1]; }
#else
  typedef double copy_d2 __attribute__((ext_vector_type(2)));
  const copy_d2 *s2 = (const copy_d2 *)src;
  copy_d2 *d2 = (copy_d2 *)dst;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < npairs; i += 4 * stride) { /* four independent 16-byte loads in flight per lane */
    const copy_d2 a = __builtin_nontemporal_load(s2 + i), b = __builtin_nontemporal_load(s2 + i + stride);
    const copy_d2 c = __builtin_nontemporal_load(s2 + i + 2 * stride), d = __builtin_nontemporal_load(s2 + i + 3 * stride);
    __builtin_nontemporal_store(a, d2 + i); __builtin_nontemporal_store(b, d2 + i + stride);
    __builtin_nontemporal_store(c, d2 + i + 2 * stride); __builtin_nontemporal_store(d, d2 + i + 3 * stride);
  }
  for (; i < npairs; i += stride) d2[i] = s2[i];
#endif
}

#endif
