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
#include <type_traits>

#ifdef QPALM_EMU
/* cross-workgroup hand-off (co_updown_persist): the emulator runs the workgroups of a grid one after the other, in order */
#define QP_FLAG_LOAD(p) (*(volatile int *)(p))
#define QP_FLAG_STORE(p, v) (*(volatile int *)(p) = (v))
#define QP_RELEASE_AGENT() do { } while (0)
#define QP_ACQUIRE_AGENT() do { } while (0)
#define QP_DRAIN_VMEM() do { } while (0)
#define QP_SLEEP() do { } while (0)
#define QP_WAVE_SYNC() emu_wave_sync()
#define QP_LDS_VBASE(p) (p)
#define QP_SCHED_BARRIER() do { } while (0)
#define QP_SETPRIO(p) do { } while (0)
#define QP_DRAIN_LDS() do { } while (0)
#else
/* cross-workgroup hand-off inside one launch (co_updown_persist): payload by plain stores, every storing wavefront drains, ONE lane
 * releases at agent scope (write-back of the XCD's L2), drains again (the compiler may drop the fence's own wait) and stores the flag
 * with an agent-scope atomic; the consumer polls that ONE word relaxed, acquires ONCE at agent scope (drops its CU's stale L1 lines),
 * drains, and the workgroup barrier behind it lets the other wavefronts load plainly. */
#define QP_FLAG_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define QP_FLAG_STORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define QP_RELEASE_AGENT() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent")
#define QP_ACQUIRE_AGENT() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent")
#define QP_DRAIN_VMEM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define QP_SLEEP() __builtin_amdgcn_s_sleep(2)
#define QP_SETPRIO(p) __builtin_amdgcn_s_setprio(p) /* issue priority of a wavefront on its SIMD */
#define QP_DRAIN_LDS() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory") /* diagnostic stamps: every LDS operation issued so far has returned */
/* keeps the scheduler from hoisting a whole unrolled loop's LDS reads to the top (register blow-up) */
#define QP_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
#define QP_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
/* an LDS address the compiler must keep in ONE VGPR (it otherwise hoists every "base + constant" of a loop into its own SGPR
 * and pays a v_mov per access to get it back into a VGPR): accesses become ds_* vbase offset:imm */
template <class T> static __device__ __forceinline__ T QP_LDS_AS *qp_lds_vbase_(T QP_LDS_AS *p) {
  unsigned a = (unsigned)(size_t)p;
  asm volatile("" : "+v"(a));
  return (T QP_LDS_AS *)(size_t)a;
}
#define QP_LDS_VBASE(p) qp_lds_vbase_(p)
#endif


#ifdef QPALM_EMU
QPD double qp_readlane(double v, int src) { return emu_exchange(v, src); }
QPD int qp_readlane_i(int v, int src) { return emu_exchange(v, src); }
#else
QPD int qp_readlane_i(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
QPD double qp_readlane(double v, int src) { /* src is wave-uniform: v_readlane_b32 x2, result lives in SGPRs */
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
#endif

/* 1/x to (almost) full fp64 precision: v_rcp_f64 + two Newton steps; avoids the ~28-instruction
 * IEEE division sequence on the serial chain of the update recurrence */
#ifndef QP_PANEL_TIMING
#define QP_PANEL_TIMING 0 /* 1 (diagnostic build): ms_dbg[8..11] = panel wave: rows of the block / recurrence / solve + write-back / barrier wait */
#endif
#ifdef QPALM_EMU
QPD double qp_rcp(double x) { return 1.0 / x; }
#else
QPD double qp_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = QP_FMA(QP_FMA(-x, r, 1.0), r, r);
  r = QP_FMA(QP_FMA(-x, r, 1.0), r, r);
  return r;
}
#endif

/* ---------------------------------------------------------------------------------------------
 * form_schur_narrow: the same assembly for SMALL QPs whose rows of A are short (MPC: ~7 entries): a quarter wavefront
 * (16 lanes, one DPP row) per column, so that a workgroup assembles 4 QP_NW columns per round instead of QP_NW.  At this
 * size the walk is a chain of ~7 dependent global round trips per column and nothing else, so the time goes with the
 * number of rounds (measured on mpc-160, n = 160: 0.25 ms per assembly with one wavefront per column, 40 rounds).
 * Same order of additions per entry as form_schur (active rows t ascending).
 * ------------------------------------------------------------------------------------------- */
QPN void form_schur_narrow(const qpg_view &V, int b, const int n, double *Lslot, bool with_AtSA, bool proximal, double gamma, char *lds) {
  const int ld = V.ld;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, grp = lane >> 4, gl = lane & 15;
  const int *Ap = V.Ap + (size_t)b * (V.n + 1), *Ai = V.Ai + (size_t)b * V.nnzA;
  const int *Atp = V.Atp + (size_t)b * (V.m + 1), *Ati = V.Ati + (size_t)b * V.nnzA;
  const double *Atss = V.Atss + (size_t)b * V.nnzA;
  const int *Ainv = V.Ainv + (size_t)b * V.nnzA;
  const int *Qp = V.Qp + (size_t)b * (V.n + 1), *Qi = V.Qi + (size_t)b * V.nnzQ;
  const double *Qx = V.Qx + (size_t)b * V.nnzQ;
  const int *active = V.active + (size_t)b * V.m;
  double *buf = (double *)lds + (size_t)(wid * 4 + grp) * n; /* one column buffer per quarter wavefront (host: 32 QP_NW n <= lds_bytes) */
  __syncthreads();
  for (int j0 = 0; j0 < n; j0 += 4 * QP_NW) {
    const int j = j0 + wid * 4 + grp;
    const bool colok = (j < n);
    const int lo = colok ? j : 0;
    if (colok) for (int i = j + gl; i < n; i += 16) buf[i] = 0.0;
    QP_WAVE_SYNC();
    if (with_AtSA) {
      const int p0 = colok ? Ap[j] : 0, p1 = colok ? Ap[j + 1] : 0;
      /* chunks of 16 entries of A(:,j): lane = entry; the wavefront runs as long as its longest column */
      int more = 1;
      for (int pb = p0; more; pb += 16) {
        const int pidx = pb + gl;
        const bool valid = pidx < p1;
        const int t_l = valid ? Ai[pidx] : 0;
        const int act_l = valid ? active[t_l] : 0;
        int k0_l = 0, cnt_l = 0;
        double vj_l = 0.0;
        if (act_l) { k0_l = Atp[t_l]; cnt_l = Atp[t_l + 1] - k0_l; vj_l = Atss[Ainv[pidx]]; }
        unsigned gmask = (unsigned)((__ballot(act_l) >> (16 * grp)) & 0xffffull); /* this column's active rows, ascending */
        while (__ballot(gmask != 0)) {
          /* two active rows per step (their entries of F are fetched together), added in ascending order */
          int sl[2], cn[2], kk0[2], ig[2];
          double vg[2], vjg[2];
#pragma unroll
          for (int g = 0; g < 2; g++) {
            const bool has = gmask != 0;
            sl[g] = has ? (__ffs((int)gmask) - 1) : 0;
            if (has) gmask &= gmask - 1;
            kk0[g] = __shfl(k0_l, 16 * grp + sl[g]); cn[g] = has ? __shfl(cnt_l, 16 * grp + sl[g]) : 0; vjg[g] = __shfl(vj_l, 16 * grp + sl[g]);
            if (!has) cn[g] = 0;
            ig[g] = 0; vg[g] = 0.0;
            if (gl < cn[g]) { ig[g] = Ati[kk0[g] + gl]; vg[g] = Atss[kk0[g] + gl]; }
          }
#pragma unroll
          for (int g = 0; g < 2; g++) {
            if (gl < cn[g] && ig[g] >= lo) buf[ig[g]] += vg[g] * vjg[g];
            for (int kb = 16; __ballot(kb < cn[g]); kb += 16) { /* rows of A with more than 16 entries */
              const int k = kb + gl;
              if (k < cn[g]) { const int i = Ati[kk0[g] + k]; if (i >= lo) buf[i] += Atss[kk0[g] + k] * vjg[g]; }
            }
            QP_WAVE_SYNC();
          }
        }
        more = __ballot(pb + 16 < p1) != 0;
      }
    }
    if (colok) {
      for (int k = Qp[j] + gl; k < Qp[j + 1]; k += 16) {
        const int i = Qi[k];
        if (i >= j) buf[i] = Qx[k] + buf[i];
      }
    }
    QP_WAVE_SYNC();
    if (colok && proximal && gl == 0) buf[j] += 1.0 / gamma;
    QP_WAVE_SYNC();
    if (colok) for (int i = j + gl; i < n; i += 16) Lslot[(size_t)j * ld + i] = buf[i];
    QP_WAVE_SYNC();
  }
  __syncthreads();
}

/* ---------------------------------------------------------------------------------------------
 * form_schur: H(:,j) for j = 0..n-1, lower triangle, written into the factor slot.
 *   H_ij = Q_ij + sum_{t active} F_it F_jt (+ 1/gamma on the diagonal), F = At_sqrt_sigma.
 * One wavefront assembles one column in an LDS column buffer: the active rows t of A(:,j) are
 * walked in ascending order (the order cholmod_aat accumulates in), lanes spread over the entries
 * of F(:,t) => conflict-free LDS adds, deterministic sums.
 * GERSH = true: no Q, full columns, returns max_j (C_jj + sum_{i!=j} |C_ij|)  (nonconvex.c:185-210).
 * (GERSH is a run-time flag so that the kernel holds ONE copy of this loop nest.)
 * ------------------------------------------------------------------------------------------- */
QPN double form_schur(const qpg_view &V, int b, const int n, double *Lslot, const bool GERSH, bool with_AtSA, bool proximal, double gamma,
                      QpShared &S, char *lds, const int wg = 0, const int nwg = 1) { /* (wg, nwg): this workgroup's share of the columns (k_co_form) */
  if (!GERSH && V.narrow_rows && nwg == 1) { form_schur_narrow(V, b, n, Lslot, with_AtSA, proximal, gamma, lds); return 0.0; }
  const int ld = V.ld; /* n = this QP's dimension; the per-QP strides below are the batch's V.n / V.m */
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int *Ap = V.Ap + (size_t)b * (V.n + 1), *Ai = V.Ai + (size_t)b * V.nnzA;
  const int *Atp = V.Atp + (size_t)b * (V.m + 1), *Ati = V.Ati + (size_t)b * V.nnzA;
  const double *Atss = V.Atss + (size_t)b * V.nnzA;
  const int *Ainv = V.Ainv + (size_t)b * V.nnzA;
  const int *Qp = V.Qp + (size_t)b * (V.n + 1), *Qi = V.Qi + (size_t)b * V.nnzQ;
  const double *Qx = V.Qx + (size_t)b * V.nnzQ;
  const int *active = V.active + (size_t)b * V.m;
  int ncb = V.lds_bytes / (8 * n);
  if (ncb > QP_NW) ncb = QP_NW;
  if (ncb < 1) ncb = 1; /* host guarantees lds_bytes >= 8n */
  double *buf = (double *)lds + (size_t)wid * n;
  double gmax = -1e300;
  __syncthreads();
  for (int j0 = wg * ncb; j0 < n; j0 += ncb * nwg) {
    const int j = j0 + wid;
    if (wid < ncb && j < n) {
      const int lo = GERSH ? 0 : j;
      for (int i = lo + lane; i < n; i += 64) buf[i] = 0.0;
      QP_WAVE_SYNC();
      if (with_AtSA) {
        /* Global latency (~0.5 us per dependent load) dominates this walk, so the metadata of up to
         * 64 rows t of A(:,j) is fetched in ONE round trip (lane = entry of the column) and the
         * entries of F(:,t) are fetched for four active rows at a time; the adds into the column
         * buffer stay sequential in t (ascending), which keeps the sums deterministic. */
        const int p0 = Ap[j], p1 = Ap[j + 1];
        for (int pb = p0; pb < p1; pb += 64) {
          const int pidx = pb + lane;
          const bool valid = pidx < p1;
          const int t_l = valid ? Ai[pidx] : 0;
          const int act_l = valid ? active[t_l] : 0;
          int k0_l = 0, cnt_l = 0;
          double vj_l = 0.0;
          if (act_l) { k0_l = Atp[t_l]; cnt_l = Atp[t_l + 1] - k0_l; vj_l = Atss[Ainv[pidx]]; }
          unsigned long long mask = __ballot(act_l);
          while (mask) {
            int sl[4], cn[4], kk0[4], ig[4];
            double vg[4], vjg[4];
            int ng = 0;
#pragma unroll
            for (int g = 0; g < 4; g++) {
              sl[g] = mask ? (__ffsll(mask) - 1) : 0;
              if (mask) { ng = g + 1; mask &= mask - 1; }
            }
#pragma unroll
            for (int g = 0; g < 4; g++) {
              kk0[g] = qp_readlane_i(k0_l, sl[g]); cn[g] = (g < ng) ? qp_readlane_i(cnt_l, sl[g]) : 0; vjg[g] = qp_readlane(vj_l, sl[g]);
              ig[g] = 0; vg[g] = 0.0;
              if (lane < cn[g]) { ig[g] = Ati[kk0[g] + lane]; vg[g] = Atss[kk0[g] + lane]; }
            }
#pragma unroll
            for (int g = 0; g < 4; g++) {
              if (g < ng) {
                if (lane < cn[g] && ig[g] >= lo) buf[ig[g]] += vg[g] * vjg[g];
                for (int kb = 64; kb < cn[g]; kb += 64) { /* rows of A with more than 64 entries */
                  const int k = kb + lane;
                  if (k < cn[g]) { const int i = Ati[kk0[g] + k]; if (i >= lo) buf[i] += Atss[kk0[g] + k] * vjg[g]; }
                }
                QP_WAVE_SYNC();
              }
            }
          }
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
/* R adjacent rows of one column: 16-byte accesses where R is even (the address is 16-byte aligned:
 * row index and leading dimension are even, slots are 256-byte aligned) */
#ifdef QPALM_EMU
struct qp_pair { double x, y; }; /* one (-w_j, -gamma) table entry, read with one 16-byte LDS access */
template <int R> QPD void qp_load_rows(const qp_gdouble *p, double *v) { for (int k = 0; k < R; k++) v[k] = p[k]; }
template <int R> QPD void qp_store_rows(qp_gdouble *p, const double *v) { for (int k = 0; k < R; k++) p[k] = v[k]; }
#else
typedef double qp_double2 __attribute__((ext_vector_type(2)));
typedef qp_double2 __attribute__((address_space(1))) qp_gdouble2;
typedef qp_double2 qp_pair;
template <int R> QPD void qp_load_rows(const qp_gdouble *p, double *v) {
  if (R % 2 == 0) {
#pragma unroll
    for (int k = 0; k < R; k += 2) { const qp_double2 t = *(const qp_gdouble2 *)(p + k); v[k] = t.x; v[k + 1] = t.y; }
  } else {
#pragma unroll
    for (int k = 0; k < R; k++) v[k] = p[k];
  }
}
template <int R> QPD void qp_store_rows(qp_gdouble *p, const double *v) {
  if (R % 2 == 0) {
#pragma unroll
    for (int k = 0; k < R; k += 2) { qp_double2 t; t.x = v[k]; t.y = v[k + 1]; *(qp_gdouble2 *)(p + k) = t; }
  } else {
#pragma unroll
    for (int k = 0; k < R; k++) p[k] = v[k];
  }
}
#endif
/* the update sweep's panel traffic with the non-temporal hint (an entry is read once and written once per sweep, the next sweep
 * is ~1 ms and 4 GB of other QPs' panels away): QP_NT_PANEL 1 = the sweep's stores, 2 = its loads and stores, 3 = also the panel reads
 * of the triangular solves, 4 = also those of the factorisation's panel update.  Measured in DESIGN section 7. */
#ifndef QP_NT_PANEL
#define QP_NT_PANEL 3
#endif
#ifdef QPALM_EMU
#define qp_load_rows_nt qp_load_rows
#define qp_store_rows_nt qp_store_rows
#define QP_LDNT(lvl, p) (*(p))
#else
#define QP_LDNT(lvl, p) ((QP_NT_PANEL >= (lvl)) ? __builtin_nontemporal_load(p) : *(p)) /* a streaming read of the panel (level = which phase, see QP_NT_PANEL) */
template <int R> QPD void qp_load_rows_nt(const qp_gdouble *p, double *v) {
  if (QP_NT_PANEL < 2) { qp_load_rows<R>(p, v); return; }
  if (R % 2 == 0) {
#pragma unroll
    for (int k = 0; k < R; k += 2) { const qp_double2 t = __builtin_nontemporal_load((const qp_gdouble2 *)(p + k)); v[k] = t.x; v[k + 1] = t.y; }
  } else {
#pragma unroll
    for (int k = 0; k < R; k++) v[k] = __builtin_nontemporal_load(p + k);
  }
}
template <int R> QPD void qp_store_rows_nt(qp_gdouble *p, const double *v) {
  if (QP_NT_PANEL < 1) { qp_store_rows<R>(p, v); return; }
  if (R % 2 == 0) {
#pragma unroll
    for (int k = 0; k < R; k += 2) { qp_double2 t; t.x = v[k]; t.y = v[k + 1]; __builtin_nontemporal_store(t, (qp_gdouble2 *)(p + k)); }
  } else {
#pragma unroll
    for (int k = 0; k < R; k++) __builtin_nontemporal_store(v[k], p + k);
  }
}
#endif
#define QP_FNB 32
#ifndef QP_FNT
#define QP_FNT 2 /* row tiles per wavefront and pass of the panel update: 2 keeps accumulators + two fragment stages under 128 VGPRs */
#endif
struct FactorLds {
  double Ld[QP_FNB][QP_FNB + 1];
  double dv[QP_FNB];
  double dg[QP_FNB];
  double colbuf[QP_FNB];
};

/* Panel update on the matrix cores:  P(rows, J : J + 16 NCT) -= L(rows, k0 : k1) D(k0 : k1) L(J : J + 16 NCT, k0 : k1)'
 * for the row tiles of one pass (tile index J/16 + tbase + ...), all eight wavefronts together.
 *
 *  - The block-row operand  -L(J + i, k) D(k)  (the same for every wavefront) is staged ONCE per chunk of QP_FKC columns
 *    in LDS by all threads (double buffered, one barrier per chunk; the global loads of chunk c+1 are in flight while
 *    the MFMAs of chunk c run) and read from there as MFMA A fragments: it is no longer re-read from HBM/L2 by every
 *    wavefront and pass.
 *  - NCT = 4 works on a 64-column super-block, so the panel L(rows, 0 : J) is streamed from HBM once per 64 columns
 *    instead of once per 32 (the re-read volume of the left-looking factorisation halves: n^3/(6*64) entries);
 *    NCT = 2 is the 32-column form used for the second half of a super-block (k range = the 32 columns before it).
 *  - NTJ = 2 works on PAIRS of adjacent panel rows: the wavefront's 32-row group is split into its even and its odd
 *    rows (two MFMA row tiles), so every panel fragment pair is ONE 16-byte load of a full 128-byte line.
 *  - Block row i of the MFMA A operand is column J + NCT*i + ct of the super-block (ct = column tile), so the NCT tile
 *    values of a lane are adjacent in LDS (16-byte reads).  Which MFMA lane handles which entry does not change any
 *    entry's sum: k ascending, one fma per k, as before. */
#ifndef QP_NI_FGEMM
#define QP_NI_FGEMM QPNI
#endif
#ifndef QP_FKC
#define QP_FKC 32 /* columns per staged chunk (J and the k ranges are multiples of it); 16 in the 256-thread instance (LDS budget) */
#endif
#define QP_FAS 66 /* row stride of the staged operand in doubles: 64 block rows + 2 (16-byte aligned, spreads the four k rows of a fragment over the banks) */
#ifndef QP_FST
#define QP_FST 4 /* panel fragments (k steps of 4 columns) in flight per wavefront; divides QP_FKC / 4 */
#endif
struct FactorStage { double As[2][QP_FKC][QP_FAS]; };

template <int NTJ, int NCT>
QP_NI_FGEMM void factor_panel_update(double *L_, const double *Dg_, char *stage_, int n_, int ld_, int J_, int tbase_, int k0_, int k1_) {
  const int n = QP_UNIFORM(n_), ld = QP_UNIFORM(ld_), J = QP_UNIFORM(J_), tbase = QP_UNIFORM(tbase_), k0 = QP_UNIFORM(k0_), k1 = QP_UNIFORM(k1_);
  qp_gdouble *L = (qp_gdouble *)L_;
  const qp_gdouble *Dg = (const qp_gdouble *)Dg_;
  FactorStage QP_LDS_AS &F = *QP_LDS_ARG(FactorStage, stage_);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  constexpr bool PAIR = (NTJ == 2);
  constexpr int NBW = 16 * NCT; /* block width */
  /* rows of this wavefront: NTJ == 2: 32 consecutive rows from rbase, tile t = rows rbase + 2 j + t; NTJ == 1: 16 rows */
  const int rbase = (J / 16 + tbase) * 16 + wid * 16 * NTJ;
  auto rowof = [&](const int t, const int j) QP_ALWAYS_INLINE { return PAIR ? (rbase + 2 * j + t) : (rbase + j); };
  auto colof = [&](const int ct, const int i) QP_ALWAYS_INLINE { return J + NCT * i + ct; };
  qp_double4 acc[NTJ][NCT];
  /* whole tile group strictly below the block and inside the matrix: no masks, 16-byte accesses */
  const bool full = QP_UNIFORM((int)((rbase >= J + NBW) && (rbase + 16 * NTJ <= n) && (J + NBW <= n))) != 0;
  if (full) {
#pragma unroll
    for (int ct = 0; ct < NCT; ct++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        double v[NTJ];
        qp_load_rows<NTJ>(L + (size_t)colof(ct, l4 + 4 * r) * ld + rowof(0, l15), v);
#pragma unroll
        for (int t = 0; t < NTJ; t++) acc[t][ct][r] = v[t];
      }
  } else {
#pragma unroll
    for (int t = 0; t < NTJ; t++) {
      const int row = rowof(t, l15);
#pragma unroll
      for (int ct = 0; ct < NCT; ct++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int col = colof(ct, l4 + 4 * r);
          acc[t][ct][r] = (row < n && col < n && row >= col) ? L[(size_t)col * ld + row] : 0.0;
        }
    }
  }
  /* panel fragment address: a pair starts at an even row (16-byte aligned: ld is a multiple of 8); groups that start
   * beyond the matrix read rows 0/1 instead (their results are never written) */
  const int rowb = PAIR ? ((rbase + 2 * l15 < n) ? rbase + 2 * l15 : 0) : ((rbase + l15 < n) ? rbase + l15 : n - 1);
  /* staging: thread e handles block row e % NBW of column e / NBW of the chunk, SE elements per thread */
  constexpr int SE = (QP_FKC * NBW + QP_T - 1) / QP_T;
  double sv[SE];
  auto stage_load = [&](const int kb) QP_ALWAYS_INLINE {
#pragma unroll
    for (int q = 0; q < SE; q++) {
      const int e = tid + q * QP_T, kk = e / NBW, i = e % NBW;
      const int k = kb + ((kk < QP_FKC) ? kk : 0);
      /* LDS slot i of a column holds block row i: the natural order already puts the NCT tile values of MFMA row
       * i / NCT next to each other */
      const int row = J + i;
      sv[q] = (kk < QP_FKC && row < n) ? -(L[(size_t)k * ld + row] * Dg[k]) : 0.0;
    }
  };
  auto stage_store = [&](const int buf) QP_ALWAYS_INLINE {
#pragma unroll
    for (int q = 0; q < SE; q++) {
      const int e = tid + q * QP_T, kk = e / NBW, i = e % NBW;
      if (kk < QP_FKC) F.As[buf][kk][i] = sv[q];
    }
  };
  constexpr int S = QP_FST, NH = QP_FKC / 4;
  static_assert(NH % S == 0, "QP_FST must divide QP_FKC / 4");
  double rb[S][NTJ];
  auto loadb = [&](const int st, const int k) QP_ALWAYS_INLINE { /* the panel re-read of the left-looking update */
    if (QP_NT_PANEL >= 4) qp_load_rows_nt<NTJ>(L + (size_t)(k + l4) * ld + rowb, rb[st]); else qp_load_rows<NTJ>(L + (size_t)(k + l4) * ld + rowb, rb[st]); };
  auto mma = [&](const int st, const int buf, const int h) QP_ALWAYS_INLINE {
    double pa[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ct++) pa[ct] = F.As[buf][4 * h + l4][NCT * l15 + ct];
#pragma unroll
    for (int ct = 0; ct < NCT; ct++)
#pragma unroll
      for (int t = 0; t < NTJ; t++) acc[t][ct] = QP_MFMA_F64(pa[ct], rb[st][t], acc[t][ct]);
  };
  __syncthreads(); /* the staging buffers may still be read by the previous call's last chunk */
  stage_load(k0);
  stage_store(0);
#pragma unroll
  for (int st = 0; st < S; st++) loadb(st, k0 + 4 * st);
  __syncthreads();
#pragma unroll 1
  for (int kb = k0, c = 0; kb < k1; kb += QP_FKC, c++) {
    const int buf = c & 1;
    const bool more = (kb + QP_FKC < k1);
    if (more) stage_load(kb + QP_FKC); /* in flight during this chunk's MFMAs */
#pragma unroll
    for (int h = 0; h < NH; h++) {
      const int st = h % S;
      mma(st, buf, h);
      const int kn = kb + 4 * (h + S); /* the fragment S steps ahead (next chunk included; past the end: a harmless re-read) */
      loadb(st, (kn < k1) ? kn : (k1 - 4));
    }
    if (more) stage_store(buf ^ 1);
    __syncthreads();
  }
  if (full) {
#pragma unroll
    for (int ct = 0; ct < NCT; ct++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        double v[NTJ];
#pragma unroll
        for (int t = 0; t < NTJ; t++) v[t] = acc[t][ct][r];
        qp_store_rows<NTJ>(L + (size_t)colof(ct, l4 + 4 * r) * ld + rowof(0, l15), v);
      }
  } else {
#pragma unroll
    for (int t = 0; t < NTJ; t++) {
      const int row = rowof(t, l15);
      if (row < n) {
#pragma unroll
        for (int ct = 0; ct < NCT; ct++)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int col = colof(ct, l4 + 4 * r);
            if (col < n && row >= col) L[(size_t)col * ld + row] = acc[t][ct][r];
          }
      }
    }
  }
}

/* Step (3) of dense_factor as its own function: one row per thread, the row's 32 panel entries live
 * in registers (64 VGPRs), L and 1/D of the diagonal block come as LDS broadcasts. */
#ifndef QP_NI_FROWS
#define QP_NI_FROWS QPNI
#endif
QP_NI_FROWS void factor_panel_rows(double *L_, char *lds_, const int n_, const int ld_, const int J_, const int jb_) {
  const int n = QP_UNIFORM(n_), ld = QP_UNIFORM(ld_), J = QP_UNIFORM(J_), jb = QP_UNIFORM(jb_); /* wave-uniform arguments back to SGPRs */
  qp_gdouble *L = (qp_gdouble *)L_;
  FactorLds QP_LDS_AS &F = *QP_LDS_ARG(FactorLds, lds_);
  constexpr int NB = QP_FNB;
  /* FULL = all 32 columns exist (every block but possibly the last): no per-column conditionals */
  auto rows = [&](auto full) QP_ALWAYS_INLINE {
    constexpr bool FULL = decltype(full)::value;
#pragma unroll 1
    for (int i = J + jb + threadIdx.x; i < n; i += QP_T) {
      qp_gdouble *base = L + (size_t)J * ld + i;
      double u[NB];
#pragma unroll
      for (int c = 0; c < NB; c++) u[c] = (FULL || c < jb) ? base[(size_t)c * ld] : 0.0;
#pragma unroll
      for (int c = 0; c < NB; c++) {
        if (FULL || c < jb) {
          double v = u[c];
#pragma unroll
          for (int c1 = 0; c1 < c; c1++) {
            v = QP_FMA(-u[c1], F.Ld[c][c1], v);
            if ((c1 & 7) == 7) QP_SCHED_BARRIER(); /* at most eight LDS operands in registers next to u[32] */
          }
          u[c] = v; /* un-normalised l*d */
        }
        QP_SCHED_BARRIER();
      }
      int ld2 = ld;
      QP_OPAQUE(ld2); /* store addresses are recomputed: 32 live column pointers would cost 64 VGPRs */
#pragma unroll
      for (int c = 0; c < NB; c++)
        if (FULL || c < jb) base[(size_t)c * ld2] = u[c] * F.dv[c];
    }
  };
  if (jb == NB) rows(std::true_type()); else rows(std::false_type());
}

/* Step (2) of dense_factor as its own function (own register allocation): the staged 32 x 32 diagonal block (rows
 * beyond the matrix = identity) is factorised by ONE wavefront with the whole block in registers: lane = row (both
 * half-waves hold the same rows), register c = column c.  Per column c: the pivot and the un-normalised column entries
 * p(c2, c) are wave-uniform values fetched with v_readlane (no LDS round trip on the chain), one division per column,
 * then p(r, c2) <- fma(-l(r, c), p(c2, c), p(r, c2)) for c2 > c: the same fma per entry, in the same order (c
 * ascending), as the LDS form it replaces (measured there: ~1 us per column; the mpc-160 factorisation spent 40 % here). */
#ifndef QP_NI_FDIAG
#define QP_NI_FDIAG QPNI
#endif
QP_NI_FDIAG void factor_diag_block(char *lds_) {
  FactorLds QP_LDS_AS &F = *QP_LDS_ARG(FactorLds, lds_);
  constexpr int NB = QP_FNB;
  const int r = (int)(threadIdx.x & (NB - 1));
  double p[NB];
#pragma unroll
  for (int c = 0; c < NB; c++) p[c] = F.Ld[r][c];
#pragma unroll
  for (int c = 0; c < NB; c++) {
    const double dc = qp_readlane(p[c], c);
    const double lic = p[c] / dc;
#pragma unroll
    for (int c2 = c + 1; c2 < NB; c2++) {
      const double s = qp_readlane(p[c], c2); /* p(c2, c), still un-normalised */
      p[c2] = QP_FMA(-lic, s, p[c2]);       /* meaningful for rows r >= c2; rows above the diagonal are never read */
    }
    p[c] = lic;
    if (r == c && threadIdx.x < NB) { F.dv[c] = 1.0 / dc; F.dg[c] = dc; } /* reciprocal pivot for the panel rows */
  }
  if (threadIdx.x < NB) {
#pragma unroll
    for (int c = 0; c < NB; c++) if (r > c) F.Ld[r][c] = p[c];
  }
}

template <int RPT>
#ifndef QP_NI_FACTOR
#define QP_NI_FACTOR QPNI
#endif
QP_NI_FACTOR void dense_factor(double *L_, double *Dg_, int n_, int ld_, char *lds_, int64_t *tdbg_) {
  const int n = QP_UNIFORM(n_), ld = QP_UNIFORM(ld_); /* wave-uniform arguments back to SGPRs */
  int64_t QP_LDS_AS *tdbg = (int64_t QP_LDS_AS *)tdbg_; /* the timers live in the kernel's static LDS */
  qp_gdouble *L = (qp_gdouble *)L_, *Dg = (qp_gdouble *)Dg_;
  FactorLds QP_LDS_AS &F = *QP_LDS_ARG(FactorLds, lds_);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int NB = QP_FNB;
  __syncthreads();
  long long tq0 = QP_CLOCK();
  for (int J = 0; J < n; J += NB) {
    const int jb = (n - J < NB) ? (n - J) : NB;
    /* ---- (1) panel update on the matrix cores.  Left-looking over 64-column super-blocks: at the start of a
     * super-block its 64 columns receive the contributions of ALL earlier columns (the panel is streamed once per
     * 64 columns); its second 32-column block then only needs the 32 columns just finished. ---------------- */
    {
      const bool super = (J % (2 * NB)) == 0;
      const int k0 = super ? 0 : J - NB, k1 = J;
      if (k1 > k0) {
        const int ntiles = (n - J + 15) / 16;
        char *stage = lds_ + ((sizeof(FactorLds) + 15) & ~(size_t)15);
        for (int tbase = 0; tbase < ntiles; tbase += QP_FNT * QP_NW) {
          const int rem = ntiles - tbase;
          const int ntj = (rem + QP_NW - 1) / QP_NW; /* same for every wavefront */
          if (super) {
            if (ntj <= 1) factor_panel_update<1, 4>(L_, Dg_, stage, n, ld, J, tbase, k0, k1);
            else factor_panel_update<2, 4>(L_, Dg_, stage, n, ld, J, tbase, k0, k1);
          } else {
            if (ntj <= 1) factor_panel_update<1, 2>(L_, Dg_, stage, n, ld, J, tbase, k0, k1);
            else factor_panel_update<2, 2>(L_, Dg_, stage, n, ld, J, tbase, k0, k1);
          }
        }
        if (tid == 0) { /* entries of L re-read: the panel rows once per pass group, the block rows once per pass */
          const long long npass = (ntiles + QP_FNT * QP_NW - 1) / (QP_FNT * QP_NW);
          tdbg[QPG_CNT_FACTOR_REREAD] += (long long)(n - J) * (k1 - k0) + npass * (super ? 2 * NB : NB) * (long long)(k1 - k0);
        }
      }
    }
    __syncthreads();
    if (tid == 0) { const long long tq1 = QP_CLOCK(); tdbg[4] += tq1 - tq0; tq0 = tq1; }
    /* ---- (2) diagonal block: staged in LDS by all threads (rows beyond the matrix = identity), factorised by
     * wavefront 0 in registers (factor_diag_block), written back by all threads ---------------------------------- */
    for (int e = tid; e < NB * NB; e += QP_T) {
      const int c = e / NB, r = e % NB;
      F.Ld[r][c] = (r >= c && r < jb) ? L[(size_t)(J + c) * ld + (J + r)] : ((r == c) ? 1.0 : 0.0);
    }
    __syncthreads();
    if (wid == 0) factor_diag_block(lds_);
    __syncthreads();
    for (int e = tid; e < jb * jb; e += QP_T) {
      const int c = e / jb, r = e % jb;
      if (r > c) L[(size_t)(J + c) * ld + (J + r)] = F.Ld[r][c];
    }
    if (tid < jb) Dg[J + tid] = F.dg[tid];
    if (tid == 0) { const long long tq1 = QP_CLOCK(); tdbg[5] += tq1 - tq0; tq0 = tq1; }
    /* ---- (3) rows below the block: l_ic = (p_ic - sum_{c1<c} u_ic1 l_c,c1) / d_c ---------------- */
    factor_panel_rows(L_, lds_, n, ld, J, jb);
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
#ifndef QP_SOLVE_FCH
#define QP_SOLVE_FCH 16 /* forward: column loads in flight per row */
#endif
#ifndef QP_SOLVE_BU
#define QP_SOLVE_BU 4 /* backward dots: 64-row chunks per iteration */
#endif
struct SolveLds {
  double tile[2][QP_SNB][QP_SNB + 1]; /* forward: the next diagonal block is fetched while the current one is solved */
  double part[QP_SNB];
};

#ifndef QP_NI_SOLVE
#define QP_NI_SOLVE QPNI
#endif
/* fwd_only != 0: only L y = b on the leading n x n block (n may be smaller than the panel), y returned unscaled:
 * the first step of a KKT row addition (L11 z = k12, Davis & Hager 2005) */
QP_NI_SOLVE void dense_solve(const double *L_, const double *Dg_, int n_, int ld_, double *xg_, char *lds_, int lds_bytes, int64_t *tdbg_ = nullptr, int fwd_only_ = 0) {
  const int n = QP_UNIFORM(n_), ld = QP_UNIFORM(ld_), fwd_only = QP_UNIFORM(fwd_only_); /* wave-uniform arguments back to SGPRs */
  int64_t QP_LDS_AS *tdbg = (int64_t QP_LDS_AS *)tdbg_; /* null or the timers in the kernel's static LDS */
  const qp_gdouble *L = (const qp_gdouble *)L_, *Dg = (const qp_gdouble *)Dg_;
  qp_gdouble *xg = (qp_gdouble *)xg_;
  char QP_LDS_AS *lds = QP_LDS_ARG(char, lds_);
  SolveLds QP_LDS_AS &T = *(SolveLds QP_LDS_AS *)lds;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int NB = QP_SNB;
  /* the right-hand side always fits: 8 n <= lds_bytes (checked by the host, qpg_batch_create) */
  double QP_LDS_AS *xs = (double QP_LDS_AS *)(lds + sizeof(SolveLds));
  (void)lds_bytes;
  __syncthreads();
  for (int i = tid; i < n; i += QP_T) xs[i] = xg[i];
  long long ts0 = QP_CLOCK();
  if (fwd_only != 2) { /* 2: x already holds y = L^{-1} b (the forward substitution was fused into the last update sweep) */
  /* forward: L y = b */
  /* strict lower triangle of a diagonal block into a tile, zero elsewhere (the block solve reads it
   * unconditionally); t0/nt = this thread's index / count among the loading threads */
  auto load_tile = [&](const int J, const int buf, const int t0, const int nt) QP_ALWAYS_INLINE {
    const int jb = (n - J < NB) ? (n - J) : NB;
    for (int e = t0; e < NB * NB; e += nt) {
      const int c = e / NB, r = e % NB;
      T.tile[buf][r][c] = (r > c && r < jb) ? L[(size_t)(J + c) * ld + (J + r)] : 0.0;
    }
  };
  load_tile(0, 0, tid, QP_T);
  for (int J = 0; J < n; J += NB) {
    const int jb = (n - J < NB) ? (n - J) : NB;
    const int cur = (J / NB) & 1;
    __syncthreads();
    /* wavefronts 1.. fetch the next diagonal block while wavefront 0 solves this one */
    if (QP_NW > 1 && wid != 0 && J + NB < n) load_tile(J + NB, cur ^ 1, tid - 64, QP_T - 64);
    if (wid == 0) { /* lane = row of the block; pivots broadcast by readlane; the row of L comes from LDS
                     * eight entries at a time (small register arrays: the kernel runs under a 128-VGPR cap) */
      const int ln = QP_FRESH_LANE(lane) & (NB - 1);
      double v = (lane < jb) ? xs[J + lane] : 0.0;
      double tr[2][8];
#pragma unroll
      for (int c = 0; c < 8; c++) tr[0][c] = T.tile[cur][ln][c];
#pragma unroll
      for (int cb = 0; cb < NB; cb += 8) {
        if (cb + 8 < NB) {
#pragma unroll
          for (int c = 0; c < 8; c++) tr[((cb >> 3) + 1) & 1][c] = T.tile[cur][ln][cb + 8 + c];
        }
#pragma unroll
        for (int c = 0; c < 8; c++) {
          const double yc = qp_readlane(v, cb + c);
          v = QP_FMA(-tr[(cb >> 3) & 1][c], yc, v);
        }
      }
      if (lane < jb) xs[J + lane] = v;
      if (QP_NW == 1 && J + NB < n) load_tile(J + NB, cur ^ 1, tid, QP_T);
    }
    __syncthreads();
    if (!QP_PANEL_TIMING && tdbg && tid == 0) { const long long t = QP_CLOCK(); tdbg[8] += t - ts0; ts0 = t; }
    for (int i = J + jb + tid; i < n; i += QP_T) { /* QP_SOLVE_FCH independent column loads in flight per row */
      double acc = xs[i];
#pragma unroll
      for (int ch = 0; ch < NB; ch += QP_SOLVE_FCH) {
        double lv[QP_SOLVE_FCH];
#pragma unroll
        for (int cc = 0; cc < QP_SOLVE_FCH; cc++) lv[cc] = QP_LDNT(3, &L[(size_t)(J + ((ch + cc < jb) ? ch + cc : jb - 1)) * ld + i]);
#pragma unroll
        for (int cc = 0; cc < QP_SOLVE_FCH; cc++) if (ch + cc < jb) acc = QP_FMA(-lv[cc], xs[J + ch + cc], acc);
      }
      xs[i] = acc;
    }
    if (!QP_PANEL_TIMING && tdbg && tid == 0) { const long long t = QP_CLOCK(); tdbg[9] += t - ts0; ts0 = t; }
  }
  }
  __syncthreads();
  if (fwd_only == 1) {
    for (int i = tid; i < n; i += QP_T) xg[i] = xs[i];
    __syncthreads();
    return;
  }
  for (int i = tid; i < n; i += QP_T) xs[i] = xs[i] / Dg[i];
  __syncthreads();
  /* backward: L' x = z */
  const int Jlast = ((n - 1) / NB) * NB;
  for (int J = Jlast; J >= 0; J -= NB) {
    const int jb = (n - J < NB) ? (n - J) : NB;
    /* the diagonal block's entries are requested first (registers), so their latency overlaps the dots */
    constexpr int TE = (QP_SNB * QP_SNB + QP_T - 1) / QP_T;
    double tv[TE];
#pragma unroll
    for (int k = 0; k < TE; k++) {
      const int e = tid + k * QP_T, c = e / NB, r = e % NB;
      tv[k] = (e < NB * NB && r > c && r < jb) ? L[(size_t)(J + c) * ld + (J + r)] : 0.0;
    }
    { /* partial dots of the block's columns with x below the block: the NB/NW columns of this
       * wavefront advance together, 4 x 64 rows per iteration => 16 independent loads per lane */
      constexpr int CW = QP_SNB / QP_NW > 0 ? QP_SNB / QP_NW : 1;
      double sacc[CW];
#pragma unroll
      for (int q = 0; q < CW; q++) sacc[q] = 0.0;
      constexpr int BU = QP_SOLVE_BU; /* row chunks of 64 per iteration: CW * BU independent loads per lane */
      for (int i0 = J + jb; i0 < n; i0 += 64 * BU) {
        double xv[BU];
#pragma unroll
        for (int u = 0; u < BU; u++) { const int i = i0 + 64 * u + lane; xv[u] = (i < n) ? xs[i] : 0.0; }
#pragma unroll
        for (int q = 0; q < CW; q++) {
          const int c = wid + q * QP_NW;
          const qp_gdouble *col = L + (size_t)(J + ((c < jb) ? c : jb - 1)) * ld;
          double lv[BU];
#pragma unroll
          for (int u = 0; u < BU; u++) { const int i = i0 + 64 * u + lane; lv[u] = QP_LDNT(3, &col[(i < n) ? i : n - 1]); }
#pragma unroll
          for (int u = 0; u < BU; u++) sacc[q] = QP_FMA(lv[u], xv[u], sacc[q]);
        }
      }
#pragma unroll
      for (int q = 0; q < CW; q++) {
        const int c = wid + q * QP_NW;
        const double sv = wave_sum(sacc[q]);
        if (lane == 0 && c < jb) T.part[c] = sv;
      }
    }
    if (!QP_PANEL_TIMING && tdbg && tid == 0) { const long long t = QP_CLOCK(); tdbg[10] += t - ts0; ts0 = t; }
#pragma unroll
    for (int k = 0; k < TE; k++) {
      const int e = tid + k * QP_T, c = e / NB, r = e % NB;
      if (e < NB * NB) T.tile[0][r][c] = tv[k];
    }
    __syncthreads();
    if (wid == 0) { /* lane = column of the block: needs L(J+c, J+lane) for c > lane, eight at a time from LDS */
      const int ln = QP_FRESH_LANE(lane) & (NB - 1);
      double v = (lane < jb) ? (xs[J + lane] - T.part[lane]) : 0.0;
      double tc[2][8];
#pragma unroll
      for (int c = 0; c < 8; c++) tc[0][c] = T.tile[0][NB - 1 - c][ln];
#pragma unroll
      for (int cb = 0; cb < NB; cb += 8) { /* columns NB-1-cb .. NB-8-cb, descending */
        if (cb + 8 < NB) {
#pragma unroll
          for (int c = 0; c < 8; c++) tc[((cb >> 3) + 1) & 1][c] = T.tile[0][NB - 1 - (cb + 8 + c)][ln];
        }
#pragma unroll
        for (int c = 0; c < 8; c++) {
          const double xc = qp_readlane(v, NB - 1 - (cb + c));
          v = QP_FMA(-tc[(cb >> 3) & 1][c], xc, v);
        }
      }
      if (lane < jb) xs[J + lane] = v;
    }
    __syncthreads();
    if (!QP_PANEL_TIMING && tdbg && tid == 0) { const long long t = QP_CLOCK(); tdbg[11] += t - ts0; ts0 = t; }
  }
  for (int i = tid; i < n; i += QP_T) xg[i] = xs[i];
  __syncthreads();
}

/* ---------------------------------------------------------------------------------------------
 * The same factorisation and triangular solves, one block column per call and the rows dealt to SEVERAL workgroups: the
 * pieces the host chains (one launch per piece, the kernel boundary is the synchronisation) when a single large QP has
 * the chip to itself (qpg "coop" mode, qpalm_capi.inc: coop_solve).  A batch never comes here.  Per entry the arithmetic
 * of the factorisation and of the forward substitution is the one of dense_factor / dense_solve (k ascending, one fma
 * per k); the backward substitution is done in outer-product form (no reduction across workgroups).
 * ------------------------------------------------------------------------------------------- */
/* (1) right-looking trailing update after block column J (32 finished columns k = J .. J+31): every later block column Jc gets
 * P(Jc:n, Jc:Jc+32) -= L(Jc:n, k) D(k) L(Jc:Jc+32, k)' for those 32 k, the (block column, row pass) items dealt round-robin to the
 * workgroups.  The same MFMA routine as dense_factor's left-looking update, called with a 32-wide k range: an entry receives
 * its k terms in ascending order, four per MFMA, exactly as there (stores and loads between the blocks are lossless), so the
 * factor is bit-identical -- but nothing here waits on a chain of J / 32 staged chunks, and all workgroups have work. */
QPN void co_factor_trailing(double *L_, double *Dg_, int n, int ld, char *lds_, int J, int wg, int nwg) {
  const int NB = QP_FNB;
  if (J + NB >= n) return;
  char *stage = lds_ + ((sizeof(FactorLds) + 15) & ~(size_t)15);
  int item = 0;
  for (int Jc = J + NB; Jc < n; Jc += NB) {
    const int ntiles = (n - Jc + 15) / 16;
    for (int tbase = 0; tbase < ntiles; tbase += QP_FNT * QP_NW, item++) {
      if (item % nwg != wg) continue;
      const int rem = ntiles - tbase;
      const int ntj = (rem + QP_NW - 1) / QP_NW;
      if (ntj <= 1) factor_panel_update<1, 2>(L_, Dg_, stage, n, ld, Jc, tbase, J, J + NB);
      else factor_panel_update<2, 2>(L_, Dg_, stage, n, ld, Jc, tbase, J, J + NB);
    }
  }
}
/* (2) the 32 x 32 diagonal block (one workgroup) */
QPN void co_factor_diag(double *L_, double *Dg_, int n, int ld, char *lds_, int J) {
  qp_gdouble *L = (qp_gdouble *)L_, *Dg = (qp_gdouble *)Dg_;
  FactorLds QP_LDS_AS &F = *QP_LDS_ARG(FactorLds, lds_);
  const int NB = QP_FNB, tid = threadIdx.x, wid = tid >> 6;
  const int jb = (n - J < NB) ? (n - J) : NB;
  __syncthreads();
  for (int e = tid; e < NB * NB; e += QP_T) {
    const int c = e / NB, r = e % NB;
    F.Ld[r][c] = (r >= c && r < jb) ? L[(size_t)(J + c) * ld + (J + r)] : ((r == c) ? 1.0 : 0.0);
  }
  __syncthreads();
  if (wid == 0) factor_diag_block(lds_);
  __syncthreads();
  for (int e = tid; e < jb * jb; e += QP_T) {
    const int c = e / jb, r = e % jb;
    if (r > c) L[(size_t)(J + c) * ld + (J + r)] = F.Ld[r][c];
  }
  if (tid < jb) Dg[J + tid] = F.dg[tid];
  __syncthreads();
}
/* (3) rows below the block: the factorised diagonal block comes back from HBM, the rows are dealt thread by thread over the grid */
QPN void co_factor_rows(double *L_, const double *Dg_, int n, int ld, char *lds_, int J, int wg, int nwg) {
  qp_gdouble *L = (qp_gdouble *)L_;
  const qp_gdouble *Dg = (const qp_gdouble *)Dg_;
  FactorLds QP_LDS_AS &F = *QP_LDS_ARG(FactorLds, lds_);
  constexpr int NB = QP_FNB;
  const int tid = threadIdx.x;
  const int jb = (n - J < NB) ? (n - J) : NB;
  if (J + jb >= n) return;
  __syncthreads();
  for (int e = tid; e < NB * NB; e += QP_T) {
    const int c = e / NB, r = e % NB;
    F.Ld[r][c] = (r > c && r < jb) ? L[(size_t)(J + c) * ld + (J + r)] : 0.0;
  }
  if (tid < NB) F.dv[tid] = (tid < jb) ? 1.0 / Dg[J + tid] : 1.0;
  __syncthreads();
  for (int i = J + jb + wg * QP_T + tid; i < n; i += QP_T * nwg) {
    qp_gdouble *base = L + (size_t)J * ld + i;
    double u[NB];
#pragma unroll
    for (int c = 0; c < NB; c++) u[c] = (c < jb) ? base[(size_t)c * ld] : 0.0;
#pragma unroll
    for (int c = 0; c < NB; c++) {
      if (c < jb) {
        double v = u[c];
#pragma unroll
        for (int c1 = 0; c1 < c; c1++) {
          v = QP_FMA(-u[c1], F.Ld[c][c1], v);
          if ((c1 & 7) == 7) QP_SCHED_BARRIER();
        }
        u[c] = v;
      }
      QP_SCHED_BARRIER();
    }
    int ld2 = ld;
    QP_OPAQUE(ld2);
#pragma unroll
    for (int c = 0; c < NB; c++)
      if (c < jb) base[(size_t)c * ld2] = u[c] * F.dv[c];
  }
}
/* The triangular solves of coop mode work on blocks of CO_SNB = 64 columns: a launch costs its latency chain (stage the diagonal
 * block, solve it on one wavefront, stream the rest), so fewer, wider blocks: 8.3 us per 32-column launch before, measured in
 * DESIGN section 7.  One wavefront = 64 lanes = the rows (forward) or columns (backward) of the block. */
#define CO_SNB 64
struct CoSolveLds {
  double tile[CO_SNB][CO_SNB + 1];
  double part[CO_SNB];
};
/* forward substitution, block J: every workgroup solves the 64 x 64 block itself (cheap, and the result is needed by all), then
 * takes its share of the rows below: x_i -= sum_c l_ic y_c, c ascending, one fma per c (dense_solve's forward arithmetic) */
/* (the block's own result goes to xo: every workgroup reads the block of x while workgroup 0 would overwrite it) */
QPN void co_solve_forward(const double *L_, int n, int ld, double *x_, double *xo_, char *lds_, int J, int wg, int nwg) {
  const qp_gdouble *L = (const qp_gdouble *)L_;
  qp_gdouble *x = (qp_gdouble *)x_, *xo = (qp_gdouble *)xo_;
  CoSolveLds QP_LDS_AS &T = *QP_LDS_ARG(CoSolveLds, lds_);
  constexpr int NB = CO_SNB;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int jb = (n - J < NB) ? (n - J) : NB;
  __syncthreads();
  for (int e = tid; e < NB * NB; e += QP_T) {
    const int c = e / NB, r = e % NB;
    T.tile[r][c] = (r > c && r < jb) ? L[(size_t)(J + c) * ld + (J + r)] : 0.0;
  }
  __syncthreads();
  if (wid == 0) {
    double v = (lane < jb) ? x[J + lane] : 0.0;
#pragma unroll 8
    for (int c = 0; c < NB; c++) {
      const double yc = qp_readlane(v, c);
      v = QP_FMA(-T.tile[lane][c], yc, v);
    }
    T.part[lane] = v;
    if (wg == 0 && lane < jb) xo[J + lane] = v;
  }
  __syncthreads();
  for (int i = J + jb + wg * QP_T + tid; i < n; i += QP_T * nwg) {
    double acc = x[i];
#pragma unroll 1
    for (int c0 = 0; c0 < NB; c0 += 16) { /* 16 column loads in flight per row */
      double lv[16];
#pragma unroll
      for (int c = 0; c < 16; c++) lv[c] = QP_LDNT(3, &L[(size_t)(J + ((c0 + c < jb) ? c0 + c : jb - 1)) * ld + i]);
#pragma unroll
      for (int c = 0; c < 16; c++) if (c0 + c < jb) acc = QP_FMA(-lv[c], T.part[c0 + c], acc);
    }
    x[i] = acc;
  }
}
/* backward substitution L' x = z in outer-product form, block J (descending): every workgroup finishes x_J itself, then takes its
 * share of the COLUMNS before the block: z_c -= sum_r l(J + r, c) x_r.  A quarter wavefront per column (64 contiguous rows of the
 * column: four per lane), fixed reduction tree: no reduction across workgroups, results independent of the grid. */
QPN void co_solve_backward(const double *L_, int n, int ld, double *x_, double *xo_, char *lds_, int J, int wg, int nwg) {
  const qp_gdouble *L = (const qp_gdouble *)L_;
  qp_gdouble *x = (qp_gdouble *)x_, *xo = (qp_gdouble *)xo_;
  CoSolveLds QP_LDS_AS &T = *QP_LDS_ARG(CoSolveLds, lds_);
  constexpr int NB = CO_SNB;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int jb = (n - J < NB) ? (n - J) : NB;
  __syncthreads();
  for (int e = tid; e < NB * NB; e += QP_T) {
    const int c = e / NB, r = e % NB;
    T.tile[r][c] = (r > c && r < jb) ? L[(size_t)(J + c) * ld + (J + r)] : 0.0;
  }
  __syncthreads();
  if (wid == 0) { /* lane = column of the block: x_c = z_c - sum_{r > c} l_rc x_r, r descending */
    double v = (lane < jb) ? x[J + lane] : 0.0;
#pragma unroll 8
    for (int r = NB - 1; r >= 0; r--) {
      const double xr = qp_readlane(v, r);
      v = QP_FMA(-T.tile[r][lane], xr, v); /* tile[r][c] = l(J + r, J + c), zero unless r > c */
    }
    T.part[lane] = (lane < jb) ? v : 0.0;
    if (wg == 0 && lane < jb) xo[J + lane] = v;
  }
  __syncthreads();
  const int grp = tid >> 4, gl = tid & 15, ngrp = QP_T / 16;
  double xb[4];
#pragma unroll
  for (int k = 0; k < 4; k++) xb[k] = T.part[4 * gl + k];
  for (int c0 = (wg * ngrp); c0 < J; c0 += ngrp * nwg) {
    const int c = c0 + grp;
    double s = 0.0;
    if (c < J) {
      const int r0 = J + 4 * gl;
      double lv[4];
#pragma unroll
      for (int k = 0; k < 4; k++) lv[k] = (r0 + k < n) ? QP_LDNT(3, &L[(size_t)c * ld + r0 + k]) : 0.0;
      s = QP_FMA(lv[1], xb[1], lv[0] * xb[0]);
      s = QP_FMA(lv[2], xb[2], s);
      s = QP_FMA(lv[3], xb[3], s);
    }
    for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (c < J && gl == 0) x[c] = x[c] - s;
  }
}

/* ---------------------------------------------------------------------------------------------
 * dense_updown: L D L' <- L D L' + sum_r s_r w_r w_r'   (s_r = +1 update, -1 downdate), the columns
 * w_r being columns cols[r] of At_sqrt_sigma.  Up to K ranks are applied per sweep over the panel,
 * so the panel is read and written once per K ranks (16 B per entry and sweep).
 *
 * Per column j and rank r (Davis & Hager method C1), with alpha_r carried along the columns:
 *     p = s w_j^2/alpha ; d_new = d + p ; gamma = -s w_j/(alpha d_new) ; alpha <- alpha d_new/d
 *     for i > j:  w_i -= w_j l_ij ;  l_ij -= gamma w_i
 * Thread t owns rows RPT t .. RPT t + RPT-1 and keeps their K running w values in registers for the
 * whole sweep.  Per block column of 32 the recurrence on the block's own rows is a serial chain (one
 * wavefront: lane = row, everything in registers; rank-indexed scalars are produced with lane = rank,
 * one reciprocal per column on the critical path, DPP row scans); it publishes the (w_j, gamma) table
 * through LDS and every thread applies the table to its rows below the block with L streamed from HBM.
 *
 * Look-ahead: wavefront 0 is the panel wave and runs ONE BLOCK AHEAD of the others.  In phase s
 *     wavefront 0 : applies table s-1 to the 32 rows of block s (handed over by their owners), runs the
 *                   recurrence of block s -> table s, writes the diagonal block back;
 *     the others  : apply table s-1 to their rows below block s, hand the rows of block s+1 over,
 *                   fetch diagonal block s+1 into LDS;
 * one barrier per phase; tables, hand-over buffers and diagonal blocks are double buffered.  The
 * arithmetic per entry is the same sequence of FMAs as without look-ahead.
 * ------------------------------------------------------------------------------------------- */
#define QP_UNB 32
#ifndef QP_UNBS
#define QP_UNBS QP_UNB /* block columns of dense_updown's sweep (dense_updown_big and the coop kernels keep QP_UNB) */
#endif
#ifndef QP_UNBS32
#define QP_UNBS32 16 /* ... of its 32-rank form: 16-column blocks, the LDS tables of 32 ranks x 32 columns x 2 buffers do not fit next to the rest */
#endif
#ifndef QP_USQ
#define QP_USQ 1 /* 1: the owners stage the square L(block s+1 rows, block s columns) in LDS one phase ahead and write the finished
                    diagonal blocks back, so that the panel wave (the serial chain of the sweep) never waits for HBM; costs
                    16 KB of LDS.  0: the panel wave streams that square from HBM itself (the 256-thread instance: 38 KB LDS). */
#endif
#ifndef QP_TQD
#define QP_TQD 4 /* columns of L in flight per thread in the trailing-row loop (register queue) */
#endif
/* Variants of this sweep that were built, measured slower (or equal) on MI355X and removed from this file in round 4 -- helper
 * wave (QP_UHELP / QP_HSPLIT), rank-split panel wave (QP_PSPLIT) and rank-split table application (QP_ASPLIT / QP_APF), delayed /
 * throttled owners (QP_ODELAY / QP_OSLEEP), the knock-out timing builds (QP_KO, QP_PROBE_NO_OWNER_TABLE) -- live as a patch in
 * tools/variants/ (tools/evidence/build_variant.sh applies it); DESIGN.md section 7 keeps their numbers. */
#define QP_CWG(U, buf, col) (U).cwg[buf][col]
template <int K> struct UpdownCfg {
  static constexpr int NB = (K > 16) ? QP_UNBS32 : QP_UNBS; /* columns per block = per phase */
  static constexpr int G = (K + 15) / 16;                    /* rank groups of the recurrence: one DPP row (16 lanes) each */
  static constexpr int KG = K / G;                            /* ranks per group: 8 or 16 */
};
template <int RPT, int K>
struct UpdownLds {           /* [2]: look-ahead double buffers, indexed by block parity */
  static constexpr int NB = UpdownCfg<K>::NB;
  double Ld[2][NB][NB + 1];
  double Lsq[QP_USQ ? 2 : 1][QP_USQ ? NB : 1][NB]; /* [parity][column][row]: the NB x NB square of L that the serial chain of the next phase
                                                      applies a table to (staged by the owners, see the phase loop) */
  double Wd[2][NB][K + 1];     /* running w of the rows of block b (+ the row's substitution accumulator): owners -> panel wave */
  double cwg[2][NB][K][2];     /* (-w_j, -gamma) per column and rank, read as one 16-byte broadcast: the two tables, by block parity */
  double stash[RPT][K][64];    /* wavefront 0 parks the running w of its own rows here while it is the panel wave */
  double Wt[K];
  double dd[2][NB];
  double ys[2][NB];            /* fused forward substitution: the solved block of the right-hand side, by block parity */
  double stash_acc[RPT][64];   /* ... and wavefront 0's own row accumulators while it is the panel wave */
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
/* The pivot of a column after rank r of a sweep is d_0 + p_0 + ... + p_r.  The sweeps take it as d_0 + prefix_tree(p) over the 16 lanes of a
 * DPP row (four steps); this is the RUNNING pivot the reference computes, rank after rank: d_r = d_{r-1} + p_r, starting from d_0 (fifteen
 * dependent steps).  Same value in exact arithmetic -- but when downdates take a pivot towards zero, the sum of the p's alone is as large
 * as d_0, its rounding error (eps |d_0|) lands on a result that may be 1e-9 |d_0|, and gamma = w / d_new carries it into the whole column;
 * the running pivot shrinks WITH the partial sums, so each addition rounds relative to what is left.  Round 5: fuzz case 682 / 1 (an
 * indefinite factor polluted to 7e-7 for two hundred iterations: 13 339 iterations where the oracle needs 2459) and the LP case 701 / 128
 * (fifty rows leaving at once, lambda_min(H) -> 1e-7: backward error of the update 1.4e-8 against 7e-16).  Used for QPs whose factor can
 * get near-singular (QpShared::seq_ranks: nonconvex, or Q without a positive diagonal).  Returns d_new of this lane's rank; lanes of ranks
 * that are not in the sweep carry p = 0 and pass the pivot on. */
template <int KG> QPD double qp_rank_pivots_seq(const double p, const int ln, const double d0) {
  double run = d0 + p; /* rank 0's; the other lanes are overwritten below */
#pragma unroll
  for (int sq = 1; sq < KG; sq++) {
    const double up = qp_row_shr<1>(run);
    if ((ln & 15) == sq) run = up + p;
  }
  return run;
}
/* d_new (after this lane's rank) and d_prev (before it) of one rank group in one column; lane & 15 = rank, every 16-lane row of the wavefront holds the
 * same values.  mode: QP_PIV_* (qpalm_device.h).  Guarded tree: the tree's error in d_r is eps max(|d_0|, |partial sums|), the running pivot's is
 * eps |d_r|; they differ only where |d_r| << |d_0|, so a column in which some |d_r| < 2^-8 |d_0| (or is NaN) is summed again the reference's way.
 * The tree's result elsewhere carries at most 2^8 eps = 3e-14 of relative error in a pivot.  Returns 1 if the column was re-summed. */
template <int KG> QPD int qp_rank_pivots(const double p, const int ln, const double d0, const int mode, double &dnew, double &dprev) {
  if (mode != QP_PIV_SEQ) {
    double incl = p;
    if (KG > 1) incl += qp_row_shr<1>(incl);
    if (KG > 2) incl += qp_row_shr<2>(incl);
    if (KG > 4) incl += qp_row_shr<4>(incl);
    if (KG > 8) incl += qp_row_shr<8>(incl);
    const double excl = qp_row_shr<1>(incl);
    dnew = d0 + incl; dprev = d0 + excl;
    if (mode == QP_PIV_TREE) return 0;
    const bool danger = !(__builtin_fabs(dnew) >= 0x1p-8 * __builtin_fabs(d0)); /* (a NaN compares false: danger) */
    if (__ballot(danger ? 1 : 0) == 0ull) return 0;
  }
  dnew = qp_rank_pivots_seq<KG>(p, ln, d0);
  const double sh = qp_row_shr<1>(dnew);
  dprev = ((ln & 15) == 0) ? d0 : sh;
  return (mode != QP_PIV_SEQ) ? 1 : 0;
}


/* lane R of every row of 16 lanes to all lanes of that row (DPP row_newbcast:R): no LDS, no SGPR */
#ifdef QPALM_EMU
template <int R> QPD double qp_row_bcast(double v) { const int lane = threadIdx.x & 63; return emu_exchange(v, (lane & ~15) | R); }
#else
template <int R> QPD double qp_row_bcast(double v) { /* ONE v_mov_b64_dpp: row_newbcast is the DPP control gfx90a+ allows on 64-bit moves */
  return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + R, 0xf, 0xf, true);
}
#endif
/* ranks r = R0 .. R1-1 of one rank group applied to a row: w_{OFF+r} += c0_r l, l += c1_r w_{OFF+r}, the pair (c0_r, c1_r) living in
 * lane r of every 16-lane row */
#ifdef QPALM_EMU
template <int K, int OFF, int R0, int R1, bool FIRST = true>
QPD void qp_apply_ranks_dpp(const double cw0, const double cw1, double (&wrow)[K], double &l) { /* same values, one fiber round instead of 2 (R1 - R0) */
  const int lane = threadIdx.x & 63;
  emu_publish2(cw0, cw1);
  for (int r = R0; r < R1 && OFF + r < K; r++) {
    const double c0 = emu_peek((lane & ~15) | r, 0), c1 = emu_peek((lane & ~15) | r, 1);
    wrow[OFF + r] = QP_FMA(c0, l, wrow[OFF + r]);
    l = QP_FMA(c1, wrow[OFF + r], l);
  }
  emu_wave_sync();
}
#else
#ifndef QP_FMAC_DPP
#define QP_FMAC_DPP 1 /* the broadcast folded into the FMA: v_fmac_f64_dpp (VOP2 + DPP row_newbcast, gfx90a+) -- two instructions per rank instead of
                         four.  The compiler does not form it from v_mov_b64_dpp + v_fmac_f64 by itself (ROCm 7.2), hence the inline assembly. */
#endif
/* Eight ranks in ONE assembly statement: w_i <- fma(pair_x of lane R0 + i, l, w_i); l <- fma(pair_y of lane R0 + i, w_i, l), i = 0..7.
 * (One statement per instruction would cost an s_nop between any two of them: the compiler's hazard recogniser assumes the worst
 * of inline assembly on gfx950 -- dst_sel / cvt-scale forwarding -- and that takes back half of what the fused form saves.)
 * A DPP operand must not have been written by one of the two preceding VALU instructions, and the compiler's hazard recogniser does
 * not look into the statement: every statement therefore starts with the `s_nop 1` itself, whatever precedes it (a register copy or a
 * spill reload of the pair placed right in front of a statement would otherwise be read stale: ADVICE round 4).  The outputs are
 * early-clobber: the pair registers (%9, %10) are read again after w_0 and l have been written. */
#define QP_FD_STEP(i, n) "v_fmac_f64_dpp %" #i ", %9, %8 row_newbcast:%" #n " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
                         "v_fmac_f64_dpp %8, %10, %" #i " row_newbcast:%" #n " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define QP_FD_EIGHT QP_FD_STEP(0, 11) QP_FD_STEP(1, 12) QP_FD_STEP(2, 13) QP_FD_STEP(3, 14) QP_FD_STEP(4, 15) QP_FD_STEP(5, 16) QP_FD_STEP(6, 17) QP_FD_STEP(7, 18)
template <int R0> QPD void qp_fmac_bcast8(double *w, double &l, const double cw0, const double cw1) {
  asm("s_nop 1\n\t" QP_FD_EIGHT
      : "+&v"(w[0]), "+&v"(w[1]), "+&v"(w[2]), "+&v"(w[3]), "+&v"(w[4]), "+&v"(w[5]), "+&v"(w[6]), "+&v"(w[7]), "+&v"(l)
      : "v"(cw0), "v"(cw1), "n"(R0), "n"(R0 + 1), "n"(R0 + 2), "n"(R0 + 3), "n"(R0 + 4), "n"(R0 + 5), "n"(R0 + 6), "n"(R0 + 7));
}
template <int K, int OFF, int R0, int R1, bool FIRST = true>
QPD void qp_apply_ranks_dpp(const double cw0, const double cw1, double (&wrow)[K], double &l) {
  if constexpr (QP_FMAC_DPP && R1 - R0 == 8 && OFF + R1 <= K) {
    qp_fmac_bcast8<R0>(&wrow[OFF + R0], l, cw0, cw1);
  } else if constexpr (R0 < R1 && OFF + R0 < K) {
    const double c0 = qp_row_bcast<R0>(cw0), c1 = qp_row_bcast<R0>(cw1);
    wrow[OFF + R0] = QP_FMA(c0, l, wrow[OFF + R0]);
    l = QP_FMA(c1, wrow[OFF + R0], l);
    qp_apply_ranks_dpp<K, OFF, R0 + 1, R1, false>(cw0, cw1, wrow, l);
  }
}
#endif
/* QP_RECUR_DPP = 1: the diagonal-block recurrence of the update sweep hands the (-w, -gamma) pair of rank r to the row lanes by DPP
 * row broadcast (the rank scalars are computed in every 16-lane row, lane & 15 = rank) instead of through the LDS table: one LDS
 * round trip per column instead of two.  The table is still written for the wavefronts that own the rows below the block. */
#ifndef QP_RECUR_DPP
#define QP_RECUR_DPP 1
#endif
#ifndef QP_AQD
#define QP_AQD 4 /* columns in flight (entry of the staged square + its pairs) in the panel wave's "table s-1 on the rows of block s" loop */
#endif
#ifndef QP_OWN_A
#define QP_OWN_A 1 /* staged square (QP_USQ): the entries L(rows of block s, columns of block s-1) are finished and written to HBM by the OWNERS of
                      those rows, which apply table s-1 to them like to every other row below; the panel wave applies the same table to its LDS
                      copy for the running w it needs (same FMAs in the same order: the same bits) and stores nothing.  Its one global store
                      per column waited ~100 ns to be accepted by the CU's memory pipeline next to the owners' streams -- longer than the
                      column's FMAs (tools/variants/README.md, "panel store probes": 312 -> 219 us per sweep with the store left out). */
#endif
#ifndef QP_SQ_WB
#define QP_SQ_WB 0 /* 1 (measured in round 4: no gain, 5186-5191 against 5206-5216 QP/s same box): the panel wave writes the entries of the staged square back into LDS, not to HBM (see its loop) */
#endif
#ifndef QP_PANEL_A_DPP
#define QP_PANEL_A_DPP (K <= 16) /* the same hand-over when the panel wave applies the finished table s-1 to the rows of block s (see there);
                                    measured: K = 16 panel wave 57.0 -> 55.0 ms per QP; K = 32: slower (4.43 vs 4.65 k QP/s), so not there */
#endif
#ifndef QP_OWNER_DPP32
#define QP_OWNER_DPP32 1 /* ... and in the owners' loop of the 32-rank form (one row per lane: a broadcast LDS read per rank and column serves two FMAs
                            only, the LDS pipe of the CU is the bound; two lane-indexed reads per column + DPP broadcasts instead) */
#endif

/* ---------------------------------------------------------------------------------------------
 * K = 32 (round 4): TWICE the ranks per pass over the panel = half the panel traffic per rank.  Two things make that fit:
 *  - Registers: a thread keeps the 32 running values of ONE row (RPT = 1: 64 VGPRs, what two rows of 16 took), so the rows are
 *    handled in PASSES of QP_T rows.  Pass p owns rows [R0, R1) = [p QP_T, (p+1) QP_T): first its rows receive the tables of all
 *    earlier columns [J0, R0) -- produced by the earlier passes, exported to HBM (512 KB for n = 1000: L2 / Infinity Cache),
 *    applied by ALL wavefronts with no serial chain at all ("rectangle") --, then the look-ahead sweep runs on the pass's own
 *    triangle, columns [R0, R1), and exports its tables.  Every entry of L is still read and written once per sweep.
 *  - The recurrence: the 32 ranks of a column are two groups of 16 handled one after the other in the same lanes (lane & 15 =
 *    rank within the group; the second group starts from the pivot the first one leaves).  Per entry this is the SAME sequence
 *    of operations as two 16-rank sweeps, so the factor is bit-identical to what the 16-rank form produces.
 * ------------------------------------------------------------------------------------------- */
template <int RPT, int K>
#ifndef QP_NI_UPDOWN
#define QP_NI_UPDOWN QPNI
#endif
/* a real function (own register allocation, see qpalm_device.h): plain pointer arguments, re-typed inside.
 * Ranks [rk0, rk0 + rkn) of the list "entering rows, then leaving rows" are applied, K per sweep. */
QP_NI_UPDOWN void dense_updown(const int *Atp_, const int *Ati_, const double *Atss_, const int n_, const int ld_,
                               double *L_, double *Dg_, double *Wst_, const int *cols_, int n_up_,
                               const int *cols_dn_, int n_dn_, QpShared *S_, char *lds, int64_t *tdbg_, int pre_jmin_ = -1,
                               double *fs_ = nullptr, int rk0_ = 0, int rkn_ = 0x7fffffff) {
  /* arguments of a real function arrive in VGPRs; these are wave-uniform: back to SGPRs, so that the
   * loops they bound are scalar loops (not exec-mask loops) and v_readlane indices are scalars */
  const int n = QP_UNIFORM(n_), ld = QP_UNIFORM(ld_), n_up = QP_UNIFORM(n_up_), n_dn = QP_UNIFORM(n_dn_);
  const int rk0 = QP_UNIFORM(rk0_), rkn = QP_UNIFORM(rkn_);
  /* pre_jmin >= 0: ONE rank whose dense vector the caller has already written to Wst[0 .. n) (first nonzero at
   * pre_jmin), sign +1 if n_up == 1 else -1: the trailing update of a KKT row addition / deletion */
  const int pre_jmin = QP_UNIFORM(pre_jmin_);
  const int pivmode = QP_UNIFORM(S_->seq_ranks); /* QP_PIV_* (set by dev_updown behind a barrier) */
  /* fs != NULL: the LAST sweep also does the forward substitution L y = b of the solve that follows (same ascending
   * column order: an entry of L is used for the substitution right after its last rank has been applied, so the panel
   * is not streamed a second time for it).  In: b, out: y.  The sweep then starts at column 0. */
  qp_gdouble *fs = (qp_gdouble *)fs_;
  const bool fuse_any = QP_UNIFORM((int)(fs_ != nullptr)) != 0;
  int64_t QP_LDS_AS *tdbg = (int64_t QP_LDS_AS *)tdbg_; /* the timers live in the kernel's static LDS */
  const qp_gint *Atp = (const qp_gint *)Atp_, *Ati = (const qp_gint *)Ati_, *cols = (const qp_gint *)cols_, *cols_dn = (const qp_gint *)cols_dn_;
  const qp_gdouble *Atss = (const qp_gdouble *)Atss_;
  qp_gdouble *L = (qp_gdouble *)L_, *Dg = (qp_gdouble *)Dg_, *Wst = (qp_gdouble *)Wst_;
  QpShared &S = *S_; /* static LDS of the kernel, reached through a generic pointer (one small reduction) */
  static_assert(K == 8 || K == 16 || K == 32, "rank groups of one DPP row (16 lanes), eight ranks at a time");
  static_assert(K <= QPG_KWST, "the staging area holds QPG_KWST dense update vectors");
  typedef UpdownLds<RPT, K> UpdownLdsT;
  UpdownLdsT QP_LDS_AS &U = *QP_LDS_ARG(UpdownLdsT, lds);
  static_assert(sizeof(UpdownLds<RPT, K>) <= ((RPT == 1) ? QP_LDS_BUDGET : QPG_LDS_DEFAULT), "update scratch must fit the dynamic LDS (the 256-thread instance only ever runs RPT = 1)");
  constexpr int NB = UpdownCfg<K>::NB, G = UpdownCfg<K>::G, KG = UpdownCfg<K>::KG;
  constexpr int PROWS = QP_T * RPT; /* rows per pass */
  constexpr bool MP = (K > 16);     /* several passes (the forms with at most 16 ranks run with RPT = all rows of the factor / QP_T: one pass) */
  static_assert(PROWS % NB == 0, "passes start on block boundaries");
  /* wavefronts renamed so that "wavefront 0" below (panel wave, owner of the first rows) is the wavefront qp_place_panel_wave picked
   * for this workgroup (S.wave_rank, see there); rows are owned by the RENAMED thread id throughout the sweep */
  const int lane = threadIdx.x & 63, wid = QP_UNIFORM(S.wave_rank[threadIdx.x >> 6]), tid = wid * 64 + lane;
  const int nr_all = n_up + n_dn;
  const int rk1 = (rkn < nr_all - rk0) ? (rk0 + rkn) : nr_all; /* ranks [rk0, rk1) */
  qp_gdouble *dummy = Wst + (size_t)QPG_KWST * n;
  qp_gdouble *Tab = dummy + QPG_DUMMY + QPG_HSTASH; /* exported tables of a multi-pass sweep: [column][K][2] */
  const int npass = MP ? (n + PROWS - 1) / PROWS : 1;
  for (int r0 = rk0; r0 < rk1; r0 += K) {
    const int kk = (rk1 - r0 < K) ? (rk1 - r0) : K;
    const bool fuse = fuse_any && (r0 + K >= nr_all);
    __syncthreads();
    long long tq0 = QP_CLOCK();
    int jmin = n;
    if (pre_jmin < 0) {
      for (int e = tid; e < kk * n; e += QP_T) Wst[e] = 0.0;
      __syncthreads();
      for (int r = wid; r < kk; r += QP_NW) {
        const int g = r0 + r;
        const int t = (g < n_up) ? cols[g] : cols_dn[g - n_up];
        for (int k = Atp[t] + lane; k < Atp[t + 1]; k += 64) {
          const int i = Ati[k];
          Wst[(size_t)r * n + i] = Atss[k];
          jmin = (i < jmin) ? i : jmin;
        }
      }
    } else jmin = (pre_jmin < n) ? pre_jmin : n - 1;
    if (fuse) jmin = 0;
    jmin = QP_UNIFORM(block_imin(S, jmin)); /* same value in every lane: keep the block loops scalar */
    /* lane r of every 16-lane row of the panel wave carries alpha_r and 1/alpha_r of rank 16 g + r (QP_RECUR_DPP; else lane r, G = 1);
     * they run on through the passes */
    double alpha[G], ialpha[G], sg[G];
    const int rl = QP_RECUR_DPP ? (lane & 15) : lane;
#pragma unroll
    for (int g = 0; g < G; g++) {
      alpha[g] = 1.0; ialpha[g] = 1.0;
      sg[g] = (16 * g + rl < kk) ? ((r0 + 16 * g + rl < n_up) ? 1.0 : -1.0) : 0.0;
    }
    const int J0 = (jmin / NB) * NB;
    if (tid == 0) {
      const long long tq1 = QP_CLOCK(); tdbg[0] += tq1 - tq0; tq0 = tq1;
      tdbg[QPG_CNT_SWEEPS] += 1;
      tdbg[QPG_CNT_SWEEP_ENTRIES] += (long long)(n - J0) * (n - J0 - 1) / 2 + (n - J0); /* strict lower part of columns J0.. + pivots */
    }
    for (int pass = 0; pass < npass; pass++) {
    const int R0 = pass * PROWS;                          /* this pass: rows [R0, R1), its triangle = columns [Js, R1) */
    const int R1 = (!MP || n - R0 < PROWS) ? n : R0 + PROWS;
    const int Js = (J0 > R0) ? J0 : R0;
    if (Js >= R1) continue;                               /* the update vectors are zero on these rows and nothing earlier touches them */
    const bool last_pass = (R1 >= n);
    const int rlim = last_pass ? ld : R1;                 /* rows n..ld-1 (padding of the panel) ride along in the last pass */
    if (MP && npass > 1) __syncthreads();                 /* the previous pass's last table export / LDS buffers */
    double w[RPT][K];
#pragma unroll
    for (int rr = 0; rr < RPT; rr++) {
      const int i = R0 + tid * RPT + rr; /* adjacent rows per thread: finished rows retire whole wavefronts */
#pragma unroll
      for (int r = 0; r < K; r++) w[rr][r] = (i < R1 && r < kk) ? Wst[(size_t)r * n + i] : 0.0;
    }
    double acc[RPT]; /* fused forward substitution: b_i - sum over the finished columns of l_ij y_j */
#pragma unroll
    for (int rr = 0; rr < RPT; rr++) { const int i = R0 + tid * RPT + rr; acc[rr] = (fuse && i < R1) ? fs[i] : 0.0; }
    /* One block of a finished table (buffer `tb` of U.cwg, ys in U.ys[tb]) applied to this thread's rows, columns Jc .. Jc + NB - 1:
     * one column per iteration.  Branch-free body: rows that do not take part read/write a private dummy cell (column stride 0);
     * the column QD ahead is prefetched into a register queue; the (-w_j, -gamma) pairs come as 16-byte LDS broadcasts, four
     * ranks at a time.  Register budget <= 128 VGPRs so that two workgroups share a CU.  The RPT adjacent rows of a thread are
     * one access group: 16-byte loads/stores (RPT even), one address and one liveness per thread. */
    auto apply_table = [&](const int tb, const int Jc, const bool ok) QP_ALWAYS_INLINE {
      constexpr int QD = (K > 16) ? 2 * QP_TQD : QP_TQD; /* one row per thread, 8-byte accesses: twice the columns in flight */
      static_assert(NB % QD == 0, "queue depth divides the block");
      constexpr bool ODPP = QP_OWNER_DPP32 && (K > 16); /* pairs by lane-indexed reads + DPP broadcast instead of K broadcast reads per column */
      const qp_pair QP_LDS_AS *tabO = (const qp_pair QP_LDS_AS *)QP_LDS_VBASE(&QP_CWG(U, tb, 0)[lane & 15][0]);
      const int i0 = R0 + tid * RPT;
      qp_gdouble *rowp = ok ? (L + (size_t)Jc * ld + i0) : (dummy + tid * RPT);
      const size_t cstride = ok ? (size_t)ld : 0;
      double q[QD][RPT];
#pragma unroll
      for (int cc = 0; cc < QD; cc++) qp_load_rows_nt<RPT>(rowp + (size_t)cc * cstride, q[cc]);
      auto group = [&](const int c0) QP_ALWAYS_INLINE {
#pragma unroll
        for (int u = 0; u < QD; u++) { /* queue slot u = fixed registers (see the panel wave's loop) */
          int c1 = c0 + u;
          QP_OPAQUE(c1); /* addresses are recomputed from c1: no per-slot induction pointers (VGPR budget) */
          double l[RPT];
          const int cpre = (c1 + QD < NB) ? c1 + QD : NB - 1;
#pragma unroll
          for (int rr = 0; rr < RPT; rr++) l[rr] = q[u][rr];
          if constexpr (ODPP) {
            qp_pair cf[G];
#pragma unroll
            for (int g = 0; g < G; g++) cf[g] = tabO[c1 * K + 16 * g];
#pragma unroll
            for (int rr = 0; rr < RPT; rr++) {
              qp_apply_ranks_dpp<K, 0, 0, 8, false>(cf[0].x, cf[0].y, w[rr], l[rr]);
              if (KG > 8 && kk > 8) qp_apply_ranks_dpp<K, 0, 8, 16, false>(cf[0].x, cf[0].y, w[rr], l[rr]);
              if constexpr (G > 1) {
                if (kk > 16) qp_apply_ranks_dpp<K, 16, 0, 8, false>(cf[G - 1].x, cf[G - 1].y, w[rr], l[rr]);
                if (kk > 24) qp_apply_ranks_dpp<K, 16, 8, 16, false>(cf[G - 1].x, cf[G - 1].y, w[rr], l[rr]);
              }
            }
          } else {
#pragma unroll
          for (int rb = 0; rb < K; rb += 4) {
            if (rb >= kk) break; /* ranks >= kk: exact no-ops, skipped */
            double cf[4][2];
#pragma unroll
            for (int r = 0; r < 4; r++) {
              cf[r][0] = (rb + r < K) ? QP_CWG(U, tb, c1)[rb + r][0] : 0.0; cf[r][1] = (rb + r < K) ? QP_CWG(U, tb, c1)[rb + r][1] : 0.0;
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
              if (rb + r < K) {
#pragma unroll
                for (int rr = 0; rr < RPT; rr++) {
                  w[rr][rb + r] = QP_FMA(cf[r][0], l[rr], w[rr][rb + r]);
                  l[rr] = QP_FMA(cf[r][1], w[rr][rb + r], l[rr]);
                }
              }
            }
            QP_SCHED_BARRIER();
          }
          }
          qp_store_rows_nt<RPT>(rowp + (size_t)c1 * cstride, l);
          if (fuse) { /* column Jc + c1 is final for these rows: its term of the forward substitution */
            const double yv = U.ys[tb][c1];
#pragma unroll
            for (int rr = 0; rr < RPT; rr++) acc[rr] = QP_FMA(-l[rr], yv, acc[rr]);
          }
          QP_SCHED_BARRIER();
          qp_load_rows_nt<RPT>(rowp + (size_t)cpre * cstride, q[u]); /* refill after the slot is free */
          QP_SCHED_BARRIER();
        }
      };
      group(0); /* peeled, see the panel wave's loop */
#pragma unroll 1
      for (int c0 = QD; c0 < NB; c0 += QD) group(c0);
    };
    /* table block `tb` of U.cwg (columns Jc .. Jc + NB - 1, complete) to the export area: a later pass applies it to its rows */
    auto export_table = [&](const int tb, const int Jc, const int t0, const int nt) QP_ALWAYS_INLINE {
      const qp_pair QP_LDS_AS *src = (const qp_pair QP_LDS_AS *)&U.cwg[tb][0][0][0];
      for (int e = t0; e < NB * K; e += nt) {
        const qp_pair v = src[e];
        Tab[((size_t)Jc * K + e) * 2] = v.x; Tab[((size_t)Jc * K + e) * 2 + 1] = v.y;
      }
    };
    if constexpr (MP) if (R0 > J0) {
      /* ===== rectangle: the tables of the earlier passes' columns [J0, R0) on this pass's rows, every wavefront, no serial chain.
       * Block b + 1 of the export area travels through registers into U.cwg[(b + 1) & 1] (+ y of its columns into U.ys) while
       * block b is applied: its load latency runs under the FMAs. ========================================================== */
      const int nrb = (R0 - J0) / NB;
      const int i0 = R0 + tid * RPT;
      const bool ok = (i0 < rlim);
      constexpr int NE = (NB * K + QP_T - 1) / QP_T;
      qp_pair stg[NE];
      double ystg = 0.0;
      auto tab_load = [&](const int b) QP_ALWAYS_INLINE {
        const int Jc = J0 + b * NB;
#pragma unroll
        for (int q = 0; q < NE; q++) {
          const int e = tid + q * QP_T;
          if (e < NB * K) { stg[q].x = Tab[((size_t)Jc * K + e) * 2]; stg[q].y = Tab[((size_t)Jc * K + e) * 2 + 1]; }
        }
        if (fuse && tid < NB) ystg = fs[Jc + tid];
      };
      auto tab_store = [&](const int b) QP_ALWAYS_INLINE {
        qp_pair QP_LDS_AS *dst = (qp_pair QP_LDS_AS *)&U.cwg[b & 1][0][0][0];
#pragma unroll
        for (int q = 0; q < NE; q++) {
          const int e = tid + q * QP_T;
          if (e < NB * K) dst[e] = stg[q];
        }
        if (fuse && tid < NB) U.ys[b & 1][tid] = ystg;
      };
      tab_load(0);
      tab_store(0);
      __syncthreads();
      for (int b = 0; b < nrb; b++) {
        if (b + 1 < nrb) tab_load(b + 1);
        if (__ballot(ok ? 1 : 0) != 0ull) apply_table(b & 1, J0 + b * NB, ok); /* whole wavefronts: the DPP broadcasts need every lane of a row */
        if (b + 1 < nrb) tab_store(b + 1);
        __syncthreads();
      }
    }
    const int nblk = (R1 - Js + NB - 1) / NB;
    /* prologue: rows of block 0 to the hand-over buffer, diagonal block 0 to LDS */
    {
      const int jb0 = (R1 - Js < NB) ? (R1 - Js) : NB;
#pragma unroll
      for (int rr = 0; rr < RPT; rr++) {
        const int i = R0 + tid * RPT + rr;
        if (i >= Js && i < Js + jb0) { /* block 0 */
#pragma unroll
          for (int r = 0; r < K; r++) U.Wd[0][i - Js][r] = w[rr][r];
          U.Wd[0][i - Js][K] = acc[rr];
        }
      }
      for (int e = tid; e < jb0 * jb0; e += QP_T) {
        const int c1 = e / jb0, c = e % jb0;
        if (c > c1) U.Ld[0][c][c1] = L[(size_t)(Js + c1) * ld + (Js + c)];
      }
      if (tid < jb0) U.dd[0][tid] = Dg[Js + tid];
      if (wid == 0) {
#pragma unroll
        for (int rr = 0; rr < RPT; rr++)
#pragma unroll
          for (int r = 0; r < K; r++) U.stash[rr][r][lane] = w[rr][r];
#pragma unroll
        for (int rr = 0; rr < RPT; rr++) U.stash_acc[rr][lane] = acc[rr];
      }
    }
    if (tid == 0) { const long long tq1 = QP_CLOCK(); tdbg[0] += tq1 - tq0; tq0 = tq1; }
    __syncthreads();
    for (int s = 0; s < nblk; s++) {
      const int J = Js + s * NB, Jp = J - NB, Jn = J + NB;
      const int jb = (R1 - J < NB) ? (R1 - J) : NB;
      const int jbn = (R1 - Jn < NB) ? ((R1 - Jn > 0) ? (R1 - Jn) : 0) : NB; /* 0 when block s is the last one of the pass */
      const int cur = s & 1, prv = cur ^ 1;
      long long tpe = QP_CLOCK();
      constexpr bool OWNA = QP_OWN_A && QP_USQ && !QP_SQ_WB;
      const int Ja = (OWNA && s > 0) ? J : Jn; /* first row the owners apply table s-1 to */
      const bool own_live0 = (R0 + 64 * RPT - 1 >= Ja); /* wavefront 0 still owns rows the owners work on */
      auto owner_block = [&]() QP_ALWAYS_INLINE {
        /* ===== owners: table s-1 on the rows below block s, then the rows of block s+1 to the hand-over buffer ========= */
        const long long tt0 = QP_CLOCK();
        bool any = false;
#pragma unroll
        for (int rr = 0; rr < RPT; rr++) { const int i = R0 + tid * RPT + rr; any = any || (i >= Ja && i < R1); }
        constexpr bool WAVES = MP && QP_OWNER_DPP32; /* DPP form: whole wavefronts take part (idle lanes work on their dummy cell) */
        if (s > 0 && (WAVES ? (__ballot(any ? 1 : 0) != 0ull) : any)) {
          const int i0 = R0 + tid * RPT;
          apply_table(prv, Jp, i0 >= Ja && i0 < rlim);
        }
        /* rows of block s+1 to the hand-over buffer of the next phase */
#pragma unroll
        for (int rr = 0; rr < RPT; rr++) {
          const int i = R0 + tid * RPT + rr;
          if (i >= Jn && i < Jn + jbn) {
#pragma unroll
            for (int r = 0; r < K; r++) U.Wd[prv][i - Jn][r] = w[rr][r];
            U.Wd[prv][i - Jn][K] = acc[rr];
          }
        }
        if (wid == 0) {
#pragma unroll
          for (int rr = 0; rr < RPT; rr++)
#pragma unroll
            for (int r = 0; r < K; r++) U.stash[rr][r][lane] = w[rr][r];
#pragma unroll
          for (int rr = 0; rr < RPT; rr++) U.stash_acc[rr][lane] = acc[rr];
        }
        if (tid == QP_T - 64) tdbg[2] += QP_CLOCK() - tt0; /* the last wavefront's rows live longest */
      };
      if (wid == 0) {
        /* ===== panel wave ===================================================================== */
        const long long tp0 = QP_CLOCK();
        QP_SETPRIO(3); /* the serial chain of the sweep goes first on its SIMD */
        double wrow[K];
        double accp;
#pragma unroll
        for (int r = 0; r < K; r++) wrow[r] = (lane < jb) ? U.Wd[cur][lane][r] : 0.0;
        accp = (lane < jb) ? U.Wd[cur][lane][K] : 0.0; /* this row's substitution accumulator (fused solve) */
        if (s > 0) {
          /* table s-1 applied to the rows of block s (lane = row): same loop as the trailing rows,
           * deeper queue (one row per lane: registers to spare, and this wave is the critical path).
           * The entries of L come from the square the owners staged in LDS (QP_USQ), else from HBM. */
          constexpr bool PSQ = QP_USQ;
          constexpr int QD = PSQ ? QP_AQD : 8;
          qp_gdouble *rowp = (lane < jb) ? (L + (size_t)Jp * ld + J + lane) : (dummy + lane);
          const size_t cstride = (lane < jb) ? (size_t)ld : 0;
          const int lrow = lane & (NB - 1);
          double q[QD];
#pragma unroll
          for (int cc = 0; cc < QD; cc++) q[cc] = PSQ ? U.Lsq[cur][cc][lrow] : rowp[(size_t)cc * cstride];
          /* unrolled by the queue depth: slot u of the queue is a fixed register, so the load
           * issued QD columns ago is the only one waited for (rotating the queue through register
           * moves would make every column wait for the newest load).
           * (Tried in round 2 and measured slower on MI355X in the full kernel, although faster in isolation: register
           * rotation of the (-w_j, -gamma) pairs, and two columns at a time skewed by one rank.  The panel wave is bound by
           * the number of instructions it issues, ~6.5 clk each, not by the FMA dependences.) */
          /* QP_PANEL_A_DPP: the (-w_j, -gamma) pairs of a column are fetched with ONE lane-indexed 16-byte LDS read per rank group (lane
           * & 15 = rank: every 16-lane row holds the group's pairs), QD columns ahead like the entries of L, and rank r's pair reaches
           * the row lanes by DPP row broadcast as in the recurrence: K broadcast reads and their waits per column leave the chain. */
          const qp_pair QP_LDS_AS *tabA = (const qp_pair QP_LDS_AS *)QP_LDS_VBASE(&QP_CWG(U, prv, 0)[lane & 15][0]);
          qp_pair cfq[QP_PANEL_A_DPP ? QD : 1][G];
          if (QP_PANEL_A_DPP) {
#pragma unroll
            for (int cc = 0; cc < QD; cc++)
#pragma unroll
              for (int g = 0; g < G; g++) cfq[cc][g] = tabA[cc * K + 16 * g];
          }
          auto group = [&](const int c0) QP_ALWAYS_INLINE {
#pragma unroll
            for (int u = 0; u < QD; u++) {
              const int c1 = c0 + u;
              double l = q[u];
              const int cpre = (c1 + QD < NB) ? c1 + QD : NB - 1;
              if (QP_PANEL_A_DPP) {
                qp_apply_ranks_dpp<K, 0, 0, 8, false>(cfq[u][0].x, cfq[u][0].y, wrow, l); /* (ranks >= kk carry zero pairs: exact no-ops) */
                if (KG > 8 && kk > 8) qp_apply_ranks_dpp<K, 0, 8, 16, false>(cfq[u][0].x, cfq[u][0].y, wrow, l);
                if constexpr (G > 1) {
                  if (kk > 16) qp_apply_ranks_dpp<K, 16, 0, 8, false>(cfq[u][G - 1].x, cfq[u][G - 1].y, wrow, l);
                  if (kk > 24) qp_apply_ranks_dpp<K, 16, 8, 16, false>(cfq[u][G - 1].x, cfq[u][G - 1].y, wrow, l);
                }
              } else {
#pragma unroll
              for (int rb = 0; rb < K; rb += 8) {
                if (rb >= kk) break; /* ranks >= kk are exact no-ops (w = 0, gamma = 0): skipped, wave-uniform */
                double cw[8], cg[8];
#pragma unroll
                for (int r = 0; r < 8; r++) { cw[r] = (rb + r < K) ? QP_CWG(U, prv, c1)[rb + r][0] : 0.0; cg[r] = (rb + r < K) ? QP_CWG(U, prv, c1)[rb + r][1] : 0.0; }
#pragma unroll
                for (int r = 0; r < 8; r++) {
                  if (rb + r < K) {
                    wrow[rb + r] = QP_FMA(cw[r], l, wrow[rb + r]);
                    l = QP_FMA(cg[r], wrow[rb + r], l);
                  }
                }
                if (K > 16) QP_SCHED_BARRIER(); /* 32 ranks: one group of eight pairs in registers at a time */
              }
              }
              /* QP_SQ_WB: the final entry goes back into the staged square (the owners write the square to HBM while they stage the next
               * one), so that the serial chain issues no global-memory instruction at all: a global store of the panel wave queues up
               * behind the owners' streaming loads and stores in the CU's memory pipeline (round 3's knock-outs: the owners' HBM traffic,
               * not their FMAs or LDS reads, slowed this loop by a third). */
              if (PSQ && QP_SQ_WB) { if (lane < NB) U.Lsq[cur][c1][lrow] = l; }
              else if (!OWNA) rowp[(size_t)c1 * cstride] = l; /* (OWNA: the rows' owners write the entry) */
              if (fuse) accp = QP_FMA(-l, U.ys[prv][c1], accp); /* column Jp + c1 is final for this row */
              QP_SCHED_BARRIER();
              q[u] = PSQ ? U.Lsq[cur][cpre][lrow] : rowp[(size_t)cpre * cstride]; /* refill AFTER the slot's register is free: no queue rotation on the back edge */
              if (QP_PANEL_A_DPP) {
#pragma unroll
                for (int g = 0; g < G; g++) cfq[u][g] = tabA[cpre * K + 16 * g];
              }
              QP_SCHED_BARRIER();
            }
          };
          /* first group peeled: the loop is then entered with as many memory operations in flight as
           * on its back edge, so the s_waitcnt counts inside are the steady-state ones */
          group(0);
#pragma unroll 1
          for (int c0 = QD; c0 < NB; c0 += QD) group(c0);
        }
        if (QP_PANEL_TIMING == 1 && lane == 0) tdbg[8] += QP_CLOCK() - tp0;
        const long long tp1 = QP_CLOCK();
        double dreg = (lane < jb) ? U.dd[cur][lane] : 1.0;          /* lane c holds the pivot of column c */
        double lnext = (lane > 0 && lane < jb) ? U.Ld[cur][lane][0] : 0.0;
        /* Rolled on purpose: unrolled, the 32 lane masks and lane-derived LDS addresses become
         * long-lived values that spill under the 128-VGPR cap, and every column then waits on
         * ~5 dependent scratch loads (measured: 1830 clk/column vs ~900 for this form). */
        double QP_LDS_AS *const wt = QP_LDS_VBASE(&U.Wt[0]);
#pragma unroll 1
        for (int c1 = 0; c1 < jb; c1++) {
          const int ln = QP_FRESH_LANE(lane);
          const double lcur = lnext;
          lnext = (ln > c1 + 1 && ln < jb) ? U.Ld[cur][ln][c1 + 1] : 0.0; /* in flight during this column (column NB is padding) */
          /* QP_PANEL_TIMING == 2 (diagnostic build): ms_dbg[8..11] = the column's four links: pivot row through LDS to lane = rank /
           * rank scalars / table entry through LDS back to every lane / the 2 K FMAs (each stamp drains the LDS queue first) */
          long long tc0 = 0;
          if (QP_PANEL_TIMING == 2) { QP_DRAIN_LDS(); tc0 = QP_CLOCK(); }
          if (ln == c1) {
#pragma unroll
            for (int r = 0; r < K; r++) wt[r] = wrow[r];
          }
          QP_WAVE_SYNC();
          /* rank-indexed scalars: lane = rank within its group (lanes >= kk carry w = 0 => gamma = 0: exact no-ops).  The groups of
           * 16 ranks follow each other in the same lanes; group g starts from the pivot group g-1 leaves. */
          const int rk = QP_RECUR_DPP ? (ln & 15) : ln;
          double d0 = qp_readlane(dreg, c1);
          double nwv[G], ngam[G];
#pragma unroll
          for (int g = 0; g < G; g++) {
            const int kkg = (kk - 16 * g < KG) ? (kk - 16 * g) : KG; /* ranks of this group in this sweep (wave-uniform; <= 0: none) */
            nwv[g] = 0.0; ngam[g] = 0.0;
            if (g > 0 && kkg <= 0) continue;
            const double wv = (rk < kkg) ? wt[(16 * g + rk) & (K - 1)] : 0.0;
            if (QP_PANEL_TIMING == 2 && g == 0) { QP_DRAIN_LDS(); const long long t = QP_CLOCK(); if (lane == 0) tdbg[8] += t - tc0; tc0 = t; }
            const double p = sg[g] * wv * wv * ialpha[g];
            double dnew, dprev;
            if (qp_rank_pivots<KG>(p, ln, d0, pivmode, dnew, dprev) && lane == 0) tdbg[QPG_CNT_SEQ_COLS] += 1; /* (rare: the guard re-summed this column) */
            const double rdn = qp_rcp(dnew), rdp = qp_rcp(dprev);
            const double gam = -sg[g] * wv * ialpha[g] * rdn;
            if (ln < KG) { QP_CWG(U, cur, c1)[16 * g + ln][0] = -wv; QP_CWG(U, cur, c1)[16 * g + ln][1] = -gam; } /* stored negated: plain FMAs below */
            /* a rank whose vector is zero in this column leaves its alpha alone: d_new / d_prev is exactly 1 there, d * rcp(d) is not.
             * (Columns above a rank's first nonzero are then exact no-ops: the result does not depend on where a sweep starts or on
             * how the ranks are grouped into sweeps.) */
            alpha[g] = (wv == 0.0) ? alpha[g] : alpha[g] * dnew * rdp;
            ialpha[g] = (wv == 0.0) ? ialpha[g] : ialpha[g] * dprev * rdn;
            d0 = qp_readlane(dnew, kkg - 1); /* pivot of the column after this group's last rank */
            nwv[g] = -wv; ngam[g] = -gam;
          }
          if (QP_PANEL_TIMING == 2) { double gg = ngam[0]; QP_OPAQUE_V(gg); const long long t = QP_CLOCK(); if (lane == 0) tdbg[9] += t - tc0; tc0 = t; }
          if (ln == c1) dreg = d0; /* final pivot of the column = value after the last rank */
          if (!QP_RECUR_DPP) QP_WAVE_SYNC();
          /* rows of the block: lane = row.  Rows <= c1 are finished, their registers may be
           * overwritten freely, so no selects: w_r -= w_j l ; l -= gamma w_r  (2 FMAs per rank) */
          if (QP_RECUR_DPP) {
            double l = lcur;
            qp_apply_ranks_dpp<K, 0, 0, 8>(nwv[0], ngam[0], wrow, l); /* (ranks >= kk carry zero pairs: exact no-ops) */
            if (KG > 8 && kk > 8) qp_apply_ranks_dpp<K, 0, 8, 16, false>(nwv[0], ngam[0], wrow, l);
            if constexpr (G > 1) {
              if (kk > 16) qp_apply_ranks_dpp<K, 16, 0, 8>(nwv[1], ngam[1], wrow, l);
              if (kk > 24) qp_apply_ranks_dpp<K, 16, 8, 16, false>(nwv[1], ngam[1], wrow, l);
            }
            if (ln > c1 && ln < jb) U.Ld[cur][ln][c1] = l;
            if (QP_PANEL_TIMING == 2) { double ll = l; QP_OPAQUE_V(ll); const long long t = QP_CLOCK(); if (lane == 0) tdbg[11] += t - tc0; }
          } else {
            static_assert(QP_RECUR_DPP || G == 1, "the LDS form of the recurrence handles one rank group");
            double l = lcur;
            /* the first eight entries come back from LDS together; with more than eight ranks an entry's registers are refilled with
             * the entry eight ranks further right after its two FMAs, so that the second group's LDS latency runs under the first
             * group's FMA chain (it used to be requested only after that chain: a second exposed round trip per column) */
            const qp_pair QP_LDS_AS *tab = (const qp_pair QP_LDS_AS *)&QP_CWG(U, cur, c1)[0][0];
            qp_pair cf[8];
#pragma unroll
            for (int r = 0; r < 8; r++) cf[r] = tab[r];
            if (QP_PANEL_TIMING == 2) { QP_DRAIN_LDS(); const long long t = QP_CLOCK(); if (lane == 0) tdbg[10] += t - tc0; tc0 = t; }
            if (K > 8 && kk > 8) {
#pragma unroll
              for (int r = 0; r < 8; r++) {
                wrow[r] = QP_FMA(cf[r].x, l, wrow[r]);
                l = QP_FMA(cf[r].y, wrow[r], l);
                cf[r] = tab[(8 + r < K) ? 8 + r : r];
                QP_SCHED_BARRIER();
              }
#pragma unroll
              for (int r = 0; r < 8; r++) {
                if (8 + r < K) {
                  wrow[8 + r] = QP_FMA(cf[r].x, l, wrow[8 + r]);
                  l = QP_FMA(cf[r].y, wrow[8 + r], l);
                }
              }
            } else {
#pragma unroll
              for (int r = 0; r < 8; r++) {
                wrow[r] = QP_FMA(cf[r].x, l, wrow[r]);
                l = QP_FMA(cf[r].y, wrow[r], l);
              }
            }
            if (ln > c1 && ln < jb) U.Ld[cur][ln][c1] = l;
            if (QP_PANEL_TIMING == 2) { double ll = l; QP_OPAQUE_V(ll); const long long t = QP_CLOCK(); if (lane == 0) tdbg[11] += t - tc0; }
          }
          QP_SCHED_BARRIER();
        }
        if (QP_PANEL_TIMING == 1 && lane == 0) tdbg[9] += QP_CLOCK() - tp1;
        const long long tp2 = QP_CLOCK();
        if (fuse) { /* the diagonal block is final: y_J = L_JJ^{-1} (b_J - contributions of the earlier blocks) */
          double v = accp;
          double lc = (lane > 0 && lane < jb) ? U.Ld[cur][lane][0] : 0.0;
#pragma unroll 1
          for (int c = 0; c < jb; c++) {
            const int ln = QP_FRESH_LANE(lane);
            const double lcn = (ln > c + 1 && ln < jb) ? U.Ld[cur][ln][c + 1] : 0.0; /* in flight during this step (column NB is padding) */
            const double yc = qp_readlane(v, c);
            v = QP_FMA(-lc, yc, v);
            lc = lcn;
          }
          if (lane < jb) { U.ys[cur][lane] = v; fs[J + lane] = v; }
        }
        /* diagonal block and pivots back to HBM (each lane re-reads what it wrote itself) */
        if (lane < jb) Dg[J + lane] = dreg;
        if (!QP_USQ) {
#pragma unroll 1
          for (int c = 0; c < jb; c++)
            if (lane > c && lane < jb) L[(size_t)(J + c) * ld + (J + lane)] = U.Ld[cur][lane][c];
        }
        /* back to being an ordinary owner: the running w of this wavefront's own rows */
#pragma unroll
        for (int rr = 0; rr < RPT; rr++)
#pragma unroll
          for (int r = 0; r < K; r++) w[rr][r] = own_live0 ? U.stash[rr][r][lane] : 0.0;
#pragma unroll
        for (int rr = 0; rr < RPT; rr++) acc[rr] = own_live0 ? U.stash_acc[rr][lane] : 0.0;
        QP_SETPRIO(0);
        if (lane == 0) { tdbg[1] += QP_CLOCK() - tp0; tdbg[QPG_CNT_SWEEP_COLS] += jb; }
        if (QP_PANEL_TIMING == 1 && lane == 0) tdbg[10] += QP_CLOCK() - tp2;
        tpe = QP_CLOCK();
      }
      if (wid != 0 || own_live0) owner_block();
      constexpr int NFREE = (QP_NW == 1) ? 0 : 1; /* wavefronts with a job of their own in this phase (the panel wave) */
      if (wid >= NFREE) { /* diagonal block s+1 for the next phase */
        const int t0 = tid - 64 * NFREE, nt = QP_T - 64 * NFREE;
        if (QP_USQ) {
          /* buffer [prv] still holds diagonal block s-1 as the panel wave left it in the last phase: back to HBM, then
           * block s+1 in its place (element by element, same thread); the square under block s goes to Lsq[prv] */
          for (int e = t0; e < NB * NB; e += nt) {
            const int c1 = e / NB, c = e % NB;
            if (c > c1) {
              if (s > 0) L[(size_t)(Jp + c1) * ld + (Jp + c)] = U.Ld[prv][c][c1];
              if (c < jbn) U.Ld[prv][c][c1] = L[(size_t)(Jn + c1) * ld + (Jn + c)];
            }
            /* the square of the LAST phase (rows of block s-1, columns of block s-2: block s-1 is a full block) as the panel wave left it: back to HBM */
            if (QP_SQ_WB && s >= 2) L[(size_t)(Jp - NB + c1) * ld + (Jp + c)] = U.Lsq[prv][c1][c];
            /* next phase: the panel wave applies table s to the rows of block s+1 */
            if (jbn > 0) U.Lsq[prv][c1][c] = (c < jbn) ? L[(size_t)(J + c1) * ld + (Jn + c)] : 0.0;
          }
        } else {
          for (int e = t0; e < jbn * jbn; e += nt) {
            const int c1 = e / jbn, c = e % jbn;
            if (c > c1) U.Ld[prv][c][c1] = L[(size_t)(Jn + c1) * ld + (Jn + c)];
          }
        }
        if (t0 < jbn) U.dd[prv][t0] = Dg[Jn + t0];
        if (MP && !last_pass && s > 0) export_table(prv, Jp, t0, nt); /* table s-1 is complete: the later passes' rows need it */
      }
      __syncthreads();
      if (QP_PANEL_TIMING == 1 && tid == 0) tdbg[11] += QP_CLOCK() - tpe;
    }
    if (QP_USQ) { /* the last diagonal block is still in LDS */
      const int sl = nblk - 1, Jl = Js + sl * NB, jbl = R1 - Jl;
      for (int e = tid; e < NB * NB; e += QP_T) {
        const int c1 = e / NB, c = e % NB;
        if (c > c1 && c < jbl) L[(size_t)(Jl + c1) * ld + (Jl + c)] = U.Ld[sl & 1][c][c1];
        /* ... and the square the panel wave updated in the last phase (rows of the last block, columns of the one before) */
        if (QP_SQ_WB && sl >= 1 && c < jbl) L[(size_t)(Jl - NB + c1) * ld + (Jl + c)] = U.Lsq[sl & 1][c1][c];
      }
    }
    if (MP && !last_pass) export_table((nblk - 1) & 1, Js + (nblk - 1) * NB, tid, QP_T); /* the pass's last table */
    } /* passes */
    if (tid == 0) { const long long tq1 = QP_CLOCK(); tdbg[7] += tq1 - tq0; tq0 = tq1; }
  }
  __syncthreads();
}


/* ---------------------------------------------------------------------------------------------
 * dense_updown_big: the same multi-rank update (same recurrence, same two FMAs per rank and entry) for factors with more
 * rows than RPT_max * QP_T = 2048, where the running w of a thread's rows no longer fit its registers (BASELINE.json
 * config 5: n = 5000).  The K running vectors stay in HBM (Wst, [K][n], read and written once per block column: K n 16
 * bytes per block against the 2 x 8 x 32 (n - J) of the panel itself) and every thread walks its rows in chunks of QP_T.
 * No look-ahead: per block column  [panel wave: recurrence on the 32 x 32 diagonal block]  barrier  [all wavefronts:
 * table applied to the rows below]  barrier.  Throughput is secondary here (one QP of this size keeps a workgroup busy for
 * seconds anyway); the arithmetic per entry is identical to dense_updown's.
 * ------------------------------------------------------------------------------------------- */
template <int K>
struct UpdownBigLds {
  double Ld[QP_UNB][QP_UNB + 1];
  double cwg[QP_UNB][K][2];
  double Wt[K];
  double dd[QP_UNB];
};
/* the recurrence on one 32 x 32 diagonal block (in U.Ld / dreg, lane = row of the block) for kk <= K ranks: leaves the table
 * (-w_j, -gamma) of the block's columns in U.cwg, the new entries in U.Ld and the new pivots in dreg; lane r carries alpha_r */
/* QP_PANEL_DPP = 1: the rank scalars are computed in every 16-lane row of the wavefront (lane & 15 = rank) and the (-w, -gamma)
 * pair of rank r reaches the row lanes by DPP row broadcast instead of through the LDS table (which is still written, for the
 * rows below the block, but not waited for): one LDS round trip per column instead of two.  Measured in DESIGN section 7. */
#ifndef QP_PANEL_DPP
#define QP_PANEL_DPP 1
#endif
template <int K>
QPD void updown_big_panel(UpdownBigLds<K> QP_LDS_AS &U, const int lane, const int jb, const int kk, const double sg,
                          double (&wrow)[K], double &dreg, double &alpha, double &ialpha, const int pivmode = QP_PIV_TREE) {
  static_assert(K == 16 || !QP_PANEL_DPP, "the DPP form needs one rank per lane of a 16-lane row");
  double lnext = (lane > 0 && lane < jb) ? U.Ld[lane][0] : 0.0;
#pragma unroll 1
  for (int c1 = 0; c1 < jb; c1++) {
    const int ln = QP_FRESH_LANE(lane);
    const double lcur = lnext;
    lnext = (ln > c1 + 1 && ln < jb) ? U.Ld[ln][c1 + 1] : 0.0; /* in flight during this column (column NB is padding) */
    if (ln == c1) {
#pragma unroll
      for (int r = 0; r < K; r++) U.Wt[r] = wrow[r];
    }
    QP_WAVE_SYNC();
    const int rk = QP_PANEL_DPP ? (ln & (K - 1)) : ln;
    const double wv = (rk < kk) ? U.Wt[rk & (K - 1)] : 0.0;
    const double d0 = qp_readlane(dreg, c1);
    const double p = sg * wv * wv * ialpha;
    double dnew, dprev;
    qp_rank_pivots<K>(p, ln, d0, pivmode, dnew, dprev);
    const double rdn = qp_rcp(dnew), rdp = qp_rcp(dprev);
    const double gam = -sg * wv * ialpha * rdn;
    if (ln < K) { U.cwg[c1][ln][0] = -wv; U.cwg[c1][ln][1] = -gam; }
    alpha = (wv == 0.0) ? alpha : alpha * dnew * rdp; /* zero in this column: exactly unchanged (see dense_updown) */
    ialpha = (wv == 0.0) ? ialpha : ialpha * dprev * rdn;
    { const double dfin = qp_readlane(dnew, kk - 1); if (ln == c1) dreg = dfin; }
    if (QP_PANEL_DPP) {
      double l = lcur;
      qp_apply_ranks_dpp<K, 0, 0, 8>(-wv, -gam, wrow, l); /* (ranks >= kk carry zero pairs: exact no-ops) */
      if (K > 8 && kk > 8) qp_apply_ranks_dpp<K, 0, 8, 16>(-wv, -gam, wrow, l);
      if (ln > c1 && ln < jb) U.Ld[ln][c1] = l;
    } else {
      QP_WAVE_SYNC();
      double l = lcur;
#pragma unroll
      for (int r = 0; r < K; r++) {
        if (r >= kk) break;
        wrow[r] = QP_FMA(U.cwg[c1][r][0], l, wrow[r]);
        l = QP_FMA(U.cwg[c1][r][1], wrow[r], l);
      }
      if (ln > c1 && ln < jb) U.Ld[ln][c1] = l;
    }
    QP_SCHED_BARRIER();
  }
}

template <int K>
QPNI void dense_updown_big(const int *Atp_, const int *Ati_, const double *Atss_, const int n_, const int ld_,
                           double *L_, double *Dg_, double *Wst_, const int *cols_, int n_up_,
                           const int *cols_dn_, int n_dn_, QpShared *S_, char *lds, int64_t *tdbg_, int pre_jmin_ = -1) {
  const int n = QP_UNIFORM(n_), ld = QP_UNIFORM(ld_), n_up = QP_UNIFORM(n_up_), n_dn = QP_UNIFORM(n_dn_), pre_jmin = QP_UNIFORM(pre_jmin_);
  int64_t QP_LDS_AS *tdbg = (int64_t QP_LDS_AS *)tdbg_;
  const qp_gint *Atp = (const qp_gint *)Atp_, *Ati = (const qp_gint *)Ati_, *cols = (const qp_gint *)cols_, *cols_dn = (const qp_gint *)cols_dn_;
  const qp_gdouble *Atss = (const qp_gdouble *)Atss_;
  qp_gdouble *L = (qp_gdouble *)L_, *Dg = (qp_gdouble *)Dg_, *Wst = (qp_gdouble *)Wst_;
  QpShared &S = *S_;
  typedef UpdownBigLds<K> LdsT;
  LdsT QP_LDS_AS &U = *QP_LDS_ARG(LdsT, lds);
  const int NB = QP_UNB;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int nr = n_up + n_dn;
  for (int r0 = 0; r0 < nr; r0 += K) {
    const int kk = (nr - r0 < K) ? (nr - r0) : K;
    __syncthreads();
    long long tq0 = QP_CLOCK();
    int jmin = n;
    if (pre_jmin < 0) {
      for (int e = tid; e < kk * n; e += QP_T) Wst[e] = 0.0;
      __syncthreads();
      for (int r = wid; r < kk; r += QP_NW) {
        const int g = r0 + r;
        const int t = (g < n_up) ? cols[g] : cols_dn[g - n_up];
        for (int k = Atp[t] + lane; k < Atp[t + 1]; k += 64) {
          const int i = Ati[k];
          Wst[(size_t)r * n + i] = Atss[k];
          jmin = (i < jmin) ? i : jmin;
        }
      }
    } else jmin = (pre_jmin < n) ? pre_jmin : n - 1;
    jmin = QP_UNIFORM(block_imin(S, jmin));
    double alpha = 1.0, ialpha = 1.0; /* lane r of wavefront 0 carries alpha_r and 1/alpha_r */
    const int rl = QP_PANEL_DPP ? (lane & (K - 1)) : lane, grank = r0 + rl; /* QP_PANEL_DPP: every 16-lane row of the panel wave carries the K ranks */
    const double sg = (rl < kk) ? ((grank < n_up) ? 1.0 : -1.0) : 0.0;
    const int J0 = (jmin / NB) * NB;
    if (tid == 0) {
      tdbg[QPG_CNT_SWEEPS] += 1;
      tdbg[QPG_CNT_SWEEP_ENTRIES] += (long long)(n - J0) * (n - J0 - 1) / 2 + (n - J0);
    }
    for (int J = J0; J < n; J += NB) {
      const int jb = (n - J < NB) ? (n - J) : NB;
      __syncthreads();
      for (int e = tid; e < jb * jb; e += QP_T) { /* diagonal block to LDS */
        const int c1 = e / jb, c = e % jb;
        if (c > c1) U.Ld[c][c1] = L[(size_t)(J + c1) * ld + (J + c)];
      }
      if (tid < jb) U.dd[tid] = Dg[J + tid];
      __syncthreads();
      if (wid == 0) { /* panel wave: lane = row of the block, the recurrence of dense_updown */
        double wrow[K];
#pragma unroll
        for (int r = 0; r < K; r++) wrow[r] = (lane < jb && r < kk) ? Wst[(size_t)r * n + J + lane] : 0.0;
        double dreg = (lane < jb) ? U.dd[lane] : 1.0;
        updown_big_panel<K>(U, lane, jb, kk, sg, wrow, dreg, alpha, ialpha, QP_UNIFORM(S.seq_ranks));
        if (lane < jb) Dg[J + lane] = dreg;
#pragma unroll 1
        for (int c = 0; c < jb; c++)
          if (lane > c && lane < jb) L[(size_t)(J + c) * ld + (J + lane)] = U.Ld[lane][c];
      }
      __syncthreads();
      /* rows below the block: one row per thread and chunk, its K running values from / to HBM */
      for (int i = J + jb + tid; i < n; i += QP_T) {
        double w[K];
#pragma unroll
        for (int r = 0; r < K; r++) w[r] = (r < kk) ? Wst[(size_t)r * n + i] : 0.0;
#pragma unroll 1
        for (int c1 = 0; c1 < jb; c1++) {
          double l = L[(size_t)(J + c1) * ld + i];
#pragma unroll
          for (int r = 0; r < K; r++) {
            if (r >= kk) break;
            w[r] = QP_FMA(U.cwg[c1][r][0], l, w[r]);
            l = QP_FMA(U.cwg[c1][r][1], w[r], l);
          }
          L[(size_t)(J + c1) * ld + i] = l;
        }
#pragma unroll
        for (int r = 0; r < K; r++) if (r < kk) Wst[(size_t)r * n + i] = w[r];
      }
    }
    if (tid == 0) { const long long tq1 = QP_CLOCK(); tdbg[7] += tq1 - tq0; tq0 = tq1; }
  }
  __syncthreads();
}


/* ---------------------------------------------------------------------------------------------
 * Coop mode (one large QP on many workgroups, qpalm_capi.inc: coop_linear_algebra): the multi-rank update of dense_updown_big
 * with ONE LAUNCH PER BLOCK COLUMN.  Every workgroup repeats the recurrence on the 32 x 32 diagonal block (a few microseconds)
 * from the same inputs and then applies the table to its share of the rows below; the kernel boundary orders block J + 1
 * after block J.  Workgroup 0 owns the block's outputs: the new diagonal block and pivots go to a stage (the other workgroups
 * may still be reading the old ones) and are written into L / D by the NEXT launch; the running alpha_r alternate between two
 * buffers.  State of one update, in the slot's stash area of Wst (hst = Wst + K n + QPG_DUMMY):
 *   hst[0..4K)  alpha, 1/alpha: buffer (J / 32) & 1 is read, the other written      hst[CO_UD_JMIN] first nonzero row
 *   hst[CO_UD_STAGED] block column held by the stage (-1 none)    hst[CO_UD_D ..) 32 pivots    hst[CO_UD_L ..) 32 x 32 entries
 * The arithmetic per entry is dense_updown_big's, so the factor is bit-identical to the one-workgroup sweep's.
 * ------------------------------------------------------------------------------------------- */
#define CO_UD_JMIN 100
#define CO_UD_STAGED 101
#define CO_UD_D 128
#define CO_UD_L 160
#define CO_UD_ROWS 128 /* rows of the panel per workgroup and pass */
#define CO_UD_FIRST ((QP_T >= 256) ? 128 : 0) /* first thread of the row phase: wavefronts 2 and 3 where the workgroup has them (the emulator's has two) */
/* phase 0 (all workgroups): Wst <- 0; phase 1 (one workgroup): the kk sparse rows of sqrt(Sigma) A scattered into it, state initialised */
template <int K>
QPD void co_updown_init(const int *Atp, const int *Ati, const double *Atss, const int n, double *Wst, double *hst, const int *cols, const int n_up,
                        const int *cols_dn, const int r0, const int kk, const int phase, QpShared &S, const int wg, const int nwg) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (phase == 0) {
    for (size_t e = (size_t)wg * QP_T + tid; e < (size_t)kk * n; e += (size_t)nwg * QP_T) Wst[e] = 0.0;
    return;
  }
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
  if (tid < 4 * K) hst[tid] = 1.0;
  if (tid == 0) { hst[CO_UD_JMIN] = (double)((jmin < n) ? jmin : n - 1); hst[CO_UD_STAGED] = -1.0; }
}
/* block column J (J >= n: only the pending stage is written back) */
template <int K>
QPD void co_updown_block(const int n, const int ld, double *L, double *Dg, double *Wst, double *hst, const int J, const int r0, const int kk,
                         const int n_up, char *lds, const int wg, const int nwg, const int pivmode) {
  typedef UpdownBigLds<K> LdsT;
  LdsT QP_LDS_AS &U = *QP_LDS_ARG(LdsT, lds);
  const int NB = QP_UNB, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (J < n && J + NB <= (int)hst[CO_UD_JMIN]) return; /* the update vectors are zero above their first entry */
  if (wg == 0) { /* the previous block's outputs, staged by workgroup 0 of the previous launch */
    const int Js = (int)hst[CO_UD_STAGED];
    if (Js >= 0) {
      const int js = (n - Js < NB) ? (n - Js) : NB;
      for (int e = tid; e < js * js; e += QP_T) {
        const int c1 = e / js, c = e % js;
        if (c > c1) L[(size_t)(Js + c1) * ld + (Js + c)] = hst[CO_UD_L + c1 * NB + c];
      }
      if (tid < js) Dg[Js + tid] = hst[CO_UD_D + tid];
    }
    __syncthreads();
    if (tid == 0) hst[CO_UD_STAGED] = -1.0;
  }
  if (J >= n) return;
  const int jb = (n - J < NB) ? (n - J) : NB, par = (J / NB) & 1;
  for (int e = tid; e < jb * jb; e += QP_T) { /* diagonal block to LDS */
    const int c1 = e / jb, c = e % jb;
    if (c > c1) U.Ld[c][c1] = L[(size_t)(J + c1) * ld + (J + c)];
  }
  if (tid < jb) U.dd[tid] = Dg[J + tid];
  __syncthreads();
  if (wid == 0) {
    double wrow[K];
#pragma unroll
    for (int r = 0; r < K; r++) wrow[r] = (lane < jb && r < kk) ? Wst[(size_t)r * n + J + lane] : 0.0;
    double dreg = (lane < jb) ? U.dd[lane] : 1.0;
    const int rl = QP_PANEL_DPP ? (lane & (K - 1)) : lane;
    double alpha = (rl < K) ? hst[par * 2 * K + rl] : 1.0, ialpha = (rl < K) ? hst[par * 2 * K + K + rl] : 1.0;
    const int grank = r0 + rl;
    const double sg = (rl < kk) ? ((grank < n_up) ? 1.0 : -1.0) : 0.0;
    updown_big_panel<K>(U, lane, jb, kk, sg, wrow, dreg, alpha, ialpha, pivmode);
    if (wg == 0) {
      if (lane < K) { hst[(1 - par) * 2 * K + lane] = alpha; hst[(1 - par) * 2 * K + K + lane] = ialpha; }
      if (lane < jb) hst[CO_UD_D + lane] = dreg;
#pragma unroll 1
      for (int c = 0; c < jb; c++)
        if (lane > c && lane < jb) hst[CO_UD_L + c * NB + lane] = U.Ld[lane][c];
      if (lane == 0) hst[CO_UD_STAGED] = (double)J;
    }
  }
  /* rows below the block (full blocks only: the last, ragged block has none): one row per thread of wavefronts 2 and 3 (wavefront 0
   * is busy with the recurrence), its kk running values from / to HBM.  The first pass's 32 entries and running values are
   * requested BEFORE the barrier, so their latency runs under the recurrence; further passes (grids smaller than the row count)
   * load after it, 16 columns at a time. */
  const int rt = tid - CO_UD_FIRST;
  const bool rower = rt >= 0 && rt < CO_UD_ROWS;
  const int i0 = J + NB + wg * CO_UD_ROWS + rt;
  double w[K], l[NB];
  if (rower && i0 < n) {
#pragma unroll
    for (int r = 0; r < K; r++) w[r] = (r < kk) ? Wst[(size_t)r * n + i0] : 0.0;
#pragma unroll
    for (int c = 0; c < NB; c++) l[c] = L[(size_t)(J + c) * ld + i0];
  }
  __syncthreads();
  if (rower && i0 < n) {
#pragma unroll
    for (int c = 0; c < NB; c++) {
#pragma unroll
      for (int r = 0; r < K; r++) {
        if (r >= kk) break;
        w[r] = QP_FMA(U.cwg[c][r][0], l[c], w[r]);
        l[c] = QP_FMA(U.cwg[c][r][1], w[r], l[c]);
      }
    }
#pragma unroll
    for (int c = 0; c < NB; c++) L[(size_t)(J + c) * ld + i0] = l[c];
#pragma unroll
    for (int r = 0; r < K; r++) if (r < kk) Wst[(size_t)r * n + i0] = w[r];
  }
  if (rower)
    for (int i = i0 + nwg * CO_UD_ROWS; i < n; i += nwg * CO_UD_ROWS) {
#pragma unroll
      for (int r = 0; r < K; r++) w[r] = (r < kk) ? Wst[(size_t)r * n + i] : 0.0;
#pragma unroll 1
      for (int h = 0; h < NB; h += 16) {
#pragma unroll
        for (int c = 0; c < 16; c++) l[c] = L[(size_t)(J + h + c) * ld + i];
#pragma unroll
        for (int c = 0; c < 16; c++) {
#pragma unroll
          for (int r = 0; r < K; r++) {
            if (r >= kk) break;
            w[r] = QP_FMA(U.cwg[h + c][r][0], l[c], w[r]);
            l[c] = QP_FMA(U.cwg[h + c][r][1], w[r], l[c]);
          }
        }
#pragma unroll
        for (int c = 0; c < 16; c++) L[(size_t)(J + h + c) * ld + i] = l[c];
      }
#pragma unroll
      for (int r = 0; r < K; r++) if (r < kk) Wst[(size_t)r * n + i] = w[r];
    }
}

/* --------------------------------------------------------------------------------------------
 * The same update as ONE launch (coop_updates = 2; VERDICT round 4, item 4: "a sweep one persistent launch").  A chain of
 * co_updown_block launches pays a kernel boundary per 32 columns (157 at n = 5000: 4.9 ms for a sweep that moves 200 MB) and every
 * workgroup repeats the block's recurrence.  Here a workgroup OWNS 128 rows of the panel for the whole sweep -- one row per thread of
 * two wavefronts, the row's kk running values in registers from the first block to the last instead of through HBM at every block --
 * and the four diagonal blocks inside those rows: it applies the tables of the blocks in front of its rows as their owners publish
 * them, then runs the recurrence of its own blocks (updown_big_panel, once per block in the whole grid), publishes each table
 * ((-w, -gamma) per column and rank + the running alpha: 8.4 KB) and a counter, and applies it to its remaining rows.  The only
 * data that crosses workgroups are the tables: L, D and the running vectors of a row are touched by its owner alone.
 * Hand-off: QP_RELEASE_AGENT / QP_ACQUIRE_AGENT above; one counter (flags[0] = blocks published so far, set to the first block
 * by the init launch), polled by one lane with a bounded spin (flags[1] != 0: gave up -- the host reports it, nothing hangs); the
 * grid is one workgroup per 128 rows (at most 64: all resident), workgroup g waits only for workgroups < g.
 * The arithmetic per entry is co_updown_block's: bit-identical factors.
 * ------------------------------------------------------------------------------------------- */
#define CO_UD_TAB(K) (QP_UNB * (K) * 2 + 2 * (K)) /* doubles per published table: the block's (-w, -gamma) pairs, then alpha and 1 / alpha per rank */
template <int K>
QPD void co_updown_persist(const int n, const int ld, double *L, double *Dg, const double *Wst, const double *hst, double *tab, int *flags,
                           const int r0, const int kk, const int n_up, char *lds, const int wg, QpShared &S, const int pivmode) {
  typedef UpdownBigLds<K> LdsT;
  LdsT QP_LDS_AS &U = *QP_LDS_ARG(LdsT, lds);
  double QP_LDS_AS *wd = (double QP_LDS_AS *)(QP_LDS_ARG(char, lds) + ((sizeof(LdsT) + 15) & ~(size_t)15)); /* [NB][K]: running values of a diagonal block's rows */
  double QP_LDS_AS *cw = (double QP_LDS_AS *)&U.cwg[0][0][0];
  const int NB = QP_UNB, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int R0 = wg * CO_UD_ROWS, R1 = (R0 + CO_UD_ROWS < n) ? R0 + CO_UD_ROWS : n;
  if (R0 >= n) return;
  const int bstart = (int)hst[CO_UD_JMIN] / NB; /* the update vectors are zero above their first entry: blocks in front of it are not touched */
  const int bown0 = R0 / NB, bown1 = (R1 + NB - 1) / NB; /* my diagonal blocks */
  if (bown1 <= bstart) return;
  if (wg == 1 && QP_UNIFORM(QP_FLAG_LOAD(flags + 2)) != 0) { /* TEST HOOK (context option coop_test_kill): this workgroup gives up as if it had timed out */
    __syncthreads();
    if (tid == 0) { QP_FLAG_STORE(flags + 2, 0); QP_FLAG_STORE(flags + 1, 1 + wg); }
    return;
  }
  const int rt = tid - CO_UD_FIRST;
  const int i = R0 + rt;
  const bool live = rt >= 0 && rt < CO_UD_ROWS && i < n;
  double w[K], l[NB];
#pragma unroll
  for (int r = 0; r < K; r++) w[r] = (live && r < kk) ? Wst[(size_t)r * n + i] : 0.0;
  auto apply_rows = [&](const int J) QP_ALWAYS_INLINE { /* table in U.cwg, the row's 32 entries of block column J in l */
#pragma unroll
    for (int c = 0; c < NB; c++) {
#pragma unroll
      for (int r = 0; r < K; r++) {
        if (r >= kk) break;
        w[r] = QP_FMA(U.cwg[c][r][0], l[c], w[r]);
        l[c] = QP_FMA(U.cwg[c][r][1], w[r], l[c]);
      }
    }
#pragma unroll
    for (int c = 0; c < NB; c++) L[(size_t)(J + c) * ld + i] = l[c];
  };
  /* ---- the blocks in front of my rows: their tables, as they are published ---- */
  int next = bstart;
  while (next < bown0) {
    if (tid == 0) {
      /* bounded wait for the owner of block `next`; flags[1] != 0 = somebody in this sweep has given up: leave at once instead of waiting out a
       * timeout of one's own behind it (the give-ups used to cascade).  The factor is then half updated: the iteration kernel sees
       * the mark when it resumes and rebuilds the factor (dev_solve), nothing iterates on it. */
      int r = QP_FLAG_LOAD(flags), dead = QP_FLAG_LOAD(flags + 1);
      for (unsigned spins = 0; r <= next && dead == 0 && spins < (1u << 22); spins++) { QP_SLEEP(); r = QP_FLAG_LOAD(flags); dead = QP_FLAG_LOAD(flags + 1); }
      if (r <= next) { if (dead == 0) QP_FLAG_STORE(flags + 1, 1 + wg); r = -1; }
      QP_ACQUIRE_AGENT();
      QP_DRAIN_VMEM();
      S.ired[0][3] = r;
    }
    __syncthreads();
    int upto = S.ired[0][3];
    if (upto < 0) return; /* a producer never arrived (flags[1] says who gave up) */
    upto = (upto < bown0) ? upto : bown0;
    for (int b = next; b < upto; b++) {
      const int J = b * NB;
      const double *T = tab + (size_t)b * CO_UD_TAB(K);
      if (live) {
#pragma unroll
        for (int c = 0; c < NB; c++) l[c] = L[(size_t)(J + c) * ld + i];
      }
      for (int e = tid; e < NB * K * 2; e += QP_T) cw[e] = T[e];
      __syncthreads();
      if (live) apply_rows(J);
      __syncthreads();
    }
    next = upto;
  }
  /* ---- my own blocks ---- */
  double alpha = 1.0, ialpha = 1.0;
  bool have_alpha = false;
  for (int b = (bown0 > bstart) ? bown0 : bstart; b < bown1; b++) {
    const int J = b * NB, jb = (n - J < NB) ? (n - J) : NB;
    for (int e = tid; e < jb * jb; e += QP_T) {
      const int c1 = e / jb, c = e % jb;
      if (c > c1) U.Ld[c][c1] = L[(size_t)(J + c1) * ld + (J + c)];
    }
    if (tid < jb) U.dd[tid] = Dg[J + tid];
    if (live && i >= J && i < J + jb) {
#pragma unroll
      for (int r = 0; r < K; r++) wd[(i - J) * K + r] = w[r];
    }
    const bool below = live && i >= J + jb; /* rows of my chunk under the block (full blocks only) */
    if (below) {
#pragma unroll
      for (int c = 0; c < NB; c++) l[c] = L[(size_t)(J + c) * ld + i]; /* in flight under the recurrence */
    }
    __syncthreads();
    if (wid == 0) {
      double wrow[K];
#pragma unroll
      for (int r = 0; r < K; r++) wrow[r] = (lane < jb && r < kk) ? wd[lane * K + r] : 0.0;
      double dreg = (lane < jb) ? U.dd[lane] : 1.0;
      const int rl = QP_PANEL_DPP ? (lane & (K - 1)) : lane;
      if (!have_alpha) { /* the carry of the block in front of mine arrived with its table */
        if (b > bstart) {
          const double *T = tab + (size_t)(b - 1) * CO_UD_TAB(K) + NB * K * 2;
          alpha = (rl < K) ? T[rl] : 1.0; ialpha = (rl < K) ? T[K + rl] : 1.0;
        }
        have_alpha = true;
      }
      const int grank = r0 + rl;
      const double sg = (rl < kk) ? ((grank < n_up) ? 1.0 : -1.0) : 0.0;
      updown_big_panel<K>(U, lane, jb, kk, sg, wrow, dreg, alpha, ialpha, pivmode);
      QP_WAVE_SYNC();
      if (lane < jb) Dg[J + lane] = dreg;
#pragma unroll 1
      for (int c = 0; c < jb; c++)
        if (lane > c && lane < jb) L[(size_t)(J + c) * ld + (J + lane)] = U.Ld[lane][c];
      if (R1 < n) { /* somebody owns rows under mine: publish */
        double *T = tab + (size_t)b * CO_UD_TAB(K);
        for (int e = lane; e < NB * K * 2; e += 64) T[e] = cw[e];
        if (lane < K) { T[NB * K * 2 + lane] = alpha; T[NB * K * 2 + K + lane] = ialpha; }
        QP_DRAIN_VMEM();
        if (lane == 0) {
          QP_RELEASE_AGENT();
          QP_DRAIN_VMEM();
          QP_FLAG_STORE(flags, b + 1);
        }
      }
    }
    __syncthreads();
    if (below) apply_rows(J);
    __syncthreads();
  }
}

#endif
