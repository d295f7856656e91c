/*
 * qpalm_device.h -- gfx950 device code of the QPALM inner loop (one workgroup == one QP).
 *
 * Execution model: a persistent workgroup of QP_T threads (QP_T/64 wavefronts, 64 lanes each) owns
 * one QP at a time and runs the whole semismooth-Newton loop of qpalm_solve (src/qpalm.c:484-711)
 * for it: residuals + termination reductions, active-set compaction, LDL^T factor / rank-k
 * update / triangular solves on a dense column-major panel in HBM, CSC SpMVs, the exact line search
 * (LDS bitonic sort + scan) and the primal update.  Independent QPs never communicate, so there is
 * no inter-workgroup synchronisation at all; control flow is workgroup-uniform.
 *
 * Arithmetic policy: the element-wise "host arithmetic" of the reference (lin_alg.c loops) is
 * reproduced operation by operation WITHOUT fused multiply-add (the file is compiled with
 * -ffp-contract=off) so that those results are bit-identical to a CPU build of the reference;
 * explicit fma() is used only inside SpMV dots and the dense LDL^T kernels.
 *
 * The same source is compiled for the host against tests/emu/hip_emu.h (QPALM_EMU) to debug the
 * logic without a GPU; that build is test-only.
 */
#ifndef QPALM_DEVICE_H
#define QPALM_DEVICE_H

#include "qpalm_types.h"

#ifndef QP_T
#define QP_T 512
#endif
#define QP_NW (QP_T / 64)
/* ranks per update sweep by rows-per-thread: the running vectors w[RPT][K] must fit the 128-VGPR budget; the 256-thread
 * instance (small QPs, four workgroups per CU) always uses 8 (its 40 KB LDS share bounds the coefficient tables) */
#ifndef QP_KSEL
#define QP_KSEL(RPT) ((RPT) <= 2 ? 16 : 8)
#endif
/* which instances of k_solve<RPT> also carry the 32-rank multi-pass sweep dense_updown<1, 32> (passes of QP_T rows): the 512-thread
 * instance up to 1024 rows (RPT 1 and 2: one or two passes) */
#ifndef QP_K32
#ifdef QPALM_EMU
#define QP_K32(RPT) ((RPT) >= 1 && (RPT) <= 2) /* the test build (any block size) exercises it too */
#define QP_LDS_BUDGET QPG_LDS_DEFAULT
#else
#define QP_K32(RPT) (QP_T >= 512 && (RPT) >= 1 && (RPT) <= 2)
#define QP_LDS_BUDGET ((QP_T >= 512) ? QPG_LDS_DEFAULT : 38912)
#endif
#endif
#define QPD __device__ __forceinline__
#define QPN __device__ __forceinline__
/* Which phases of the iteration are real calls (own register allocation, fewer values live across them in the loop body) instead
 * of inlined into dev_solve: 0 none, 2 (default) the outer-iteration phases (sigma / gamma updates, store_solution, dual objective),
 * 1 also the line search and the active sets.  Same box, round 3: spilled VGPRs of qp512::k_solve<2> 199 / 146 / 123; headline
 * 5042-5050 / 5039-5044 / 4817-4820 QP/s (the line search as a call loses its address arithmetic: 6.0 -> 7.5 ms per QP);
 * mpc-160 391 / 401 / 398 k QP/s. */
#ifndef QP_PHASE_CALLS
#define QP_PHASE_CALLS 2
#endif
#if QP_PHASE_CALLS && !defined(QPALM_EMU)
#define QPP __device__ __noinline__
#else
#define QPP __device__ __forceinline__
#endif
#if QP_PHASE_CALLS == 1 && !defined(QPALM_EMU) /* 1: the per-iteration phases (line search, active sets) too; 2: only the outer-iteration phases */
#define QPPH __device__ __noinline__
#else
#define QPPH __device__ __forceinline__
#endif
/* a real call: the callee gets its own register allocation instead of inheriting the live values of
 * the whole iteration loop.  Its pointer arguments are generic, so the callee re-types them (HBM arrays
 * are global memory, the LDS block is LDS): flat accesses would tie vmcnt to lgkmcnt. */
#ifdef QPALM_EMU
#define QPNI __device__ inline
typedef double qp_gdouble;
typedef int qp_gint;
typedef char qp_gchar;
#define QP_LDS_ARG(T, p) ((T *)(p))
#define QP_LDS_AS
#else
#define QPNI __device__ __noinline__
typedef double __attribute__((address_space(1))) qp_gdouble;
typedef int __attribute__((address_space(1))) qp_gint;
typedef char __attribute__((address_space(1))) qp_gchar;
#define QP_LDS_AS __attribute__((address_space(3)))
/* the LDS block handed to a real function: typed as LDS, its (wave-uniform) offset back in an SGPR and
 * known to be 16-byte aligned, so that accesses are ds_*_b128 with scalar base */
static __device__ __forceinline__ unsigned qp_lds_arg_offset_(char *p) {
  const unsigned a = (unsigned)__builtin_amdgcn_readfirstlane((int)(size_t)(char QP_LDS_AS *)p);
  __builtin_assume((a & 15u) == 0);
  return a;
}
#define QP_LDS_ARG(T, p) ((T QP_LDS_AS *)(size_t)qp_lds_arg_offset_((char *)(p)))
#endif

#ifdef QPALM_EMU
typedef emu_double4 qp_double4;
#define QP_MFMA_F64(a, b, c) emu_mfma_f64_16x16x4((a), (b), (c))
#define QP_DYN_LDS() (emu::dyn_lds())
#define QP_FMA(a, b, c) std::fma((a), (b), (c))
#define QP_SQRT(x) std::sqrt(x)
#define QP_CLOCK() ((long long)wall_clock64())
#define QP_UNIFORM(x) (x)
#define QP_UNIFORM_PTR(p) (p)
#define QP_OPAQUE(x) do { } while (0)
#define QP_OPAQUE_V(x) do { } while (0)
#define QP_FRESH_LANE(lane) (lane)
#define QP_CALL_BLOCK() 1
#define QP_ALWAYS_INLINE
/* emulation: one "CU", wavefront w sits on "SIMD" w & 3 (exercises the panel-wave rotation) */
#define QP_HW_CU_KEY() 0
#define QP_HW_SIMD() ((int)(threadIdx.x >> 6) & 3)
#else
typedef double qp_double4 __attribute__((ext_vector_type(4)));
#define QP_MFMA_F64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
extern __shared__ __attribute__((aligned(16))) char qp_dyn_lds_[];
/* The dynamic LDS block as a pointer the optimiser cannot trace back to the symbol.  Otherwise
 * inter-procedural constant propagation rewrites the `lds` argument of the real functions into the
 * symbol itself, and a non-kernel function reaches dynamic LDS through a table in memory
 * (llvm.amdgcn.dynlds.offset.table): an s_load + wait inside the serial loops. */
static __device__ __forceinline__ char *qp_dyn_lds_opaque_() {
  unsigned a = (unsigned)(size_t)(char __attribute__((address_space(3))) *)qp_dyn_lds_;
  asm volatile("" : "+s"(a));
  return (char *)(char __attribute__((address_space(3))) *)(size_t)a;
}
#define QP_DYN_LDS() (qp_dyn_lds_opaque_())
#define QP_FMA(a, b, c) fma((a), (b), (c))
#define QP_SQRT(x) sqrt(x)
#define QP_CLOCK() ((long long)wall_clock64())
#define QP_UNIFORM(x) __builtin_amdgcn_readfirstlane(x) /* value is wave-uniform: keep it in an SGPR */
template <class P> static __device__ __forceinline__ P qp_uniform_ptr_(P p) { /* a wave-uniform pointer (function arguments arrive in VGPRs) back in an SGPR pair */
  const unsigned long long a = (unsigned long long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
  return (P)(((unsigned long long)hi << 32) | lo);
}
#define QP_UNIFORM_PTR(p) qp_uniform_ptr_(p)
/* stops LICM/CSE from keeping ~100 per-array addresses live across the whole iteration loop */
/* (readfirstlane first: after a branch the compiler could not prove uniform the value may sit in a VGPR phi, and a
 * plain "+s" constraint is then an illegal VGPR-to-SGPR copy; on an SGPR value the readfirstlane folds away) */
#define QP_OPAQUE(x) do { (x) = __builtin_amdgcn_readfirstlane(x); asm volatile("" : "+s"(x)); } while (0)
#define QP_OPAQUE_V(x) asm volatile("" : "+v"(x)) /* same for a value that lives in a VGPR (function arguments) */
/* the lane id recomputed on the spot (2 VALU ops): inside latency-critical loops this keeps lane-derived
 * LDS addresses and lane masks out of long-lived (= spilled, under the 128-VGPR cap) registers */
static __device__ __forceinline__ int qp_fresh_lane_() {
  int z = 0;
  asm volatile("s_mov_b32 %0, 0" : "=s"(z));
  return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, (unsigned)z));
}
#define QP_FRESH_LANE(lane) qp_fresh_lane_()
/* A wave-uniform "true" the compiler cannot see through: `if (QP_CALL_BLOCK()) callee(...)` gives the call a basic block
 * of its own behind a scalar branch.  Why: ROCm 7.2's register allocator saves caller-saved VGPRs around a call with
 * copies at the top of the block that holds the call; when that block starts with the `s_or_b64 exec` closing a divergent
 * loop, the copies land AHEAD of it and run with the loop's lanes still switched off (tools/evidence/scan_exec_prologue.py finds
 * them in the -save-temps assembly; __graft_entry__.build() runs it over the shipped library). */
static __device__ __forceinline__ int qp_opaque_true_() {
  int one;
  asm volatile("s_mov_b32 %0, 1" : "=s"(one));
  return one;
}
#define QP_CALL_BLOCK() qp_opaque_true_()
#define QP_ALWAYS_INLINE __attribute__((always_inline))
/* where this wavefront runs: HW_REG_HW_ID (id 4: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13) and
 * HW_REG_XCC_ID (id 20, bits 3:0) -- s_getreg_b32 with (size-1) << 11 | offset << 6 | id */
#define QP_HW_SIMD() ((int)__builtin_amdgcn_s_getreg((2 - 1) << 11 | 4 << 6 | 4))
#define QP_HW_CU_KEY() ((int)(__builtin_amdgcn_s_getreg((8 - 1) << 11 | 8 << 6 | 4) | (__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) << 8)))
#endif

/* ---- c_max / c_min / c_absval exactly as the reference's macros (include/global_opts.h) ---- */
QPD double qmax(double a, double b) { return (a > b) ? a : b; }
QPD double qmin(double a, double b) { return (a < b) ? a : b; }
QPD double qabs(double x) { return (x < 0) ? -x : x; }

/* =============================================================================================
 * workgroup primitives
 * =========================================================================================== */
#define QP_NRED 20
struct QpShared {          /* small static LDS block */
  double red[QP_NW][QP_NRED];
  double bc[QP_NRED];      /* broadcast scalars */
  int    ired[QP_NW][4];
  int    ibc[8];
  int    hw_simd[QP_NW];   /* SIMD each wavefront of this workgroup sits on */
  int    placement;        /* diagnostic code of the placement (QPGStats.placement) */
  int    panel_wave;       /* the wavefront that runs the serial chains of the update sweep (qp_place_panel_wave) */
  int    seq_ranks;        /* how the update sweeps sum the ranks' contributions to a pivot (QP_PIV_*, set by dev_updown from qp_pivot_mode) */
  int    wave_rank[QP_NW]; /* the sweep's name for each hardware wavefront: 0 = panel wave (owner of the first rows), then the wavefronts that
                              sit on the SIMDs where the CU's panel waves run (they get the rows that retire first), then the rest */
};

/* How the pivots of a column after each rank of a sweep are summed (qp_rank_pivots, qpalm_dense.h).  The reference (cholmod_updown,
 * solver_interface.c:415-421,433-439) carries the running pivot d_r = d_{r-1} + p_r rank after rank; a prefix tree d_0 + (p_0 + .. + p_r)
 * is equal in exact arithmetic, four dependent steps instead of fifteen, and as accurate UNLESS a pivot shrinks a lot inside the sweep
 * (then the tree's rounding error is relative to d_0, not to the pivot: DESIGN.md section 5).  One predicate for every sweep form. */
#define QP_PIV_TREE 0  /* prefix tree, unguarded (context option sequential_rank_sums = 0: A/B runs only) */
#define QP_PIV_SEQ 1   /* the running pivot in every column (= 1; and by itself for nonconvex QPs and QPs whose Q has a column without a positive diagonal) */
#define QP_PIV_GUARD 2 /* the default: prefix tree, and a column in which some pivot comes out below 2^-8 of the column's pivot before the sweep (or is not
                          finite) is summed again as running pivots -- wave-uniform, one compare per column */
QPD int qp_pivot_mode(const qpg_view &V, int b) {
  if (V.seq_mode > 0 || (V.seq_mode < 0 && (V.sc[b].nc_flag != 0 || V.sc[b].seq_hint != 0))) return QP_PIV_SEQ;
  return (V.seq_mode < 0) ? QP_PIV_GUARD : QP_PIV_TREE;
}

QPD double wave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
QPD double wave_max(double v) {
  for (int o = 32; o > 0; o >>= 1) { double u = __shfl_xor(v, o); v = (u > v) ? u : v; }
  return v;
}
QPD int wave_isum(int v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
QPD int wave_imax(int v) {
  for (int o = 32; o > 0; o >>= 1) { int u = __shfl_xor(v, o); v = (u > v) ? u : v; }
  return v;
}
QPD int wave_imin(int v) {
  for (int o = 32; o > 0; o >>= 1) { int u = __shfl_xor(v, o); v = (u < v) ? u : v; }
  return v;
}

/* Reduce NM maxima and NS sums over the workgroup; every thread receives the results.
 * Fixed tree (butterfly inside a wavefront, wavefronts combined in index order) => deterministic. */
template <int NM, int NS>
QPD void block_reduce(QpShared &S, double *vm, double *vs) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NM; k++) vm[k] = wave_max(vm[k]);
#pragma unroll
  for (int k = 0; k < NS; k++) vs[k] = wave_sum(vs[k]);
  __syncthreads(); /* S.red may still be read by a previous reduction's consumers */
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < NM; k++) S.red[wid][k] = vm[k];
#pragma unroll
    for (int k = 0; k < NS; k++) S.red[wid][NM + k] = vs[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NM; k++) {
    double v = S.red[0][k];
    for (int w = 1; w < QP_NW; w++) { double u = S.red[w][k]; v = (u > v) ? u : v; }
    vm[k] = v;
  }
#pragma unroll
  for (int k = 0; k < NS; k++) {
    double v = S.red[0][NM + k];
    for (int w = 1; w < QP_NW; w++) v += S.red[w][NM + k];
    vs[k] = v;
  }
}

QPD int block_isum(QpShared &S, int v) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  v = wave_isum(v);
  __syncthreads();
  if (lane == 0) S.ired[wid][0] = v;
  __syncthreads();
  int t = 0;
  for (int w = 0; w < QP_NW; w++) t += S.ired[w][0];
  return t;
}
QPD int block_imin(QpShared &S, int v) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  v = wave_imin(v);
  __syncthreads();
  if (lane == 0) S.ired[wid][1] = v;
  __syncthreads();
  int t = S.ired[0][1];
  for (int w = 1; w < QP_NW; w++) t = (S.ired[w][1] < t) ? S.ired[w][1] : t;
  return t;
}

/* Ordered stream compaction: out[] receives, ascending, every i in [0,count) with flag(i) != 0.
 * Two lists are produced in one pass (enter / leave of newton.c:134-149).  Returns the counts via
 * S.ibc[0..1] (valid for all threads after the call). */
template <class F0, class F1>
QPD void block_compact2(QpShared &S, int count, F0 f0, F1 f1, int *out0, int *out1, int &n0, int &n1) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int base0 = 0, base1 = 0;
  for (int i0 = 0; i0 < count; i0 += QP_T) {
    const int i = i0 + (int)threadIdx.x;
    const int p0 = (i < count) ? (f0(i) ? 1 : 0) : 0;
    const int p1 = (i < count) ? (f1(i) ? 1 : 0) : 0;
    const unsigned long long b0 = __ballot(p0), b1 = __ballot(p1);
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int pre0 = __popcll(b0 & below), pre1 = __popcll(b1 & below);
    __syncthreads();
    if (lane == 0) { S.ired[wid][2] = __popcll(b0); S.ired[wid][3] = __popcll(b1); }
    __syncthreads();
    int w0 = 0, w1 = 0, t0 = 0, t1 = 0;
    for (int w = 0; w < QP_NW; w++) {
      if (w < wid) { w0 += S.ired[w][2]; w1 += S.ired[w][3]; }
      t0 += S.ired[w][2]; t1 += S.ired[w][3];
    }
    if (p0) out0[base0 + w0 + pre0] = i;
    if (p1) out1[base1 + w1 + pre1] = i;
    base0 += t0; base1 += t1;
  }
  n0 = base0; n1 = base1;
  __syncthreads();
}

/* y[r] = sum_k val[k] * x[idx[k]], k in [ptr[r], ptr[r+1]) -- one sub-wavefront of G lanes per
 * compressed column/row, coalesced walk of idx/val, butterfly reduce.  Used for A'*yh (CSC of A),
 * A*d (CSC of A') and Q*d (full symmetric pattern).  post(r, sum) consumes the result. */
#ifndef QP_SPMV_ROWS
#define QP_SPMV_ROWS 4 /* rows in flight per lane group; 8 spills under the 128-VGPR cap (measured slower) */
#endif
template <int G, class XP, class Post>
QPD void spmv_rows(int nrows, const int *__restrict__ ptr, const int *__restrict__ idx,
                   const double *__restrict__ val, XP x, Post post) { /* XP: const double * in HBM, or the vector staged in LDS (spmv_stage_x) */
  /* Every dependent global load costs 0.5-1.5 us here, and a row is a chain of three (pointer ->
   * index/value -> x[index]).  So each group of G lanes walks U rows at once with branch-free,
   * clamped loads: the U chains are in flight together.  Per row the arithmetic is unchanged: lane
   * `sub` accumulates entries sub, sub+G, ... in order, then the xor tree. */
  constexpr int U = QP_SPMV_ROWS, RG = QP_T / G;
  const int sub = threadIdx.x & (G - 1), grp = threadIdx.x / G;
  for (int r0 = 0; r0 < nrows; r0 += U * RG) {
    int k[U], k1[U];
    double acc[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int r = r0 + grp + u * RG;
      const bool valid = r < nrows;
      k[u] = valid ? ptr[r] + sub : 0;
      k1[u] = valid ? ptr[r + 1] : 0;
      acc[u] = 0.0;
    }
    while (true) {
      bool more = false;
#pragma unroll
      for (int u = 0; u < U; u++) more = more || (k[u] < k1[u]);
      if (!more) break;
      double vv[U], xx[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int kk = (k[u] < k1[u]) ? k[u] : 0; /* clamped: entry 0 always exists when any row has work */
        vv[u] = val[kk];
        xx[u] = x[idx[kk]];
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (k[u] < k1[u]) acc[u] = QP_FMA(vv[u], xx[u], acc[u]);
        k[u] += G;
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      double a = acc[u];
#pragma unroll
      for (int o = G / 2; o > 0; o >>= 1) a += __shfl_xor(a, o);
      const int r = r0 + grp + u * RG;
      if (r < nrows && sub == 0) post(r, a);
    }
  }
}

/* The gathered vector of an SpMV staged in LDS: the third of a row's three dependent round trips (pointer -> index / value -> x[index])
 * becomes an LDS read.  Returns nullptr when the vector does not fit the dynamic LDS block (the caller then gathers from HBM).
 * The caller synchronises before the SpMV and before the block is used for anything else.  Same values, same arithmetic. */
typedef const double QP_LDS_AS *qp_lds_cdouble;
#ifndef QP_SPMV_LDS
#define QP_SPMV_LDS 1 /* 0: gather from HBM as in rounds 1-4 (A/B) */
#endif
QPD qp_lds_cdouble spmv_stage_x(const double *x, int nx, char *lds, int lds_bytes) {
  if (!QP_SPMV_LDS || (size_t)nx * sizeof(double) > (size_t)lds_bytes) return nullptr;
  double QP_LDS_AS *xs = (double QP_LDS_AS *)lds;
  for (int i = threadIdx.x; i < nx; i += QP_T) xs[i] = x[i];
  return xs;
}

/* per-QP pointer helpers */
struct QpPtr {
  const qpg_view &V; int b;
  QPD QpPtr(const qpg_view &v, int bb) : V(v), b(bb) {}
  QPD double *vn(double *base) const { return base + (size_t)b * V.n; }
  QPD double *vm(double *base) const { return base + (size_t)b * V.m; }
  QPD int *im(int *base) const { return base + (size_t)b * V.m; }
};

#include "qpalm_dense.h"
#include "qpalm_iter.h"

#endif
