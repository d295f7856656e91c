// Round 6: what an fp64 instruction costs a wavefront on gfx950 -- plain FMA, the DPP form of the update sweep (v_fmac_f64_dpp row_newbcast), the same with half
// of the lanes switched off, FMA with a scalar operand, and v_mfma_f64_16x16x4_f64 -- dependent and independent, one wavefront alone and four per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/evidence/mb_f64rate.hip -o /tmp/mbr && /tmp/mbr
// The question behind it: would the table application of the sweep be cheaper on the matrix cores?  Per row and block it is 2 K 32 = 1024
// FMAs as a recurrence (flop-minimal: a K-state linear system) against (32 + K)^2 = 2304 MACs as a GEMM; that pays only if an MFMA MAC is > 2.25 x cheaper.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 2048
#define STAMP0 long long c0 = clock64();
#define STAMP1(ops) long long c1 = clock64(); if (threadIdx.x == 0 && blockIdx.x == 0) out[1024] = (double)(c1 - c0) / (double)(ops);
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void k_fma_dep(double *out, double a, double b, int half) {
  double v = out[threadIdx.x];
  if (half && (threadIdx.x & 63) >= 32) return;
  STAMP0
  for (int i = 0; i < N; i++) {
    asm volatile("v_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2" : "+v"(v) : "v"(a), "v"(b));
  }
  STAMP1(4 * N) out[threadIdx.x] = v;
}
__global__ void k_fma_indep(double *out, double a, double b) {
  double v0 = out[threadIdx.x], v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3;
  STAMP0
  for (int i = 0; i < N; i++) {
    asm volatile("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(a), "v"(b));
  }
  STAMP1(4 * N) out[threadIdx.x] = v0 + v1 + v2 + v3;
}
__global__ void k_fma_sgpr_dep(double *out, double a, double b) { /* multiplier from an SGPR pair */
  double v = out[threadIdx.x];
  double sa = __builtin_bit_cast(double, (long long)__builtin_amdgcn_readfirstlane((int)__double2loint(a)) | ((long long)__builtin_amdgcn_readfirstlane(__double2hiint(a)) << 32));
  STAMP0
  for (int i = 0; i < N; i++) {
    asm volatile("v_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2" : "+v"(v) : "s"(sa), "v"(b));
  }
  STAMP1(4 * N) out[threadIdx.x] = v;
}
/* the sweep's pair of instructions: w += c0(bcast) * l ; l += c1(bcast) * w  (two dependent DPP FMAs per rank) */
#define DPP2(n) "v_fmac_f64_dpp %0, %2, %1 row_newbcast:" #n " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f64_dpp %1, %3, %0 row_newbcast:" #n " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
__global__ void k_dpp_dep(double *out, double a, double b, int half) {
  double w = out[threadIdx.x], l = w + 1.0;
  if (half && (threadIdx.x & 63) >= 32) return;
  STAMP0
  for (int i = 0; i < N; i++) {
    asm volatile("s_nop 1\n\t" DPP2(0) DPP2(1) DPP2(2) DPP2(3) : "+&v"(w), "+&v"(l) : "v"(a), "v"(b));
  }
  STAMP1(8 * N) out[threadIdx.x] = w + l;
}
#define DPP1(d, n) "v_fmac_f64_dpp %" #d ", %8, %9 row_newbcast:" #n " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
__global__ void k_dpp_indep(double *out, double a, double b) {
  double v0 = out[threadIdx.x], v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7;
  STAMP0
  for (int i = 0; i < N; i++) {
    asm volatile("s_nop 1\n\t" DPP1(0, 0) DPP1(1, 1) DPP1(2, 2) DPP1(3, 3) DPP1(4, 4) DPP1(5, 5) DPP1(6, 6) DPP1(7, 7)
                 : "+&v"(v0), "+&v"(v1), "+&v"(v2), "+&v"(v3), "+&v"(v4), "+&v"(v5), "+&v"(v6), "+&v"(v7) : "v"(a), "v"(b));
  }
  STAMP1(8 * N) out[threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
}
/* the halved chain: per rank  t = 0; t += gamma(bcast) w [off the chain]; t += c(bcast) l [the chain]; w += -wj(bcast) l [off the chain] */
#define HC(n) "v_mov_b64 %2, 0\n\tv_fmac_f64_dpp %2, %4, %0 row_newbcast:" #n " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
              "v_fmac_f64_dpp %0, %3, %1 row_newbcast:" #n " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
              "v_fmac_f64_dpp %2, %5, %1 row_newbcast:" #n " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_mov_b64 %1, %2\n\t"
__global__ void k_halfchain(double *out, double a, double b) {
  double w = out[threadIdx.x], l = w + 1.0, t = 0.0;
  STAMP0
  for (int i = 0; i < N; i++) {
    asm volatile("s_nop 1\n\t" HC(0) HC(1) HC(2) HC(3) : "+&v"(w), "+&v"(l), "+&v"(t) : "v"(a), "v"(b), "v"(a));
  }
  STAMP1(4 * N) out[threadIdx.x] = w + l + t;   /* per RANK (the two-FMA form above: per instruction, two per rank) */
}
__global__ void k_mfma_dep(double *out, double a, double b) {
  d4 acc = {0, 0, 0, 0};
  double x = out[threadIdx.x] + a, y = b;
  STAMP0
  for (int i = 0; i < N; i++) {
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc, 0, 0, 0);
  }
  STAMP1(2 * N) out[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
__global__ void k_mfma_indep(double *out, double a, double b) {
  d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  double x = out[threadIdx.x] + a, y = b;
  STAMP0
  for (int i = 0; i < N; i++) {
    a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
  }
  STAMP1(4 * N) out[threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}
/* v_permlane32_swap on a dependent chain (the hand-over of QP_SPLIT_A) */
typedef unsigned u2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double lower_to_upper(double keep, double src) {
  const u2v a = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(keep), (unsigned)__double2loint(src), false, false);
  const u2v b = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(keep), (unsigned)__double2hiint(src), false, false);
  return __hiloint2double((int)b[0], (int)a[0]);
}
__global__ void k_swap_dep(double *out, double a, double b) {
  double v = out[threadIdx.x], k = v + 1.0;
  STAMP0
  for (int i = 0; i < N; i++) {
    v = lower_to_upper(k, v); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v) : "v"(a), "v"(b));
    v = lower_to_upper(k, v); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v) : "v"(a), "v"(b));
  }
  STAMP1(2 * N) out[threadIdx.x] = v + k;   /* per (two swaps + one FMA) */
}
/* one step of the split loop: two swaps, sixteen DPP FMAs -- against thirty-two DPP FMAs (the unsplit column) */
#define DPP8 DPP2(0) DPP2(1) DPP2(2) DPP2(3) DPP2(4) DPP2(5) DPP2(6) DPP2(7)
__global__ void k_split_step(double *out, double a, double b) {
  double w = out[threadIdx.x], l = w + 1.0, k = w + 2.0;
  STAMP0
  for (int i = 0; i < N; i++) {
    l = lower_to_upper(k, l);
    asm volatile("s_nop 1\n\t" DPP8 : "+&v"(w), "+&v"(l) : "v"(a), "v"(b));
  }
  STAMP1(N) out[threadIdx.x] = w + l;
}
__global__ void k_unsplit_step(double *out, double a, double b) {
  double w = out[threadIdx.x], l = w + 1.0;
  STAMP0
  for (int i = 0; i < N; i++) {
    asm volatile("s_nop 1\n\t" DPP8 "s_nop 1\n\t" DPP8 : "+&v"(w), "+&v"(l) : "v"(a), "v"(b));
  }
  STAMP1(N) out[threadIdx.x] = w + l;
}
int main() {
  double *d; hipMalloc(&d, 8192 * 8); hipMemset(d, 0, 8192 * 8);
  double r;
#define RUN(label, kern, threads, ...) hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, __VA_ARGS__); hipDeviceSynchronize(); hipMemcpy(&r, d + 1024, 8, hipMemcpyDeviceToHost); \
  printf("%-58s %4d threads  %7.2f clk per op\n", label, threads, r);
  for (int threads : {64, 256, 1024}) {   /* one wavefront alone; one per SIMD; four per SIMD (the sweep's occupancy) */
    RUN("v_fma_f64, dependent", k_fma_dep, threads, d, 1.0000001, 1e-9, 0)
    RUN("v_fma_f64, dependent, lanes 32..63 off", k_fma_dep, threads, d, 1.0000001, 1e-9, 1)
    RUN("v_fma_f64, 4 independent", k_fma_indep, threads, d, 1.0000001, 1e-9)
    RUN("v_fma_f64 with an SGPR multiplier, dependent", k_fma_sgpr_dep, threads, d, 1.0000001, 1e-9)
    RUN("v_fmac_f64_dpp pair (the sweep's two per rank), dependent", k_dpp_dep, threads, d, 1e-9, 1e-9, 0)
    RUN("v_fmac_f64_dpp pair, dependent, lanes 32..63 off", k_dpp_dep, threads, d, 1e-9, 1e-9, 1)
    RUN("v_fmac_f64_dpp, 8 independent", k_dpp_indep, threads, d, 1e-9, 1e-9)
    RUN("halved chain: 5 instructions per RANK, 1 on the chain", k_halfchain, threads, d, 1e-9, 1e-9)
    RUN("2 x v_permlane32_swap + 1 FMA, dependent", k_swap_dep, threads, d, 1e-9, 1e-9)
    RUN("split step: 2 swaps + mov + 16 DPP FMAs", k_split_step, threads, d, 1e-9, 1e-9)
    RUN("unsplit step: 32 DPP FMAs", k_unsplit_step, threads, d, 1e-9, 1e-9)
    RUN("v_mfma_f64_16x16x4_f64, dependent (1024 MACs each)", k_mfma_dep, threads, d, 1e-9, 1e-9)
    RUN("v_mfma_f64_16x16x4_f64, 4 independent", k_mfma_indep, threads, d, 1e-9, 1e-9)
  }
  return 0;
}
