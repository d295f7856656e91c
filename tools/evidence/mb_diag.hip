// Replica of one column step of dense_updown's block recurrence, with parts switchable, timed with clock64.
#include <hip/hip_runtime.h>
#include <cstdio>
#define K 16
#define NB 32
struct Lds { double Ld[NB][NB + 1]; double cwg[NB][K][2]; double Wt[K]; };
template <int N> __device__ __forceinline__ double row_shr(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x110 + N, 0xf, 0xf, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x110 + N, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rl(double v, int s) { return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), s), __builtin_amdgcn_readlane(__double2loint(v), s)); }
__device__ __forceinline__ double rcpnr(double x) { double r = __builtin_amdgcn_rcp(x); r = fma(fma(-x, r, 1.0), r, r); r = fma(fma(-x, r, 1.0), r, r); return r; }
#define WS() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
template <int F> __global__ void __launch_bounds__(64) k_diag(double* out, int reps, int kk) {
  __shared__ Lds U;
  const int lane = threadIdx.x;
  double wrow[K];
  for (int r = 0; r < K; r++) wrow[r] = 1e-3 * (lane + r + 1);
  for (int c = 0; c < NB; c++) { U.Ld[lane & 31][c] = 1e-2 * (c + 1); for (int r = 0; r < K; r++) { U.cwg[c][r][0] = 1e-3; U.cwg[c][r][1] = 1e-4; } }
  if (lane < K) U.Wt[lane] = 0.1;
  __syncthreads();
  double alpha = 1.0, ialpha = 1.0, dreg = 2.0 + lane, sg = lane < kk ? 1.0 : 0.0;
  long long c0 = clock64();
  for (int rep = 0; rep < reps; rep++) {
    double lnext = U.Ld[lane & 31][0];
#pragma unroll
    for (int c1 = 0; c1 < NB; c1++) {
      const double lcur = lnext;
      if (F & 16) { if (c1 + 1 < NB) lnext = (lane > c1 + 1 && lane < NB) ? U.Ld[lane][c1 + 1] : 0.0; }
      if (F & 1) { if (lane == c1) {
#pragma unroll
        for (int r = 0; r < K; r++) U.Wt[r] = wrow[r]; }
        WS(); }
      double wv = 0.1, gam = 1e-4;
      if (F & 1) wv = (lane < kk) ? U.Wt[lane] : 0.0;
      if (F & 2) {
        const double d0 = rl(dreg, c1);
        const double p = sg * wv * wv * ialpha;
        double incl = p;
        incl += row_shr<1>(incl); incl += row_shr<2>(incl); incl += row_shr<4>(incl); incl += row_shr<8>(incl);
        const double excl = row_shr<1>(incl);
        const double dnew = d0 + incl, dprev = d0 + excl;
        const double rdn = rcpnr(dnew), rdp = rcpnr(dprev);
        gam = -sg * wv * ialpha * rdn;
        alpha = alpha * dnew * rdp; ialpha = ialpha * dprev * rdn;
        const double dfin = rl(dnew, kk - 1);
        if (lane == c1) dreg = dfin;
      }
      if (F & 4) { if (lane < K) { U.cwg[c1][lane][0] = -wv; U.cwg[c1][lane][1] = -gam; } WS(); }
      double l = lcur;
      if (F & 8) {
        double cw[K], cg[K];
#pragma unroll
        for (int r = 0; r < K; r++) { cw[r] = U.cwg[c1][r][0]; cg[r] = U.cwg[c1][r][1]; }
#pragma unroll
        for (int r = 0; r < K; r++) { wrow[r] = fma(cw[r], l, wrow[r]); l = fma(cg[r], wrow[r], l); }
      }
      if (F & 16) { if (lane > c1 && lane < NB) U.Ld[lane][c1] = l; }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  long long c1_ = clock64();
  double acc = alpha + ialpha + dreg; for (int r = 0; r < K; r++) acc += wrow[r];
  out[lane] = acc;
  if (lane == 0) out[64] = (double)(c1_ - c0) / (reps * NB);
}
int main() { double* d; hipMalloc(&d, 1024); double r;
#define RUN(F) hipLaunchKernelGGL(k_diag<F>, dim3(1), dim3(64), 0, 0, d, 200, 16); hipDeviceSynchronize(); hipMemcpy(&r, d + 64, 8, hipMemcpyDeviceToHost); printf("flags %2d: %8.1f clk per column\n", F, r);
  RUN(31) RUN(1) RUN(2) RUN(4) RUN(8) RUN(16) RUN(3) RUN(12) RUN(15) RUN(0)
  return 0; }
