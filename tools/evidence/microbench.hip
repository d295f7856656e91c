// Latency microbenchmarks for gfx950 (single wavefront unless noted). hipcc --offload-arch=gfx950 -O3 tools/evidence/microbench.hip -o /tmp/mb && /tmp/mb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N 4096
__global__ void k_fma_chain(double* out, double a, double b) { double v = out[threadIdx.x]; long long t0 = wall_clock64(); long long c0 = clock64();
  for (int i = 0; i < N; i++) v = fma(v, a, b);
  long long c1 = clock64(); long long t1 = wall_clock64(); out[threadIdx.x] = v; if (threadIdx.x == 0) { out[64] = (double)(c1 - c0) / N; out[65] = (double)(t1 - t0) * 10.0 / N; } }
__global__ void k_fma_indep(double* out, double a, double b) { double v0 = out[threadIdx.x], v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3; long long c0 = clock64(); long long t0 = wall_clock64();
  for (int i = 0; i < N; i++) { v0 = fma(v0, a, b); v1 = fma(v1, a, b); v2 = fma(v2, a, b); v3 = fma(v3, a, b); }
  long long c1 = clock64(); long long t1 = wall_clock64(); out[threadIdx.x] = v0 + v1 + v2 + v3; if (threadIdx.x == 0) { out[64] = (double)(c1 - c0) / (4 * N); out[65] = (double)(t1 - t0) * 10.0 / (4 * N); } }
__global__ void k_lds_chain(double* out) { __shared__ int idx[1024]; for (int i = threadIdx.x; i < 1024; i += blockDim.x) idx[i] = (i * 17 + 5) & 1023; __syncthreads();
  int p = threadIdx.x; long long c0 = clock64(); long long t0 = wall_clock64();
  for (int i = 0; i < N; i++) p = idx[p];
  long long c1 = clock64(); long long t1 = wall_clock64(); out[threadIdx.x] = p; if (threadIdx.x == 0) { out[64] = (double)(c1 - c0) / N; out[65] = (double)(t1 - t0) * 10.0 / N; } }
__global__ void k_lds_wr_rd(double* out) { __shared__ double buf[64]; double v = out[threadIdx.x]; long long c0 = clock64(); long long t0 = wall_clock64();
  for (int i = 0; i < N; i++) { buf[threadIdx.x] = v; __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); v = buf[(threadIdx.x + 1) & 63] + 1.0; __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }
  long long c1 = clock64(); long long t1 = wall_clock64(); out[threadIdx.x] = v; if (threadIdx.x == 0) { out[64] = (double)(c1 - c0) / N; out[65] = (double)(t1 - t0) * 10.0 / N; } }
__global__ void k_gload_chain(double* out, const int* idx) { int p = threadIdx.x; long long c0 = clock64(); long long t0 = wall_clock64();
  for (int i = 0; i < 512; i++) p = idx[p];
  long long c1 = clock64(); long long t1 = wall_clock64(); out[threadIdx.x] = p; if (threadIdx.x == 0) { out[64] = (double)(c1 - c0) / 512; out[65] = (double)(t1 - t0) * 10.0 / 512; } }
__global__ void k_barrier(double* out) { long long c0 = clock64(); long long t0 = wall_clock64();
  for (int i = 0; i < N; i++) __syncthreads();
  long long c1 = clock64(); long long t1 = wall_clock64(); if (threadIdx.x == 0) { out[64] = (double)(c1 - c0) / N; out[65] = (double)(t1 - t0) * 10.0 / N; } }
__global__ void k_div_chain(double* out, double a) { double v = out[threadIdx.x] + 2.0; long long c0 = clock64(); long long t0 = wall_clock64();
  for (int i = 0; i < N; i++) v = a / v + 1.5;
  long long c1 = clock64(); long long t1 = wall_clock64(); out[threadIdx.x] = v; if (threadIdx.x == 0) { out[64] = (double)(c1 - c0) / N; out[65] = (double)(t1 - t0) * 10.0 / N; } }
__global__ void k_bperm_chain(double* out) { double v = out[threadIdx.x]; long long c0 = clock64(); long long t0 = wall_clock64();
  for (int i = 0; i < N; i++) v = __shfl(v, (threadIdx.x + 1) & 63) + 1.0;
  long long c1 = clock64(); long long t1 = wall_clock64(); out[threadIdx.x] = v; if (threadIdx.x == 0) { out[64] = (double)(c1 - c0) / N; out[65] = (double)(t1 - t0) * 10.0 / N; } }
__global__ void k_readlane_chain(double* out) { double v = out[threadIdx.x]; long long c0 = clock64(); long long t0 = wall_clock64();
  for (int i = 0; i < N; i++) { int lo = __builtin_amdgcn_readlane(__double2loint(v), 5), hi = __builtin_amdgcn_readlane(__double2hiint(v), 5); v = v + __hiloint2double(hi, lo); }
  long long c1 = clock64(); long long t1 = wall_clock64(); out[threadIdx.x] = v; if (threadIdx.x == 0) { out[64] = (double)(c1 - c0) / N; out[65] = (double)(t1 - t0) * 10.0 / N; } }
int main() { double* d; hipMalloc(&d, 4096 * 8); hipMemset(d, 0, 4096 * 8); int* idx; std::vector<int> h(1 << 20); for (int i = 0; i < (1 << 20); i++) h[i] = (int)(((long long)i * 40503 + 12345) & ((1 << 20) - 1)); hipMalloc(&idx, h.size() * 4); hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  double r[2];
#define RUN(name, threads, ...) hipLaunchKernelGGL(name, dim3(1), dim3(threads), 0, 0, __VA_ARGS__); hipDeviceSynchronize(); hipMemcpy(r, d + 64, 16, hipMemcpyDeviceToHost); printf("%-22s threads=%4d  %8.1f clk  %8.1f ns per op\n", #name, threads, r[0], r[1]);
  RUN(k_fma_chain, 64, d, 1.0000001, 1e-9) RUN(k_fma_indep, 64, d, 1.0000001, 1e-9) RUN(k_lds_chain, 64, d) RUN(k_lds_wr_rd, 64, d) RUN(k_gload_chain, 64, d, idx)
  RUN(k_barrier, 512, d) RUN(k_barrier, 64, d) RUN(k_div_chain, 64, d, 3.0) RUN(k_bperm_chain, 64, d) RUN(k_readlane_chain, 64, d)
  return 0; }
