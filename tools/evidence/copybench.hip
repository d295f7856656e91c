// HBM copy yardstick variants (which form reaches the ~6.3 TB/s the guide quotes): hipcc --offload-arch=gfx950 -O3 tools/evidence/copybench.hip -o /tmp/cb && /tmp/cb
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
template <int U, bool NT> __global__ __launch_bounds__(256) void k_copy(const d2* __restrict__ s, d2* __restrict__ d, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    d2 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = NT ? __builtin_nontemporal_load(s + i + u * stride) : s[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; u++) { if (NT) __builtin_nontemporal_store(v[u], d + i + u * stride); else d[i + u * stride] = v[u]; }
  }
  for (; i < n; i += stride) d[i] = s[i];
}
// contiguous chunk per workgroup (each workgroup streams its own 128 KB pieces)
template <int U> __global__ __launch_bounds__(256) void k_copy_chunk(const d2* __restrict__ s, d2* __restrict__ d, size_t n) {
  const size_t per = (size_t)U * 256;
  for (size_t base = (size_t)blockIdx.x * per; base < n; base += (size_t)gridDim.x * per) {
    d2 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) { size_t i = base + u * 256 + threadIdx.x; v[u] = (i < n) ? s[i] : d2{0, 0}; }
#pragma unroll
    for (int u = 0; u < U; u++) { size_t i = base + u * 256 + threadIdx.x; if (i < n) d[i] = v[u]; }
  }
}
__global__ __launch_bounds__(256) void k_read(const d2* __restrict__ s, double* out, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x; double a = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) { d2 v = s[i]; a += v.x + v.y; }
  if (a == 1.2345) out[0] = a;
}
int main() {
  const size_t bytes = (size_t)1 << 30, n = bytes / 16;
  d2 *s, *d; double* o; hipMalloc(&s, bytes); hipMalloc(&d, bytes); hipMalloc(&o, 8); hipMemset(s, 1, bytes); hipMemset(d, 0, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
#define RUN(name, grid, mult, ...) { float best = 1e30f; for (int r = 0; r < 6; r++) { hipEventRecord(e0); hipLaunchKernelGGL(__VA_ARGS__, dim3(grid), dim3(256), 0, 0, s, d, n); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (r && ms < best) best = ms; } printf("%-28s grid %6d : %7.1f GB/s\n", name, grid, mult * bytes / (best * 1e-3) / 1e9); }
  for (int g : {1024, 2048, 4096, 8192, 16384}) {
    RUN("copy U1", g, 2.0, (k_copy<1, false>)) RUN("copy U4", g, 2.0, (k_copy<4, false>)) RUN("copy U4 nt", g, 2.0, (k_copy<4, true>)) RUN("copy U8", g, 2.0, (k_copy<8, false>))
    RUN("copy chunk U8", g, 2.0, (k_copy_chunk<8>)) RUN("copy chunk U16", g, 2.0, (k_copy_chunk<16>))
  }
  { float best = 1e30f; for (int r = 0; r < 6; r++) { hipEventRecord(e0); hipLaunchKernelGGL(k_read, dim3(4096), dim3(256), 0, 0, s, o, n); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (r && ms < best) best = ms; } printf("read only                    : %7.1f GB/s\n", bytes / (best * 1e-3) / 1e9); }
  hipMemcpyDtoD(d, s, bytes); hipDeviceSynchronize();
  { float best = 1e30f; for (int r = 0; r < 4; r++) { hipEventRecord(e0); hipMemcpyDtoDAsync(d, s, bytes, 0); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; } printf("hipMemcpyDtoD                : %7.1f GB/s\n", 2.0 * bytes / (best * 1e-3) / 1e9); }
  return 0;
}
