// does a scalar load see a vector store of another wave of the same workgroup after {s_waitcnt vmcnt(0); barrier}?  (and with s_dcache_inv)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i16v __attribute__((ext_vector_type(16)));
__global__ void k(double *tab, int *bad, int rounds, int inv) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  double *t = tab + (size_t)blockIdx.x * 4096;
  int nbad = 0;
  for (int r = 1; r <= rounds; r++) {
    if (inv) { __builtin_amdgcn_s_dcache_inv(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    for (int blk = 0; blk < 8; blk++) {
      if (wid == 0) { // producer: 512 doubles of block blk
        for (int e = lane; e < 512; e += 64) t[blk * 512 + e] = (double)(r * 100000 + blk * 512 + e);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();
      if (wid != 0) {
        for (int c = 0; c < 512; c += 8) {
          i16v v;
          const double *p = t;
          unsigned long long a = (unsigned long long)(size_t)p;
          const double *ps = (const double *)(size_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)a));
          int off = (blk * 512 + c) * 8;
          asm volatile("s_load_dwordx16 %0, %1, %2\n s_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(ps), "s"(off) : "memory");
          for (int q = 0; q < 8; q++) {
            const double got = __hiloint2double(v[2 * q + 1], v[2 * q]);
            if (got != (double)(r * 100000 + blk * 512 + c + q)) nbad++;
          }
        }
      }
      __syncthreads();
    }
  }
  if (lane == 0 && nbad) atomicAdd(bad, nbad);
}
int main() {
  double *tab; int *bad, h;
  hipMalloc(&tab, 1024 * 4096 * 8); hipMalloc(&bad, 4);
  for (int inv = 0; inv < 2; inv++) {
    hipMemset(bad, 0, 4); hipMemset(tab, 0, 1024 * 4096 * 8);
    hipLaunchKernelGGL(k, dim3(512), dim3(512), 0, 0, tab, bad, 20, inv);
    hipDeviceSynchronize();
    hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
    printf("inv per round %d: mismatches %d (of %d)\n", inv, h, 512 * 7 * 20 * 8 * 512);
  }
  return 0;
}
