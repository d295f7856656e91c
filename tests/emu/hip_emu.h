/*
 * hip_emu.h -- development/test harness ONLY: runs the repository's HIP kernel SOURCE on the host.
 *
 * There is no GPU in the build container, so the kernel source (qpalm_amd/csrc/qpalm_device.h) is
 * also compiled with g++ against this header: every GPU thread of a workgroup becomes a ucontext
 * fiber, __syncthreads()/wave intrinsics switch fibers, workgroups run one after another.  This is
 * NOT a product path and NOT a fallback: the shipped library (libqpalm_gfx950.so) is built by
 * hipcc only and fails loudly without a GPU; the emulated build lives under tests/ and is loaded
 * only by the CPU tests (tests/test_emu_*.py) to exercise kernel and host logic before the same
 * source is run on a real MI355X by the `-m gpu` tests.
 *
 * Fibers switch only at barriers / wave intrinsics, and the scheduling order can be reversed
 * (QPALM_EMU_REVERSE=1) so that a missing barrier shows up as a wrong result in one of the orders.
 */
#ifndef HIP_EMU_H
#define HIP_EMU_H

#include <ucontext.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define QPALM_EMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__

struct emu_dim3 { unsigned x, y, z; emu_dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
typedef emu_dim3 dim3;

namespace emu {
struct Fiber { ucontext_t ctx; char *stack; bool done; unsigned tid; };
struct Wave { uint64_t slot[64][8]; int count; unsigned gen; };
struct Block {
  std::vector<Fiber> fibers;
  std::vector<Wave> waves;
  ucontext_t main_ctx;
  Fiber *cur;
  int bar_count; unsigned bar_gen; const char *bar_file; int bar_line;
  unsigned nthreads;
  char *dyn_lds;
};
extern Block g_block;
extern emu_dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;
void yield_fiber();
}  // namespace emu

#define threadIdx (emu::g_threadIdx)
#define blockIdx (emu::g_blockIdx)
#define blockDim (emu::g_blockDim)
#define gridDim (emu::g_gridDim)

/* every thread of the workgroup must arrive at the SAME barrier (source line): on the hardware s_barrier only counts
 * arrivals, so wavefronts that pair different barriers run on silently out of step -- here that aborts */
static inline void emu_syncthreads_at(const char *file, int line) {
  emu::Block &B = emu::g_block;
  unsigned gen = B.bar_gen;
  if (B.bar_count == 0) { B.bar_file = file; B.bar_line = line; }
  else if (B.bar_line != line || B.bar_file != file) {
    fprintf(stderr, "hip_emu: thread %d is at the barrier %s:%d while others wait at %s:%d\n", (int)emu::g_threadIdx.x, file, line, B.bar_file, B.bar_line);
    abort();
  }
  if (++B.bar_count == (int)B.nthreads) { B.bar_count = 0; B.bar_gen++; }
  else while (B.bar_gen == gen) emu::yield_fiber();
}
#define __syncthreads() emu_syncthreads_at(__FILE__, __LINE__)
static inline void emu_wave_sync() {
  emu::Wave &W = emu::g_block.waves[emu::g_threadIdx.x >> 6];
  unsigned gen = W.gen;
  if (++W.count == 64) { W.count = 0; W.gen++; }
  else while (W.gen == gen) emu::yield_fiber();
}
template <class T> static inline T emu_exchange(T v, int src) {
  static_assert(sizeof(T) <= 8, "emu_exchange");
  emu::Wave &W = emu::g_block.waves[emu::g_threadIdx.x >> 6];
  int lane = emu::g_threadIdx.x & 63;
  uint64_t bits = 0; memcpy(&bits, &v, sizeof(T));
  W.slot[lane][0] = bits;
  emu_wave_sync();
  uint64_t r = W.slot[src & 63][0];
  emu_wave_sync();
  T out; memcpy(&out, &r, sizeof(T));
  return out;
}
/* two values per lane published once, then read from any lane without further switches (the emulator's form of a run of DPP
 * broadcasts of the same registers): emu_publish2(a, b); ... emu_peek(lane, 0 | 1) ...; emu_wave_sync(); */
static inline void emu_publish2(double a, double b) {
  emu::Wave &W = emu::g_block.waves[emu::g_threadIdx.x >> 6];
  int lane = emu::g_threadIdx.x & 63;
  memcpy(&W.slot[lane][4], &a, 8);
  memcpy(&W.slot[lane][5], &b, 8);
  emu_wave_sync();
}
static inline double emu_peek(int lane, int k) {
  emu::Wave &W = emu::g_block.waves[emu::g_threadIdx.x >> 6];
  double v; memcpy(&v, &W.slot[lane & 63][4 + k], 8);
  return v;
}
template <class T> static inline T __shfl(T v, int src, int width = 64) {
  int lane = emu::g_threadIdx.x & 63;
  int base = lane & ~(width - 1);
  return emu_exchange(v, base + (src & (width - 1)));
}
template <class T> static inline T __shfl_xor(T v, int mask, int width = 64) {
  int lane = emu::g_threadIdx.x & 63;
  return emu_exchange(v, lane ^ mask);
}
template <class T> static inline T __shfl_down(T v, unsigned d, int width = 64) {
  int lane = emu::g_threadIdx.x & 63;
  int src = lane + (int)d;
  if ((src & ~(width - 1)) != (lane & ~(width - 1))) src = lane;
  return emu_exchange(v, src);
}
template <class T> static inline T __shfl_up(T v, unsigned d, int width = 64) {
  int lane = emu::g_threadIdx.x & 63;
  int src = lane - (int)d;
  if (src < 0 || (src & ~(width - 1)) != (lane & ~(width - 1))) src = lane;
  return emu_exchange(v, src);
}
static inline unsigned long long __ballot(int pred) {
  emu::Wave &W = emu::g_block.waves[emu::g_threadIdx.x >> 6];
  int lane = emu::g_threadIdx.x & 63;
  W.slot[lane][1] = pred ? 1 : 0;
  emu_wave_sync();
  unsigned long long m = 0;
  for (int l = 0; l < 64; l++) m |= (unsigned long long)W.slot[l][1] << l;
  emu_wave_sync();
  return m;
}
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int atomicAdd(int *p, int v) { int o = *p; *p += v; return o; }
static inline unsigned atomicAdd(unsigned *p, unsigned v) { unsigned o = *p; *p += v; return o; }
static inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { unsigned long long o = *p; *p += v; return o; }
#include <time.h>
static inline long long wall_clock64() { /* 100 MHz like the gfx950 constant-rate counter */
  struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t);
  return (long long)t.tv_sec * 100000000ll + t.tv_nsec / 10;
}

/* v_mfma_f64_16x16x4_f64 (cdna_hip_programming.md section 3): A[i=l&15][k=l>>4], B[k=l>>4][j=l&15],
 * C/D: col = l&15, row = (l>>4) + 4*reg. */
struct emu_double4 { double x, y, z, w; double &operator[](int i) { return (&x)[i]; } };
static inline emu_double4 emu_mfma_f64_16x16x4(double a, double b, emu_double4 c) {
  emu::Wave &W = emu::g_block.waves[emu::g_threadIdx.x >> 6];
  int lane = emu::g_threadIdx.x & 63;
  memcpy(&W.slot[lane][2], &a, 8);
  memcpy(&W.slot[lane][3], &b, 8);
  emu_wave_sync();
  emu_double4 d = c;
  int col = lane & 15;
  for (int r = 0; r < 4; r++) {
    int row = (lane >> 4) + 4 * r;
    double acc = c[r];
    for (int k = 0; k < 4; k++) {
      double av, bv;
      memcpy(&av, &W.slot[k * 16 + row][2], 8);
      memcpy(&bv, &W.slot[k * 16 + col][3], 8);
      acc = std::fma(av, bv, acc);
    }
    d[r] = acc;
  }
  emu_wave_sync();
  return d;
}

namespace emu {
char *dyn_lds();
void run_block_impl(unsigned nthreads, unsigned bx, unsigned gx, size_t shmem, void (*tramp)(void *), void *arg);

template <class K, class... Args> void launch(K kernel, emu_dim3 grid, emu_dim3 block, size_t shmem, Args... args) {
  auto fn = [&]() { kernel(args...); };
  using Fn = decltype(fn);
  for (unsigned b = 0; b < grid.x; b++)
    run_block_impl(block.x, b, grid.x, shmem, [](void *p) { (*static_cast<Fn *>(p))(); }, &fn);
}
}  // namespace emu

#endif
