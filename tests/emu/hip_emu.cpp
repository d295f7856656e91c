/* hip_emu.cpp -- fiber scheduler of the test-only HIP emulation (see hip_emu.h). */
#include "hip_emu.h"

namespace emu {
Block g_block;
emu_dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;
static const size_t kStack = 256 * 1024;
static void (*g_tramp)(void *);
static void *g_arg;

char *dyn_lds() { return g_block.dyn_lds; }

void yield_fiber() {
  Fiber *f = g_block.cur;
  swapcontext(&f->ctx, &g_block.main_ctx);
  g_threadIdx.x = f->tid;  /* restored by the scheduler too; kept for clarity */
}

static void fiber_entry() {
  Fiber *f = g_block.cur;
  g_tramp(g_arg);
  f->done = true;
  swapcontext(&f->ctx, &g_block.main_ctx);
}

void run_block_impl(unsigned nthreads, unsigned bx, unsigned gx, size_t shmem, void (*tramp)(void *), void *arg) {
  Block &B = g_block;
  if (nthreads % 64 != 0) { fprintf(stderr, "hip_emu: block size must be a multiple of 64\n"); abort(); }
  static const bool reverse = getenv("QPALM_EMU_REVERSE") && atoi(getenv("QPALM_EMU_REVERSE"));
  if (B.fibers.size() < nthreads) {
    size_t old = B.fibers.size();
    B.fibers.resize(nthreads);
    for (size_t k = old; k < nthreads; k++) B.fibers[k].stack = (char *)malloc(kStack);
  }
  B.waves.assign(nthreads / 64, Wave());
  for (auto &w : B.waves) { w.count = 0; w.gen = 0; }
  B.nthreads = nthreads; B.bar_count = 0; B.bar_gen = 0;
  const size_t kGuard = 4096;                 /* catches kernels that run past their dynamic LDS allocation */
  const size_t lds_sz = shmem ? shmem : 16;
  B.dyn_lds = (char *)calloc(lds_sz + kGuard, 1);
  memset(B.dyn_lds + lds_sz, 0xA5, kGuard);
  g_tramp = tramp; g_arg = arg;
  g_blockIdx = emu_dim3(bx); g_blockDim = emu_dim3(nthreads); g_gridDim = emu_dim3(gx);
  for (unsigned t = 0; t < nthreads; t++) {
    Fiber &f = B.fibers[t];
    f.done = false; f.tid = t;
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack; f.ctx.uc_stack.ss_size = kStack; f.ctx.uc_link = &B.main_ctx;
    makecontext(&f.ctx, (void (*)())fiber_entry, 0);
  }
  for (;;) {
    bool alive = false;
    for (unsigned k = 0; k < nthreads; k++) {
      unsigned t = reverse ? nthreads - 1 - k : k;
      Fiber &f = B.fibers[t];
      if (f.done) continue;
      alive = true;
      B.cur = &f;
      g_threadIdx = emu_dim3(t);
      swapcontext(&B.main_ctx, &f.ctx);
    }
    if (!alive) break;
  }
  for (size_t k = 0; k < kGuard; k++)
    if ((unsigned char)B.dyn_lds[lds_sz + k] != 0xA5) { fprintf(stderr, "hip_emu: dynamic LDS overrun at byte %zu (allocation %zu)\n", lds_sz + k, lds_sz); abort(); }
  free(B.dyn_lds); B.dyn_lds = nullptr;
}
}  // namespace emu
