/*
 * qpalm_emu.cpp -- TEST-ONLY build of the kernel source + C ABI on the host (see hip_emu.h).
 * Never loaded by the product; qpg_backend_name() returns "host-emulation" so that tests can tell
 * the two libraries apart.
 */
#include "hip_emu.h"

#include <chrono>
#include <string>

#include "../../qpalm_amd/csrc/qpalm_kernels.h"

static std::string g_rt_err;
static int rt_device_init(int, std::string &) { return 0; }
static int rt_malloc(void **pp, size_t bytes) { *pp = calloc(bytes ? bytes : 1, 1); return *pp ? 0 : 1; }

#define RT_BACKEND_NAME "host-emulation"
#define RT_DEVICE_INIT(device, why) rt_device_init((device), (why))
#define RT_THREAD_DEVICE(device) (void)(device)
#define RT_MALLOC(pp, bytes) rt_malloc((void **)(pp), (bytes))
#define RT_FREE(p) free(p)
#define RT_HOST_ALLOC(pp, bytes) rt_malloc((void **)(pp), (bytes))
#define RT_HOST_FREE(p) free(p)
#define RT_MEMCPY_H2D(dst, src, bytes) memcpy((void *)(dst), (const void *)(src), (bytes))
#define RT_MEMCPY_D2H(dst, src, bytes) memcpy((void *)(dst), (const void *)(src), (bytes))
#define RT_MEMCPY2D_H2D(dst, dpitch, src, spitch, width, height) do { for (size_t r_ = 0; r_ < (size_t)(height); r_++) memcpy((char *)(dst) + r_ * (dpitch), (const char *)(src) + r_ * (spitch), (width)); } while (0)
#define RT_MEMSET(dst, val, bytes) memset((void *)(dst), (val), (bytes))
#define RT_MEMCPY_H2D_ASYNC(dst, src, bytes, k) (memcpy((void *)(dst), (const void *)(src), (bytes)), 0)
#define RT_COPY_MARK(k) do { } while (0)
#define RT_COPY_WAIT(k) do { } while (0)
#define RT_SYNC() 0
#define RT_STICKY() 0
#define RT_STICKY_CLEAR() do { } while (0)
#define RT_LAST_ERROR() (g_rt_err.c_str())
#define RT_LAUNCH(kernel, grid, block, shmem, ...) emu::launch(kernel, emu_dim3(grid), emu_dim3(block), (shmem), __VA_ARGS__)
#define RT_GRAPHS 0 /* the emulator runs every launch at once: nothing to record */
typedef int rt_graph_t;
#define RT_GRAPH_BEGIN() 1
#define RT_GRAPH_END(pexec) 1
#define RT_GRAPH_LAUNCH(exec) 0
#define RT_GRAPH_FREE(exec) (void)0
#define RT_USE_STREAM(k) (void)(k)
#define RT_RELEASE_STREAMS() do { } while (0)
#define RT_TIMED_LAUNCH(ms, kernel, grid, block, shmem, ...)                                   \
  do {                                                                                         \
    auto t0_ = std::chrono::steady_clock::now();                                               \
    emu::launch(kernel, emu_dim3(grid), emu_dim3(block), (shmem), __VA_ARGS__);                \
    (ms) = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0_).count(); \
  } while (0)

#include "../../qpalm_amd/csrc/qpalm_capi.inc"
