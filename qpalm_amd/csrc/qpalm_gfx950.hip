/*
 * qpalm_gfx950.hip -- the shipped translation unit: gfx950 kernels + C ABI (include/qpalm_gfx950.h).
 * Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared (see qpalm_amd/build.py).
 * There is no CPU path in this library: qpg_ctx_create fails without a gfx950 device.
 */
#include <hip/hip_runtime.h>

#include <cstdio>
#include <string>
#include <type_traits>

#include "qpalm_types.h"

/* The kernels are instantiated twice (same source, different workgroup size):
 *   qp512: 512 threads = 8 wavefronts per QP, two workgroups per CU (76 KB dynamic LDS each, QPG_LDS_DEFAULT) -- the general instance;
 *   qp256: 256 threads = 4 wavefronts per QP, FOUR workgroups per CU (38 KB dynamic LDS each) -- for small QPs (factor of
 *          at most 256 rows, e.g. the mpc-160 workload): their phases are bound by dependent-load latency, not by work
 *          per QP, so the lever is more QPs in flight per CU (measured: DESIGN.md section 7). */
#define QP_T 512
#define QP_KSEL(RPT) ((RPT) <= 2 ? 16 : 8)
#define QP_FKC 32
#ifndef QP_USQ_512
#define QP_USQ_512 1
#endif
#define QP_USQ QP_USQ_512
namespace qp512 {
#include "qpalm_kernels.h"
}
#undef QP_T
#undef QP_KSEL
#undef QP_FKC
#undef QP_USQ
#undef QPALM_KERNELS_H
#undef QPALM_DEVICE_H
#undef QPALM_DENSE_H
#undef QPALM_ITER_H
#undef QPALM_KKT_H
#undef QPALM_SPARSE_H
#define QP_T 256
#define QP_KSEL(RPT) 8
#define QP_FKC 16
#define QP_USQ 0
namespace qp256 {
#include "qpalm_kernels.h"
static_assert(sizeof(UpdownLds<1, 8>) <= 38912 && sizeof(FactorLds) + sizeof(FactorStage) + 64 <= 38912 && sizeof(SolveLds) + 8 * 256 <= 38912,
              "the 256-thread instance runs with 38 KB of dynamic LDS");
}
#undef QP_T
#undef QP_KSEL
#undef QP_FKC
#undef QP_USQ
#undef QP_UNBS
#undef QPALM_KERNELS_H
#undef QPALM_DEVICE_H
#undef QPALM_DENSE_H
#undef QPALM_ITER_H
#undef QPALM_KKT_H
#undef QPALM_SPARSE_H
#ifndef QP_TINY_K
#define QP_TINY_K 8 /* ranks per update sweep of the 128-thread instance */
#endif
#ifndef QP_LDS_TINY
#define QP_LDS_TINY 22016
#endif
#ifndef QP_TINY_PER_CU
#define QP_TINY_PER_CU 7
#endif
#define QP_T 128
#define QP_KSEL(RPT) QP_TINY_K
#define QP_FKC 8
#define QP_USQ 0
#define QP_UNBS 16
#define QP_FST 2
namespace qp128 {
#include "qpalm_kernels.h"
static_assert(sizeof(UpdownLds<2, QP_TINY_K>) <= QP_LDS_TINY && sizeof(FactorLds) + sizeof(FactorStage) + 64 <= QP_LDS_TINY && sizeof(SolveLds) + 8 * 256 <= QP_LDS_TINY,
              "the 128-thread instance runs with 21.5 KB of dynamic LDS");
static_assert((sizeof(IterShared) + QP_LDS_TINY) * QP_TINY_PER_CU <= 160 * 1024, "workgroups of the 128-thread instance per CU");
}
#undef QP_FST
#undef QP_T
#define QP_T 512 /* the general instance's workgroup size (limits quoted by the host code) */
#define QP_T_SMALL 256
#define QP_LDS_SMALL 38912
#define QP_T_TINY 128
#define QP_TWO_INSTANCES 1

/* per host thread: qpg_batch_create allocates the device arena on a thread of its own, whose failure reaches the caller through
 * qpg_batch::arena_rc -- it must not write (or clear) the error state of whatever API call the caller's thread is in */
static thread_local std::string g_rt_err;
static thread_local int g_rt_sticky = 0; /* a failed copy/memset is remembered until the API call returns (api_ok() in qpalm_capi.inc) */
static int rt_check(hipError_t e, const char *what) {
  if (e == hipSuccess) return 0;
  g_rt_err = std::string(what) + ": " + hipGetErrorString(e);
  g_rt_sticky = 1;
  return 1;
}
static int rt_sync() {
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipGetLastError();
  return rt_check(e, "device");
}
static int rt_device_init(int device, std::string &why) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count == 0) { why = "no HIP device visible (this backend has no CPU fallback)"; return 1; }
  if (device < 0 || device >= count) { why = "device index out of range"; return 1; }
  if (hipSetDevice(device) != hipSuccess) { why = "hipSetDevice failed"; return 1; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) { why = "hipGetDeviceProperties failed"; return 1; }
  if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) { why = std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only"; return 1; }
  return 0;
}
#include <map>
#include <mutex>
template <class K> static void rt_allow_lds(K kernel, size_t shmem) { /* once per (device, kernel) and size (the attribute call is not free: coop mode launches thousands of kernels) */
  if (shmem <= 48 * 1024) return;
  static std::map<std::pair<int, const void *>, size_t> granted;
  static std::mutex mu;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(mu);
  size_t &g = granted[std::make_pair(dev, (const void *)kernel)];
  if (g >= shmem) return;
  (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
  g = shmem;
}

#define RT_BACKEND_NAME "gfx950-hip"
#define RT_DEVICE_INIT(device, why) rt_device_init((device), (why))
#define RT_THREAD_DEVICE(device) (void)hipSetDevice(device) /* the current device is per host thread */
#define RT_MALLOC(pp, bytes) rt_check(hipMalloc((void **)(pp), (bytes)), "hipMalloc")
#define RT_FREE(p) (void)hipFree(p)
#define RT_HOST_ALLOC(pp, bytes) rt_check(hipHostMalloc((void **)(pp), (bytes), hipHostMallocDefault), "hipHostMalloc")
#define RT_HOST_FREE(p) (void)hipHostFree(p)
#define RT_MEMCPY_H2D(dst, src, bytes) rt_check(hipMemcpy((void *)(dst), (const void *)(src), (bytes), hipMemcpyHostToDevice), "hipMemcpy H2D")
#define RT_MEMCPY_D2H(dst, src, bytes) rt_check(hipMemcpy((void *)(dst), (const void *)(src), (bytes), hipMemcpyDeviceToHost), "hipMemcpy D2H")
#define RT_MEMCPY2D_H2D(dst, dpitch, src, spitch, width, height) rt_check(hipMemcpy2D((void *)(dst), (dpitch), (const void *)(src), (spitch), (width), (height), hipMemcpyHostToDevice), "hipMemcpy2D H2D")
#define RT_MEMSET(dst, val, bytes) rt_check(hipMemset((void *)(dst), (val), (bytes)), "hipMemset")
/* asynchronous uploads from page-locked staging on a copy stream of their own; RT_COPY_MARK(k) records "everything issued so far" for
 * staging buffer k, RT_COPY_WAIT(k) waits for it on the host */
static thread_local hipStream_t g_rt_copy_stream = 0;
static thread_local hipEvent_t g_rt_copy_event[2] = {0, 0};
static thread_local int g_rt_copy_marked[2] = {0, 0};
static int rt_copy_async(void *dst, const void *src, size_t bytes) {
  if (!g_rt_copy_stream && rt_check(hipStreamCreateWithFlags(&g_rt_copy_stream, hipStreamNonBlocking), "hipStreamCreate")) return 1;
  return rt_check(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, g_rt_copy_stream), "hipMemcpyAsync H2D");
}
static void rt_copy_mark(int k) {
  if (!g_rt_copy_event[k] && hipEventCreateWithFlags(&g_rt_copy_event[k], hipEventDisableTiming) != hipSuccess) { g_rt_copy_event[k] = 0; (void)hipStreamSynchronize(g_rt_copy_stream); return; }
  (void)hipEventRecord(g_rt_copy_event[k], g_rt_copy_stream);
  g_rt_copy_marked[k] = 1;
}
static void rt_copy_wait(int k) {
  if (g_rt_copy_marked[k]) { rt_check(hipEventSynchronize(g_rt_copy_event[k]), "hipEventSynchronize"); g_rt_copy_marked[k] = 0; }
}
#define RT_MEMCPY_H2D_ASYNC(dst, src, bytes, k) rt_copy_async((void *)(dst), (const void *)(src), (bytes))
#define RT_COPY_MARK(k) rt_copy_mark(k)
#define RT_COPY_WAIT(k) rt_copy_wait(k)
#define RT_SYNC() rt_sync()
#define RT_STICKY() (g_rt_sticky)
#define RT_STICKY_CLEAR() (g_rt_sticky = 0)
#define RT_LAST_ERROR() (g_rt_err.c_str())
/* launches go to g_rt_stream: the null stream, or the capturing stream while a launch sequence is recorded into a graph.  The capture
 * state is per host thread: while one thread records a coop launch chain, launches of another thread (another batch / context) keep
 * going to the null stream instead of being recorded into the wrong graph. */
static thread_local hipStream_t g_rt_stream = 0, g_rt_capture_stream = 0;
/* Side streams for the coop launch chains of DIFFERENT QPs (round 5: batches between "a few" and "the chip is full"): the chains of the
 * members of one round are independent of each other, so each goes to its own stream and the hardware overlaps them.  The streams are
 * ordinary (blocking) streams: they synchronise implicitly with the null stream, on which the iteration kernel before and after them
 * and the scalar read-back run -- no events needed. */
#define RT_SIDE_STREAMS 32
static thread_local hipStream_t g_rt_side[RT_SIDE_STREAMS] = {0};
static thread_local hipStream_t g_rt_user_stream = 0; /* where plain launches and graph replays go while no capture is active */
static void rt_use_stream(int k) {
  if (k < 0) { g_rt_user_stream = 0; g_rt_stream = 0; return; }
  k %= RT_SIDE_STREAMS;
  if (!g_rt_side[k]) {
    const hipError_t e = hipStreamCreate(&g_rt_side[k]);
    if (e != hipSuccess) { /* the chain runs on the null stream instead (correct, not concurrent); said once, not swallowed */
      g_rt_side[k] = 0; (void)hipGetLastError();
      static thread_local bool told = false;
      if (!told) { told = true; g_rt_err = std::string("hipStreamCreate (coop side stream): ") + hipGetErrorString(e) + " -- the members' chains run one after the other"; fprintf(stderr, "qpalm_gfx950: %s\n", g_rt_err.c_str()); }
    }
  }
  g_rt_user_stream = g_rt_side[k];
  g_rt_stream = g_rt_user_stream;
}
/* the calling thread's side streams go with the context that used them (they are created lazily per host thread; a thread that ran coop
 * solves and drops its context no longer leaks them) */
static void rt_release_streams() {
  g_rt_user_stream = 0; g_rt_stream = 0;
  for (int k = 0; k < RT_SIDE_STREAMS; k++) if (g_rt_side[k]) { (void)hipStreamDestroy(g_rt_side[k]); g_rt_side[k] = 0; }
}
#define RT_RELEASE_STREAMS() rt_release_streams()
#define RT_USE_STREAM(k) rt_use_stream(k)
#define RT_LAUNCH(kernel, grid, block, shmem, ...)                                         \
  do {                                                                                      \
    rt_allow_lds(kernel, (shmem));                                                          \
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), (shmem), g_rt_stream, __VA_ARGS__); \
  } while (0)
/* HIP graphs for the fixed launch chains of coop mode (one kernel node per launch, linear dependencies): the host enqueues one
 * graph instead of up to ~500 kernels per factorisation / solve / update sweep */
#define RT_GRAPHS 1
typedef hipGraphExec_t rt_graph_t;
static int rt_graph_begin() {
  if (!g_rt_capture_stream && hipStreamCreateWithFlags(&g_rt_capture_stream, hipStreamNonBlocking) != hipSuccess) { g_rt_capture_stream = 0; return 1; }
  if (hipStreamBeginCapture(g_rt_capture_stream, hipStreamCaptureModeRelaxed) != hipSuccess) { (void)hipGetLastError(); return 1; }
  g_rt_stream = g_rt_capture_stream;
  return 0;
}
static int rt_graph_end(rt_graph_t *exec) {
  hipGraph_t graph = nullptr;
  g_rt_stream = g_rt_user_stream;
  if (hipStreamEndCapture(g_rt_capture_stream, &graph) != hipSuccess || !graph) { (void)hipGetLastError(); return 1; }
  const hipError_t e = hipGraphInstantiate(exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) { (void)hipGetLastError(); return 1; }
  return 0;
}
#define RT_GRAPH_BEGIN() rt_graph_begin()
#define RT_GRAPH_END(pexec) rt_graph_end(pexec)
#define RT_GRAPH_LAUNCH(exec) rt_check(hipGraphLaunch((exec), g_rt_user_stream), "hipGraphLaunch")
#define RT_GRAPH_FREE(exec) (void)hipGraphExecDestroy(exec)
#define RT_TIMED_LAUNCH(ms, kernel, grid, block, shmem, ...)                                \
  do {                                                                                      \
    hipEvent_t e0_, e1_;                                                                    \
    hipEventCreate(&e0_); hipEventCreate(&e1_);                                             \
    rt_allow_lds(kernel, (shmem));                                                          \
    hipEventRecord(e0_, 0);                                                                 \
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), (shmem), 0, __VA_ARGS__);           \
    hipEventRecord(e1_, 0);                                                                 \
    hipEventSynchronize(e1_);                                                               \
    hipEventElapsedTime(&(ms), e0_, e1_);                                                   \
    hipEventDestroy(e0_); hipEventDestroy(e1_);                                             \
  } while (0)

#include "qpalm_capi.inc"
