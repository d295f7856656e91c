/*
 * qpalm_gfx950.hip -- the shipped translation unit: gfx950 kernels + C ABI (include/qpalm_gfx950.h).
 * Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared (see qpalm_amd/build.py).
 * There is no CPU path in this library: qpg_ctx_create fails without a gfx950 device.
 */
#include <hip/hip_runtime.h>

#include <string>

#include "qpalm_kernels.h"

static std::string g_rt_err;
static int g_rt_sticky = 0; /* a failed copy/memset is remembered until the API call returns (api_ok() in qpalm_capi.inc) */
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
template <class K> static void rt_allow_lds(K kernel, size_t shmem) {
  if (shmem > 48 * 1024) (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
}

#define RT_BACKEND_NAME "gfx950-hip"
#define RT_DEVICE_INIT(device, why) rt_device_init((device), (why))
#define RT_MALLOC(pp, bytes) rt_check(hipMalloc((void **)(pp), (bytes)), "hipMalloc")
#define RT_FREE(p) (void)hipFree(p)
#define RT_MEMCPY_H2D(dst, src, bytes) rt_check(hipMemcpy((void *)(dst), (const void *)(src), (bytes), hipMemcpyHostToDevice), "hipMemcpy H2D")
#define RT_MEMCPY_D2H(dst, src, bytes) rt_check(hipMemcpy((void *)(dst), (const void *)(src), (bytes), hipMemcpyDeviceToHost), "hipMemcpy D2H")
#define RT_MEMSET(dst, val, bytes) rt_check(hipMemset((void *)(dst), (val), (bytes)), "hipMemset")
#define RT_SYNC() rt_sync()
#define RT_STICKY() (g_rt_sticky)
#define RT_STICKY_CLEAR() (g_rt_sticky = 0)
#define RT_LAST_ERROR() (g_rt_err.c_str())
#define RT_LAUNCH(kernel, grid, block, shmem, ...)                                         \
  do {                                                                                      \
    rt_allow_lds(kernel, (shmem));                                                          \
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), (shmem), 0, __VA_ARGS__);           \
  } while (0)
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
