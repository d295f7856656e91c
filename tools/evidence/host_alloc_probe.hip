// Setup-path probe (run on the GPU box): what page-locked allocation, first-touch of pageable memory (4 KB / transparent huge pages),
// hipHostRegister, the device arena allocation and host-to-device copies cost.   hipcc -O2 -o host_alloc_probe host_alloc_probe.hip -lpthread
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void touch(char *p, size_t bytes, int nt) {
  std::vector<std::thread> th;
  for (int t = 0; t < nt; t++) th.emplace_back([=]() { const size_t a = bytes / nt * t, b = (t == nt - 1) ? bytes : bytes / nt * (t + 1); memset(p + a, 1, b - a); });
  for (auto &t : th) t.join();
}
int main() {
  hipFree(0);
  const size_t GB = (size_t)1 << 30;
  double t;
  for (size_t sz : {(size_t)384 << 20, 2 * GB, 6 * GB}) {
    void *p = nullptr;
    t = now(); hipHostMalloc(&p, sz, hipHostMallocDefault); const double ta = now() - t;
    t = now(); touch((char *)p, sz, 8); const double tt = now() - t;
    t = now(); touch((char *)p, sz, 8); const double tt2 = now() - t;
    t = now(); hipHostFree(p); const double tf = now() - t;
    printf("hipHostMalloc %5.2f GB: alloc %.3f s, first touch (8 thr) %.3f s, second %.3f s, free %.3f s\n", sz / 1e9, ta, tt, tt2, tf);
  }
  for (int huge = 0; huge < 2; huge++)
    for (int nt : {8, 32}) {
      const size_t sz = 6 * GB;
      t = now();
      char *p = (char *)mmap(nullptr, sz, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
      if (huge) madvise(p, sz, MADV_HUGEPAGE);
      const double ta = now() - t;
      t = now(); touch(p, sz, nt); const double tt = now() - t;
      t = now(); touch(p, sz, nt); const double tt2 = now() - t;
      double tr = -1, tc = -1;
      void *d = nullptr; hipMalloc(&d, sz);
      t = now(); hipMemcpy(d, p, sz, hipMemcpyHostToDevice); const double tcp = now() - t;
      if (nt == 8) {
        t = now(); const int rc = hipHostRegister(p, sz, hipHostRegisterDefault); tr = now() - t;
        if (rc == 0) { t = now(); hipMemcpy(d, p, sz, hipMemcpyHostToDevice); tc = now() - t; hipHostUnregister(p); }
      }
      hipFree(d);
      t = now(); munmap(p, sz); const double tf = now() - t;
      printf("mmap %s 6.4 GB, %2d threads: map %.3f s, first touch %.3f s, second %.3f s, H2D pageable %.3f s, register %.3f s, H2D registered %.3f s, unmap %.3f s\n",
             huge ? "THP " : "4 KB", nt, ta, tt, tt2, tcp, tr, tc, tf);
    }
  {
    const size_t sz = 70 * GB;
    void *d = nullptr;
    t = now(); const int rc = hipMalloc(&d, sz); const double ta = now() - t;
    t = now(); hipMemset(d, 0, sz); hipDeviceSynchronize(); const double tm = now() - t;
    t = now(); hipFree(d); const double tf = now() - t;
    printf("hipMalloc 75 GB: rc %d alloc %.3f s, memset %.3f s, free %.3f s\n", rc, ta, tm, tf);
    t = now(); hipMalloc(&d, sz); printf("   again: alloc %.3f s\n", now() - t); hipFree(d);
  }
  FILE *f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
  if (f) { char b[128] = {0}; fgets(b, 127, f); printf("THP: %s", b); fclose(f); }
  printf("hardware threads %u\n", std::thread::hardware_concurrency());
  return 0;
}
