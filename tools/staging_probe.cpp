// Host-pointer staging options for the seams a Go caller holds (pageable slices): what does it cost to get
// 120 MB to the device and 38 MB back?   hipcc -O2 -o /tmp/staging_probe tools/staging_probe.cpp -lpthread
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static void par_memcpy(char *d, const char *s, size_t n, int threads) {
  if (threads <= 1) { memcpy(d, s, n); return; }
  std::vector<std::thread> th;
  size_t per = (n + threads - 1) / threads;
  for (int t = 0; t < threads; t++) {
    size_t a = t * per, b = a + per > n ? n : a + per;
    if (a < b) th.emplace_back([=] { memcpy(d + a, s + a, b - a); });
  }
  for (auto &t : th) t.join();
}
int main() {
  const size_t n = 120u << 20;
  char *h = (char *)malloc(n);
  memset(h, 1, n);
  char *d; CK(hipMalloc(&d, n));
  hipStream_t st; CK(hipStreamCreate(&st));
  for (int rep = 0; rep < 3; rep++) {
    double t0 = now();
    CK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st));
    printf("pageable hipMemcpyAsync H2D: %.2f ms (%.1f GB/s)\n", (now() - t0) * 1e3, n / (now() - t0) / 1e9);
  }
  for (int rep = 0; rep < 3; rep++) {
    double t0 = now();
    CK(hipHostRegister(h, n, hipHostRegisterDefault));
    double t1 = now();
    CK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st));
    double t2 = now();
    CK(hipHostUnregister(h));
    printf("register %.2f ms + copy %.2f ms (%.1f GB/s) + unregister %.2f ms = %.2f ms\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3,
           n / (t2 - t1) / 1e9, (now() - t2) * 1e3, (now() - t0) * 1e3);
  }
  for (int threads : {1, 2, 4, 8}) for (size_t chunk : {(size_t)4 << 20, (size_t)16 << 20}) {
    const int slots = 3;
    char *pin[slots]; hipEvent_t ev[slots];
    for (int i = 0; i < slots; i++) { CK(hipHostMalloc((void **)&pin[i], chunk, hipHostMallocDefault)); CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming)); }
    for (int rep = 0; rep < 2; rep++) {
      double t0 = now();
      int k = 0;
      for (size_t off = 0; off < n; off += chunk, k++) {
        int sl = k % slots;
        size_t m = off + chunk > n ? n - off : chunk;
        if (k >= slots) CK(hipEventSynchronize(ev[sl]));
        par_memcpy(pin[sl], h + off, m, threads);
        CK(hipMemcpyAsync(d + off, pin[sl], m, hipMemcpyHostToDevice, st));
        CK(hipEventRecord(ev[sl], st));
      }
      CK(hipStreamSynchronize(st));
      if (rep) printf("ring %d x %zu MB, %d copy threads: %.2f ms (%.1f GB/s)\n", slots, chunk >> 20, threads, (now() - t0) * 1e3, n / (now() - t0) / 1e9);
    }
    for (int i = 0; i < slots; i++) { CK(hipHostFree(pin[i])); CK(hipEventDestroy(ev[i])); }
  }
  // pinned source for reference
  char *p; CK(hipHostMalloc((void **)&p, n, hipHostMallocDefault)); memset(p, 1, n);
  double t0 = now();
  CK(hipMemcpyAsync(d, p, n, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st));
  printf("pinned H2D: %.2f ms (%.1f GB/s)\n", (now() - t0) * 1e3, n / (now() - t0) / 1e9);
  t0 = now();
  CK(hipMemcpyAsync(p, d, n, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
  printf("pinned D2H: %.2f ms (%.1f GB/s)\n", (now() - t0) * 1e3, n / (now() - t0) / 1e9);
  t0 = now();
  CK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
  printf("pageable D2H: %.2f ms (%.1f GB/s)\n", (now() - t0) * 1e3, n / (now() - t0) / 1e9);
  t0 = now(); memcpy(h, p, n); printf("1-thread memcpy 120 MB: %.2f ms\n", (now() - t0) * 1e3);
  return 0;
}
