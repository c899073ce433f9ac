// Microbenchmark / probe: how many kernels of ONE process's streams can wait for one another on ONE device?
// N streams, one single-wave kernel each; kernel k waits for word k - 1 (set by kernel k - 1) and sets word k.  They
// are launched in REVERSE order (the waiters first): two streams that share a hardware queue in the wrong order never
// finish (each wait gives up after 2 s and says so).  HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues
// (default 4) -- the question behind tests/conftest.py's run_in_child and pcgx_icp_fit_multi's same-device rule.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/stream_chain.cpp -o tools/micro/stream_chain.bin
//   GPU_MAX_HW_QUEUES=24 tools/micro/stream_chain.bin 16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void link(unsigned long long *words, int k, int *gave_up) {
  if (threadIdx.x != 0 || k < 0) return;
  if (k > 0) {
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(&words[16 * (k - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0ull) {
      if (wall_clock64() - t0 > 200000000ll) {  // 2 s
        gave_up[k] = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(16);
    }
  }
  __hip_atomic_store(&words[16 * k], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 8;
  const int warm = argc > 2 ? atoi(argv[2]) : 1;  // 1: every stream has run a kernel before (its queue exists)
  const int stride = argc > 3 ? atoi(argv[3]) : 1;  // the chain's streams are every `stride`-th of n * stride created ones
  std::vector<hipStream_t> all(n * stride), st(n);
  for (auto &s : all) CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  for (int k = 0; k < n; k++) st[k] = all[k * stride];
  if (argc > 4) {  // ... and the stream behind each of them has been used first (a tree build on a pooled context)
    for (int k = 0; k < n; k++) hipLaunchKernelGGL(link, dim3(1), dim3(64), 0, all[k * stride + 1], (unsigned long long *)nullptr, -1, (int *)nullptr);
    CHECK(hipDeviceSynchronize());
  }
  unsigned long long *words;
  int *gave_up;
  char *scratch;
  CHECK(hipMalloc((void **)&scratch, 4096 * (size_t)n));
  CHECK(hipHostMalloc((void **)&words, 16 * 8 * n, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent));
  CHECK(hipHostMalloc((void **)&gave_up, 4 * n, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent));
  for (int rep = 0; rep < 2; rep++) {
    for (int k = 0; k < n; k++) { words[16 * k] = rep == 0 && warm ? 1ull : 0ull; gave_up[k] = 0; }
    if (rep == 0 && !warm) continue;
    for (int k = n - 1; k >= 0; k--) {
      if (argc > 5) {  // ... with a memset and a small pageable copy in front of the kernel, as a session's first step has
        static char pageable[256];
        CHECK(hipMemsetAsync(scratch + 4096 * k, 0, 1024, st[k]));
        CHECK(hipMemcpyAsync(scratch + 4096 * k + 2048, pageable, 8, hipMemcpyHostToDevice, st[k]));
      }
      hipLaunchKernelGGL(link, dim3(1), dim3(64), 0, st[k], words, k, gave_up);
    }
    CHECK(hipDeviceSynchronize());
  }
  int bad = 0;
  for (int k = 0; k < n; k++) bad += gave_up[k];
  printf("%d streams (of %d created), GPU_MAX_HW_QUEUES=%s, warm=%d: %s (%d waits gave up)\n", n, n * stride, getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "default",
         warm, bad ? "STUCK" : "all chained", bad);
  return 0;
}
