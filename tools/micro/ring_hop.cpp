// Microbenchmark: what ONE hop of the sharded walk costs -- a tagged 64-bit word stored by one kernel and polled by
// another, both resident (two streams; on the target machine: two GPUs), as the walkers of strict_chain_kernel hand the
// state on (csrc/strict.hip, strict_enqueue_ring).  Ping-pong of kHops words between two single-wave kernels:
//   (a) through pinned host memory, system-scope stores and loads (what the ring uses: every GPU of a node -- and every
//       process -- can map it; a hop is a PCIe write + a PCIe read);
//   (b) through device memory, agent-scope (what the chunks of ONE GPU's chain kernel use);
//   poll pause s_sleep 1 / 8 / 32.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/ring_hop.cpp -o tools/micro/ring_hop.bin && tools/micro/ring_hop.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <bool kSystem, int kSleep>
__global__ void side(unsigned long long *mine, unsigned long long *theirs, int hops, int first, long long *ticks) {
  if (threadIdx.x != 0) return;
  long long t0 = 0;
  for (int h = 1; h <= hops; h++) {
    if (first || h > 1 || true) {
      if (first) {  // my turn first: store, then wait for the answer
        if (h == 1) t0 = wall_clock64();
        if (kSystem) __hip_atomic_store(theirs, (unsigned long long)h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else __hip_atomic_store(theirs, (unsigned long long)h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      long long spins = 0;
      while (true) {
        const unsigned long long v = kSystem ? __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                                             : __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v == (unsigned long long)h) break;
        if (++spins > 200000000ll) return;  // (an exit every wave reaches)
        __builtin_amdgcn_s_sleep(kSleep);
      }
      if (!first) {
        if (kSystem) __hip_atomic_store(theirs, (unsigned long long)h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else __hip_atomic_store(theirs, (unsigned long long)h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  if (first) *ticks = wall_clock64() - t0;
}

template <bool kSystem, int kSleep>
static void run(const char *what, unsigned long long *a, unsigned long long *b, long long *d_ticks, hipStream_t s0, hipStream_t s1) {
  const int hops = 2000;
  CHECK(hipMemset(d_ticks, 0, 8));
  a[0] = b[0] = 0;  // (host-visible in both cases: the device words are in managed-free pinned memory for (a), set by memset for (b))
  hipLaunchKernelGGL((side<kSystem, kSleep>), dim3(1), dim3(64), 0, s1, b, a, hops, 0, d_ticks);
  hipLaunchKernelGGL((side<kSystem, kSleep>), dim3(1), dim3(64), 0, s0, a, b, hops, 1, d_ticks);
  CHECK(hipDeviceSynchronize());
  long long t = 0;
  CHECK(hipMemcpy(&t, d_ticks, 8, hipMemcpyDeviceToHost));
  printf("%-46s s_sleep %2d: %.2f us per hop (one way)\n", what, kSleep, (double)t / 100.0 / (2.0 * hops));
}

int main() {
  hipStream_t s0, s1;
  CHECK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  long long *d_ticks;
  CHECK(hipMalloc((void **)&d_ticks, 8));
  unsigned long long *h;  // pinned host memory, two words a cache line apart
  CHECK(hipHostMalloc((void **)&h, 4096, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent));
  run<true, 1>("pinned host memory, system scope", h, h + 64, d_ticks, s0, s1);
  run<true, 8>("pinned host memory, system scope", h, h + 64, d_ticks, s0, s1);
  run<true, 32>("pinned host memory, system scope", h, h + 64, d_ticks, s0, s1);
  unsigned long long *d;  // device memory (fine-grained host-visible is not needed: the host only zeroes it)
  CHECK(hipHostMalloc((void **)&d, 8, 0));  // placeholder so that `a[0] = b[0] = 0` in run() has host words to clear
  unsigned long long *dev;
  CHECK(hipMalloc((void **)&dev, 4096));
  CHECK(hipMemset(dev, 0, 4096));
  // (device words cannot be cleared from the host by a plain store: run them once with fresh memory per variant)
  {
    const int hops = 2000;
    for (int v = 0; v < 2; v++) {
      CHECK(hipMemset(dev, 0, 4096));
      CHECK(hipMemset(d_ticks, 0, 8));
      if (v == 0) {
        hipLaunchKernelGGL((side<false, 1>), dim3(1), dim3(64), 0, s1, dev + 64, dev, hops, 0, d_ticks);
        hipLaunchKernelGGL((side<false, 1>), dim3(1), dim3(64), 0, s0, dev, dev + 64, hops, 1, d_ticks);
      } else {
        hipLaunchKernelGGL((side<false, 8>), dim3(1), dim3(64), 0, s1, dev + 64, dev, hops, 0, d_ticks);
        hipLaunchKernelGGL((side<false, 8>), dim3(1), dim3(64), 0, s0, dev, dev + 64, hops, 1, d_ticks);
      }
      CHECK(hipDeviceSynchronize());
      long long t = 0;
      CHECK(hipMemcpy(&t, d_ticks, 8, hipMemcpyDeviceToHost));
      printf("%-46s s_sleep %2d: %.2f us per hop (one way)\n", "device memory, agent scope", v == 0 ? 1 : 8, (double)t / 100.0 / (2.0 * hops));
    }
  }
  return 0;
}
