// Microbenchmark: what a word costs on its way from one workgroup to another INSIDE a launch, by where the two run.
// gfx950 has eight XCDs with an L2 each; workgroups are dealt to them round-robin (workgroup i -> XCD i % 8).  Two
// workgroups bounce a counter (64-bit words, a line each): A stores k, B polls until it reads k and stores k into its own
// word, A polls for that ... 2000 bounces, wall clock at both ends.
//   scope "agent": __hip_atomic_* with agent scope (sc1: what a cross-XCD exchange needs);
//   scope "wg+glc": workgroup-scope atomics -- L1 is bypassed?  (what the ISA gives below agent scope), same XCD only.
// Also: the XCD a workgroup runs on (XCC_ID hardware register), to check the dealing.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/xcd_pingpong.cpp -o tools/micro/xcd_pingpong.bin && tools/micro/xcd_pingpong.bin
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ unsigned long long g_t[4];
__device__ unsigned int g_xcc[64];

template <int kScope>
__device__ __forceinline__ unsigned long long ld(unsigned long long *p) {
  if (kScope == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <int kScope>
__device__ __forceinline__ void st(unsigned long long *p, unsigned long long v) {
  if (kScope == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int kScope>
__global__ __launch_bounds__(64) void bounce(unsigned long long *words, int a, int b, int n) {
  if (threadIdx.x == 0) {
    unsigned int xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g_xcc[blockIdx.x & 63] = xcc & 0xf;
  }
  if ((int)blockIdx.x != a && (int)blockIdx.x != b) return;
  if (threadIdx.x != 0) return;
  unsigned long long *wa = words, *wb = words + 32;  // (256 bytes apart)
  const unsigned long long t0 = wall_clock64();
  if ((int)blockIdx.x == a) {
    for (int k = 1; k <= n; k++) {
      st<kScope>(wa, (unsigned long long)k);
      long spins = 0;
      while (ld<kScope>(wb) != (unsigned long long)k)
        if (++spins > 100000000) return;
    }
    g_t[0] = t0;
    g_t[1] = wall_clock64();
  } else {
    for (int k = 1; k <= n; k++) {
      long spins = 0;
      while (ld<kScope>(wa) != (unsigned long long)k)
        if (++spins > 100000000) return;
      st<kScope>(wb, (unsigned long long)k);
    }
  }
}

int main() {
  unsigned long long *words;
  hipMalloc(&words, 4096);
  const int n = 2000;
  const int pairs[][2] = {{0, 8}, {0, 16}, {8, 16}, {0, 1}, {0, 4}, {3, 5}, {1, 9}};
  for (int scope = 0; scope < 2; scope++)
    for (auto &pr : pairs) {
      if (scope == 1 && (pr[0] % 8) != (pr[1] % 8)) continue;  // (below agent scope: same XCD only, or it never ends)
      hipMemset(words, 0, 4096);
      unsigned long long zero[4] = {0, 0, 0, 0};
      hipMemcpyToSymbol(HIP_SYMBOL(g_t), zero, sizeof(zero));
      if (scope == 0) hipLaunchKernelGGL(bounce<0>, dim3(32), dim3(64), 0, 0, words, pr[0], pr[1], n);
      else hipLaunchKernelGGL(bounce<1>, dim3(32), dim3(64), 0, 0, words, pr[0], pr[1], n);
      if (hipDeviceSynchronize() != hipSuccess) {
        printf("kernel failed\n");
        return 1;
      }
      unsigned long long t[4];
      unsigned int xcc[64];
      hipMemcpyFromSymbol(t, HIP_SYMBOL(g_t), sizeof(t));
      hipMemcpyFromSymbol(xcc, HIP_SYMBOL(g_xcc), sizeof(xcc));
      printf("scope %-6s workgroups %2d (XCC %u) <-> %2d (XCC %u): %.3f us a bounce (there and back)%s\n", scope == 0 ? "agent" : "wg", pr[0],
             xcc[pr[0]], pr[1], xcc[pr[1]], t[1] ? (double)(t[1] - t[0]) / 100.0 / n : -1.0, t[1] ? "" : "  (gave up)");
    }
  return 0;
}
