// Microbenchmark: how long after a kernel's first workgroup has begun does its last one begin, by the size of the
// kernel's arguments (the waves fetch them with scalar loads before anything else) -- run it with and without
// HIP_FORCE_DEV_KERNARG=1 (arguments in device memory instead of host memory).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/kernarg_entry.cpp -o tools/micro/kernarg_entry.bin && tools/micro/kernarg_entry.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>

__device__ unsigned long long g_in[1024], g_in0[1024];
template <int W> struct Args { unsigned int w[W]; };

__global__ void hold(unsigned long long ticks) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
template <int W>
__global__ __launch_bounds__(256) void with_args(Args<W> a, unsigned int *sink) {
  const unsigned long long t0 = wall_clock64();  // (may or may not be ahead of the argument loads' wait)
  unsigned int s = 0;
  for (int k = 0; k < W; k++) s += a.w[k];
  asm volatile("" ::"s"(s));
  if (threadIdx.x == 0) {
    g_in0[blockIdx.x] = t0;
    g_in[blockIdx.x] = wall_clock64();
  }
  if (s == 0x12345u) sink[0] = s;
}

template <int W> void run(unsigned int *sink) {
  Args<W> a;
  for (int k = 0; k < W; k++) a.w[k] = k + 1;
  double spread = 0, spread0 = 0;
  for (int rep = 0; rep < 6; rep++) {
    hipDeviceSynchronize();
    hipLaunchKernelGGL(hold, dim3(1), dim3(64), 0, 0, 20000ull);
    hipLaunchKernelGGL((with_args<W>), dim3(1024), dim3(256), 0, 0, a, sink);
    hipDeviceSynchronize();
    static unsigned long long hi[1024], h0[1024];
    hipMemcpyFromSymbol(hi, HIP_SYMBOL(g_in), sizeof hi);
    hipMemcpyFromSymbol(h0, HIP_SYMBOL(g_in0), sizeof h0);
    if (rep) {
      spread += (double)(*std::max_element(hi, hi + 1024) - *std::min_element(h0, h0 + 1024)) / 100.0;
      spread0 += (double)(*std::max_element(h0, h0 + 1024) - *std::min_element(h0, h0 + 1024)) / 100.0;
    }
  }
  printf("%4d bytes of arguments: first clock read -> last workgroup has its arguments %5.1f us (first clock reads spread over %5.1f us)\n",
         (int)sizeof(Args<W>) + 8, spread / 5, spread0 / 5);
}

int main() {
  unsigned int *sink;
  hipMalloc(&sink, 4);
  run<2>(sink);
  run<8>(sink);
  run<16>(sink);
  run<32>(sink);
  run<48>(sink);
  run<64>(sink);
  run<128>(sink);
  return 0;
}
