// Microbenchmark: what a kernel boundary costs behind a kernel that WROTE a lot.  gfx950's eight L2s are written back
// when a kernel ends / before the next one's workgroups start on that XCD; how long do the next kernel's workgroups wait,
// by the bytes the writer left dirty and by the kind of store (plain, nontemporal)?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/kernel_boundary.cpp -o tools/micro/kernel_boundary.bin && tools/micro/kernel_boundary.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>

// (a word per workgroup, folded on the host: 2048 atomics on one word would be a 20 us kernel of their own)
__device__ unsigned long long g_out[2048], g_in[1024];

template <int kKind>
__global__ __launch_bounds__(256) void writer(float4 *p, long n4) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const f4 v = {(float)i, 1.0f, 2.0f, 3.0f};
    if (kKind == 1) __builtin_nontemporal_store(v, reinterpret_cast<f4 *>(&p[i]));
    else p[i] = make_float4(v.x, v.y, v.z, v.w);
  }
  if (threadIdx.x == 0) g_out[blockIdx.x] = wall_clock64();
}
__global__ __launch_bounds__(256) void reader(const float4 *p, long n4, float *out) {
  float m = 0.0f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) m += p[i].x;
  if (m == 12345.678f) out[0] = m;
  if (threadIdx.x == 0) g_out[blockIdx.x] = wall_clock64();
}
// (keeps the GPU busy while the host enqueues the kernels under test: the boundary is then the GPU's, not the host's)
__global__ void hold(unsigned long long ticks) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
__global__ __launch_bounds__(256) void next_kernel() {
  if (threadIdx.x == 0) g_in[blockIdx.x] = wall_clock64();
}

int main() {
  const long max4 = 640000000 / 16;
  float4 *p;
  float *out;
  hipMalloc(&p, max4 * 16);
  hipMalloc(&out, 4);
  hipMemset(p, 0, max4 * 16);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const long mbs[] = {0, 1, 8, 32, 64, 160, 640};
  for (int kind = 0; kind < 3; kind++)
    for (long mb : mbs) {
      const long n4 = mb * 1000000 / 16;
      double first = 0, last = 0, tot = 0;
      const int reps = 6;
      for (int rep = 0; rep < reps; rep++) {
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(hold, dim3(1), dim3(64), 0, 0, 30000ull);  // 300 us
        if (kind == 0) hipLaunchKernelGGL((writer<0>), dim3(2048), dim3(256), 0, 0, p, n4);
        if (kind == 1) hipLaunchKernelGGL((writer<1>), dim3(2048), dim3(256), 0, 0, p, n4);
        if (kind == 2) hipLaunchKernelGGL(reader, dim3(2048), dim3(256), 0, 0, p, n4, out);
        hipLaunchKernelGGL(next_kernel, dim3(1024), dim3(256), 0, 0);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        static unsigned long long ho[2048], hi[1024];
        hipMemcpyFromSymbol(ho, HIP_SYMBOL(g_out), sizeof ho);
        hipMemcpyFromSymbol(hi, HIP_SYMBOL(g_in), sizeof hi);
        const unsigned long long w_out = *std::max_element(ho, ho + 2048);
        if (rep) {
          first += ((double)*std::min_element(hi, hi + 1024) - (double)w_out) / 100.0;
          last += ((double)*std::max_element(hi, hi + 1024) - (double)w_out) / 100.0;
          tot += ms * 1e3;
        }
      }
      printf("%-22s %4ld MB: the next kernel's first workgroup in %6.1f us after the last wave out, its last %6.1f us; both kernels %7.1f us\n",
             kind == 0 ? "plain stores" : kind == 1 ? "nontemporal stores" : "loads only", mb, first / (reps - 1), last / (reps - 1), tot / (reps - 1));
    }
  return 0;
}
