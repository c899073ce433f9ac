// Microbenchmark: latency of dependent float32 additions on one wave (gfx950), 1 / 2 / 4 independent chains per
// lane, plain and packed, with and without other waves on the same SIMD.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/dep_add.cpp -o gpurun_out/dep_add && gpurun_out/dep_add
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int kChains, bool kPacked>
__global__ void chains(const float *__restrict__ t, int n, float *out, long long *cycles) {
  float x[4] = {1.0f, 2.0f, 3.0f, 4.0f};
  v2f p[4] = {{1.0f, 2.0f}, {3.0f, 4.0f}, {5.0f, 6.0f}, {7.0f, 8.0f}};
  const float a = t[threadIdx.x & 7], b = t[8 + (threadIdx.x & 7)];
  const long long c0 = clock64();
  for (int i = 0; i < n; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const float term = (u & 1) ? a : b;
#pragma unroll
      for (int k = 0; k < kChains; k++) {
        if (kPacked) p[k] = p[k] + (v2f){term, term};
        else x[k] = x[k] + term;
      }
    }
  }
  const long long c1 = clock64();
  float r = 0.0f;
  for (int k = 0; k < kChains; k++) r += kPacked ? p[k].x + p[k].y : x[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = c1 - c0;
}

template <int kChains, bool kPacked>
void run(const char *name, int threads, int blocks, const float *d_t, float *d_out, long long *d_cyc) {
  const int n = 4096;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((chains<kChains, kPacked>), dim3(blocks), dim3(threads), 0, 0, d_t, n, d_out, d_cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((chains<kChains, kPacked>), dim3(blocks), dim3(threads), 0, 0, d_t, n, d_out, d_cyc);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  long long cyc = 0;
  hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost);
  const double steps = (double)n * 16;
  printf("%-28s threads %4d blocks %4d: %.2f ns per step (%d chains%s), s_memtime ticks per step %.2f\n", name, threads, blocks,
         ms * 1e6 / steps, kChains, kPacked ? ", packed" : "", (double)cyc / steps);
}

int main() {
  std::vector<float> h(16);
  for (int i = 0; i < 16; i++) h[i] = 1e-3f * (i + 1);
  float *d_t, *d_out;
  long long *d_cyc;
  hipMalloc(&d_t, 64);
  hipMalloc(&d_out, 1 << 22);
  hipMalloc(&d_cyc, 8);
  hipMemcpy(d_t, h.data(), 64, hipMemcpyHostToDevice);
  run<1, false>("1 wave", 64, 1, d_t, d_out, d_cyc);
  run<2, false>("1 wave", 64, 1, d_t, d_out, d_cyc);
  run<4, false>("1 wave", 64, 1, d_t, d_out, d_cyc);
  run<1, true>("1 wave", 64, 1, d_t, d_out, d_cyc);
  run<2, true>("1 wave", 64, 1, d_t, d_out, d_cyc);
  run<4, true>("1 wave", 64, 1, d_t, d_out, d_cyc);
  run<1, false>("8 waves (2 per SIMD)", 512, 1, d_t, d_out, d_cyc);
  run<1, false>("16 waves (4 per SIMD)", 1024, 1, d_t, d_out, d_cyc);
  run<1, false>("8 waves x 512 blocks", 512, 512, d_t, d_out, d_cyc);
  run<1, true>("8 waves (2 per SIMD)", 512, 1, d_t, d_out, d_cyc);
  return 0;
}
