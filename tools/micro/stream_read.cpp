// Microbenchmark: how fast gfx950 streams 120 MB (C3's cloud) through a reduction -- float4 loads, a max per lane -- by
// grid size, loads in flight per thread and load flavour; a 1.5 GB buffer is swept between repetitions so that the
// data comes from HBM, not from the 256 MB Infinity Cache.  (The voxel filter's min/max pass: 36 us for the stream.)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/stream_read.cpp -o tools/micro/stream_read.bin && tools/micro/stream_read.bin
#include <hip/hip_runtime.h>
#include <cstdio>

template <int U, bool kNt>
__global__ __launch_bounds__(256) void reduce_max(const float4 *__restrict__ p, long n4, float *out) {
  float m = -1e30f;
  const long step = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += U * step) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const long j = i + u * step < n4 ? i + u * step : i;
      if (kNt) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(&p[j]));
        v[u] = make_float4(t.x, t.y, t.z, t.w);
      } else {
        v[u] = p[j];
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) m = fmaxf(fmaxf(m, v[u].x), fmaxf(v[u].y, fmaxf(v[u].z, v[u].w)));
  }
  if (m == 12345.678f) out[0] = m;
}

__global__ void sweep(float4 *p, long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) p[i].x += 1.0f;
}
// (the sweep with streaming stores: do its lines stay behind dirty?)
__global__ void sweep_nt(float4 *p, long n4) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const f4 v = {(float)i, 1.0f, 2.0f, 3.0f};
    __builtin_nontemporal_store(v, reinterpret_cast<f4 *>(&p[i]));
  }
}
// (the same sweep, reading only: the caches are left full of clean lines)
__global__ void sweep_clean(const float4 *p, long n4, float *out) {
  float m = 0.0f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) m += p[i].x;
  if (m == 12345.678f) out[0] = m;
}

// The min/max pass's own shape: a thread takes 48 consecutive bytes (4 packed points) with three 16-byte loads, U such
// groups in flight.
template <int U, bool kNt>
__global__ __launch_bounds__(256) void reduce_max3(const float4 *__restrict__ p, long n4, float *out) {
  float m = -1e30f;
  const long groups = n4 / 3, step = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < groups; i += U * step) {
    float4 v[U][3];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const long j = i + u * step < groups ? i + u * step : i;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        if (kNt) {
          typedef float f4 __attribute__((ext_vector_type(4)));
          const f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(&p[3 * j + k]));
          v[u][k] = make_float4(t.x, t.y, t.z, t.w);
        } else {
          v[u][k] = p[3 * j + k];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
      for (int k = 0; k < 3; k++) m = fmaxf(fmaxf(m, v[u][k].x), fmaxf(v[u][k].y, fmaxf(v[u][k].z, v[u][k].w)));
  }
  if (m == 12345.678f) out[0] = m;
}

int main() {
  const long n4 = 120000000 / 16, big4 = 1500000000 / 16;
  float4 *p, *big;
  float *out;
  hipMalloc(&p, n4 * 16);
  hipMalloc(&big, big4 * 16);
  hipMalloc(&out, 4);
  hipMemset(p, 0, n4 * 16);
  hipMemset(big, 0, big4 * 16);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto run = [&](const char *what, auto launch, int cold) {
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
      if (cold == 1) hipLaunchKernelGGL(sweep, dim3(4096), dim3(256), 0, 0, big, big4);
      if (cold == 2) hipLaunchKernelGGL(sweep_clean, dim3(4096), dim3(256), 0, 0, big, big4, out);
      if (cold == 3) hipLaunchKernelGGL(sweep_nt, dim3(4096), dim3(256), 0, 0, big, big4);
      hipEventRecord(e0);
      launch();
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep && ms < best) best = ms;
    }
    printf("%-44s %s: %6.1f us = %5.2f TB/s\n", what, cold == 3 ? "from HBM, behind nt stores" : cold == 1 ? "from HBM, caches dirty" : cold == 2 ? "from HBM, caches clean" : "re-read", best * 1e3, 120e6 / (best * 1e-3) / 1e12);
  };
  for (int cold = 3; cold >= 0; cold--) {
    run("1024 blocks, 2 loads in flight", [&] { hipLaunchKernelGGL((reduce_max<2, false>), dim3(1024), dim3(256), 0, 0, p, n4, out); }, cold);
    run("2048 blocks, 4 loads in flight", [&] { hipLaunchKernelGGL((reduce_max<4, false>), dim3(2048), dim3(256), 0, 0, p, n4, out); }, cold);
    run("4096 blocks, 4 loads in flight", [&] { hipLaunchKernelGGL((reduce_max<4, false>), dim3(4096), dim3(256), 0, 0, p, n4, out); }, cold);
    run("4096 blocks, 8 loads in flight", [&] { hipLaunchKernelGGL((reduce_max<8, false>), dim3(4096), dim3(256), 0, 0, p, n4, out); }, cold);
    run("8192 blocks, 4 loads in flight", [&] { hipLaunchKernelGGL((reduce_max<4, false>), dim3(8192), dim3(256), 0, 0, p, n4, out); }, cold);
    run("29297 blocks (one pass), 1 load", [&] { hipLaunchKernelGGL((reduce_max<1, false>), dim3(29297), dim3(256), 0, 0, p, n4, out); }, cold);
    run("4096 blocks, 4 nontemporal loads", [&] { hipLaunchKernelGGL((reduce_max<4, true>), dim3(4096), dim3(256), 0, 0, p, n4, out); }, cold);
    run("1024 blocks, 2 nontemporal loads", [&] { hipLaunchKernelGGL((reduce_max<2, true>), dim3(1024), dim3(256), 0, 0, p, n4, out); }, cold);
    run("1024 blocks, 4 nontemporal loads", [&] { hipLaunchKernelGGL((reduce_max<4, true>), dim3(1024), dim3(256), 0, 0, p, n4, out); }, cold);
    run("2048 blocks, 4 nontemporal loads", [&] { hipLaunchKernelGGL((reduce_max<4, true>), dim3(2048), dim3(256), 0, 0, p, n4, out); }, cold);
    run("8192 blocks, 2 nontemporal loads", [&] { hipLaunchKernelGGL((reduce_max<2, true>), dim3(8192), dim3(256), 0, 0, p, n4, out); }, cold);
    run("1024 blocks, 2x3 loads (48 B a thread)", [&] { hipLaunchKernelGGL((reduce_max3<2, false>), dim3(1024), dim3(256), 0, 0, p, n4, out); }, cold);
    run("1024 blocks, 2x3 nontemporal (48 B)", [&] { hipLaunchKernelGGL((reduce_max3<2, true>), dim3(1024), dim3(256), 0, 0, p, n4, out); }, cold);
    run("4096 blocks, 2x3 nontemporal (48 B)", [&] { hipLaunchKernelGGL((reduce_max3<2, true>), dim3(4096), dim3(256), 0, 0, p, n4, out); }, cold);
    run("4096 blocks, 1x3 nontemporal (48 B)", [&] { hipLaunchKernelGGL((reduce_max3<1, true>), dim3(4096), dim3(256), 0, 0, p, n4, out); }, cold);
  }
  return 0;
}
