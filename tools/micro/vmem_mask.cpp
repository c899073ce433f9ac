// Microbenchmark: what a 16-byte-per-lane gather (global_load_dwordx4, every lane its own record of a 16 MB table) costs
// a CU's texture path on gfx950 when only k of the wave's 64 lanes are switched on -- the question behind the kNN search
// kernel, whose scans run until the wave's longest lane is done (a quarter of the lanes active in its second scan).
// Every wave issues R rounds of four independent loads; the grid fills the chip (5 waves per SIMD, as the search
// kernel's registers allow).  If the time does not fall with k, a load costs per INSTRUCTION, not per lane.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/vmem_mask.cpp -o tools/micro/vmem_mask.bin && tools/micro/vmem_mask.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int U>
__global__ __launch_bounds__(64) void gather(const float4 *__restrict__ tab, unsigned mask_n, int rounds, int active, int local,
                                             float *__restrict__ out) {
  const int lane = threadIdx.x;
  unsigned idx = (blockIdx.x * 64u + lane) * 2654435761u;
  float acc = 0.0f;
  if (lane < active) {
    for (int r = 0; r < rounds; r++) {
      float4 p[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        // local: the lanes of a wave read neighbouring records (a query-sorted batch); else anywhere in the table
        const unsigned at = local ? ((blockIdx.x * 977u + r * U + u) * 61u + lane * 3u) & mask_n : (idx + u * 0x9e3779b9u) & mask_n;
        p[u] = tab[at];
      }
#pragma unroll
      for (int u = 0; u < U; u++) acc += p[u].x + p[u].w;
      idx = idx * 1664525u + 1013904223u + (unsigned)(acc > 1e30f);
    }
  }
  if (acc == 12345.678f) out[0] = acc;
}

int main() {
  const unsigned n = 1u << 20;  // 16 MB of records
  float4 *tab;
  float *out;
  hipMalloc(&tab, n * sizeof(float4));
  hipMalloc(&out, 4);
  hipMemset(tab, 0, n * sizeof(float4));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int waves = 15632, rounds = 14;  // the search kernel's launch: 1M queries, ~56 lane-slots of scans each
  for (int inflight : {4, 8})
  for (int local = 1; local >= 0; local--)
    for (int active : {64, 32, 16, 8, 1}) {
      float best = 1e9f;
      for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0);
        if (inflight == 4) hipLaunchKernelGGL(gather<4>, dim3(waves), dim3(64), 0, 0, tab, n - 1, rounds, active, local, out);
        else hipLaunchKernelGGL(gather<8>, dim3(waves), dim3(64), 0, 0, tab, n - 1, rounds / 2, active, local, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
      }
      printf("%d in flight, %s records, %2d of 64 lanes active: %7.1f us for %d waves x %d loads  (%.1f cycles of a CU's texture path per load at 2.4 GHz)\n",
             inflight, local ? "neighbouring" : "scattered   ", active, best * 1e3, waves, rounds * 4,
             best * 1e-3 * 2.4e9 / ((double)waves * rounds * 4 / 256.0));
    }
  return 0;
}
