// Microbenchmark: the one-launch Fit's sums (csrc/icp_small.hip) -- a float32 sum of N terms in order, 64 terms a step
// of the wave: (a) v_readlane + v_add_f32 per term, (b) the running sum hopping from lane to lane (v_add_f32_dpp
// wave_shr:1), (c) like (b) with row_shr:1 inside rows of 16 and a hop between rows; ns per term by the wall clock and
// shader-clock cycles per term (s_memtime), alone on the CU and beside waves that poll a word in memory.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/dpp_chain.cpp -o tools/micro/dpp_chain.bin && tools/micro/dpp_chain.bin
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ unsigned long long g_t[8];
__device__ unsigned int g_flag;

#include "chain64.inc"

template <int kKind>
__global__ __launch_bounds__(512) void chain(const float *__restrict__ t, int nblocks, float *out, int pollers) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave != 1) {
    if (!pollers) return;
    // the other waves: look at a word until wave 1 is through
    while (__hip_atomic_load(&g_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(1);
    return;
  }
  float s = 0.0f;
  const unsigned long long w0 = wall_clock64(), c0 = clock64();
  float next = t[lane];
  for (int b = 0; b < nblocks; b++) {
    const float term = next;
    if (b + 1 < nblocks) next = t[(b + 1) * 64 + lane];  // (the next block's terms on their way under this block's adds)
    if (kKind == 0) {
#pragma unroll
      for (int k = 0; k < 64; k++) s = s + __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, term), k));
    } else if (kKind == 3) {
      // the 64 terms through LDS: written by the lanes, read back four at a time at an address that is the same for all
      // lanes (a broadcast) -- the add's operand is then a vector register, and the adds are the only dependent chain
      __shared__ float s_t[8][64];
      s_t[wave][lane] = term;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < 16; q++) {
        const float4 t4 = *reinterpret_cast<const float4 *>(&s_t[wave][4 * q]);
        s = s + t4.x;
        s = s + t4.y;
        s = s + t4.z;
        s = s + t4.w;
      }
      __builtin_amdgcn_wave_barrier();
    } else if (kKind == 2) {
      s = chain64(s, term);
    } else {
      float run = s + term;
#pragma unroll
      for (int k = 1; k < 64; k++)
        run = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, run), 0x138, 0xf, 0xf, false)) + term;
      s = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, run), 63));
    }
  }
  const unsigned long long w1 = wall_clock64(), c1 = clock64();
  if (lane == 0) {
    g_t[0] = w1 - w0;
    g_t[1] = c1 - c0;
    out[0] = s;
    __hip_atomic_store(&g_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

int main() {
  const int nblocks = 256;  // 16384 terms
  float *t, *out;
  hipMalloc(&t, nblocks * 64 * 4);
  hipMalloc(&out, 64);
  float *h = new float[nblocks * 64];
  for (int i = 0; i < nblocks * 64; i++) h[i] = 1.0f / (float)(1 + i % 97);
  hipMemcpy(t, h, nblocks * 64 * 4, hipMemcpyHostToDevice);
  float ref = 0.0f;
  for (int i = 0; i < nblocks * 64; i++) ref = ref + h[i];
  for (int kind = 0; kind < 4; kind++)
    for (int pollers = 0; pollers < 2; pollers++)
      for (int rep = 0; rep < 2; rep++) {
        unsigned int zero = 0;
        hipMemcpyToSymbol(HIP_SYMBOL(g_flag), &zero, 4);
        if (kind == 0) hipLaunchKernelGGL(chain<0>, dim3(1), dim3(512), 0, 0, t, nblocks, out, pollers);
        else if (kind == 3) hipLaunchKernelGGL(chain<3>, dim3(1), dim3(512), 0, 0, t, nblocks, out, pollers);
        else if (kind == 2) hipLaunchKernelGGL(chain<2>, dim3(1), dim3(512), 0, 0, t, nblocks, out, pollers);
        else hipLaunchKernelGGL(chain<1>, dim3(1), dim3(512), 0, 0, t, nblocks, out, pollers);
        hipDeviceSynchronize();
        unsigned long long g[8];
        float r;
        hipMemcpyFromSymbol(g, HIP_SYMBOL(g_t), sizeof(g));
        hipMemcpy(&r, out, 4, hipMemcpyDeviceToHost);
        if (rep == 1)
          printf("%-28s %s: %.2f ns a term, %.1f shader cycles a term (shader clock %.2f GHz); sum %s\n", kind == 0 ? "v_readlane + v_add_f32" : (kind == 2 ? "the same, v_readlane 8 ahead" : (kind == 3 ? "terms broadcast out of LDS" : "v_add_f32_dpp wave_shr:1")),
                 pollers ? "beside polling waves" : "alone               ", g[0] * 10.0 / (nblocks * 64), (double)g[1] / (nblocks * 64), (double)g[1] / (g[0] * 10.0),
                 r == ref ? "== the host's sequential sum" : "DIFFERS");
      }
  return 0;
}
