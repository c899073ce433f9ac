// Microbenchmark: what one vector load costs a CU's texture path (TA / TCP / TD) on gfx950 by the SHAPE of its addresses,
// with the table small enough (8 KB) that every access hits the vector L1 -- the part of the kNN search kernel's bound
// that is not misses.  20 waves per CU, every wave issues R rounds of four independent loads.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/vmem_shape.cpp -o tools/micro/vmem_shape.bin && tools/micro/vmem_shape.bin
#include <hip/hip_runtime.h>
#include <cstdio>

enum Shape { SAME, CONSECUTIVE, PAIRS, QUADS_OF_LINES, OWN_LINE, kShapes };
static const char *kNames[kShapes] = {"all lanes the same 16 bytes", "lanes consecutive (1 KB, 8 lines)", "four lanes per 64 bytes, groups apart",
                                      "sixteen lanes per line, lines apart", "every lane a line of its own"};

template <int W>
__global__ __launch_bounds__(64) void loads(const float *__restrict__ tab, int rounds, int shape, float *__restrict__ out) {
  const int lane = threadIdx.x;
  unsigned s = blockIdx.x * 7u;
  float acc = 0.0f;
  for (int r = 0; r < rounds; r++) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      unsigned rec;  // 16-byte record index into a 512-record (8 KB) table
      const unsigned k = s + r * 4 + u;
      if (shape == SAME) rec = k & 511u;
      else if (shape == CONSECUTIVE) rec = (k * 64u + lane) & 511u;
      else if (shape == PAIRS) rec = ((k + (lane >> 2) * 37u) * 4u + (lane & 3)) & 511u;
      else if (shape == QUADS_OF_LINES) rec = ((k + (lane >> 4) * 5u) * 8u + (lane & 7)) & 511u;
      else rec = ((k + lane * 13u) * 8u) & 511u;
      if (W == 4) {
        const float4 p = reinterpret_cast<const float4 *>(tab)[rec];
        acc += p.x + p.w;
      } else {
        acc += tab[rec * 4];
      }
    }
  }
  if (acc == 12345.678f) out[0] = acc;
}

int main() {
  float *tab, *out;
  hipMalloc(&tab, 8192);
  hipMalloc(&out, 4);
  hipMemset(tab, 0, 8192);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int waves = 256 * 20 * 3, rounds = 16;
  for (int w : {4, 1})
    for (int shape = 0; shape < kShapes; shape++) {
      float best = 1e9f;
      for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0);
        if (w == 4) hipLaunchKernelGGL(loads<4>, dim3(waves), dim3(64), 0, 0, tab, rounds, shape, out);
        else hipLaunchKernelGGL(loads<1>, dim3(waves), dim3(64), 0, 0, tab, rounds, shape, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
      }
      printf("%2d bytes per lane, %-40s %7.1f us  = %5.1f cycles of a CU per load instruction (2.4 GHz)\n", w * 4, kNames[shape], best * 1e3,
             best * 1e-3 * 2.4e9 / ((double)waves * rounds * 4 / 256.0));
    }
  return 0;
}
