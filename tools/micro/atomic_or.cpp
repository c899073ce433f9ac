// Microbenchmark: 10M device-scope atomicOr (no return) on the words of a small bitmap -- what a bit per dense cell,
// set by the key kernel, would cost the voxel filter.  Random words, words in point order (a wave's 64
// lanes on a few neighbouring words), and with the lanes of a wave that hit one word combined first.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/atomic_or.cpp -o tools/micro/atomic_or.bin && tools/micro/atomic_or.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>

__global__ void set_bits(const uint32_t *__restrict__ key, long n, uint32_t *__restrict__ bitmap, int dedup) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const uint32_t k = key[i], w = k >> 5;
    uint32_t bit = 1u << (k & 31u);
    if (dedup) {  // lanes of this wave with the same word: the first of them ORs for all
      uint64_t m = __ballot(1);
      bool leader = true;
      // match by the word's low 12 bits would not be exact: compare with every earlier distinct leader (few per wave here)
      uint64_t todo = m;
      while (todo) {
        const int l = __ffsll((long long)todo) - 1;
        const uint32_t wl = (uint32_t)__shfl((int)w, l);
        const uint64_t same = __ballot(w == wl);
        uint32_t acc = 0;
        // OR of the bits of the lanes in `same`
        uint32_t b = (w == wl) ? bit : 0u;
        for (int o = 32; o > 0; o >>= 1) b |= (uint32_t)__shfl_xor((int)b, o);
        acc = b;
        if ((int)(threadIdx.x & 63) == l) atomicOr(&bitmap[wl], acc);
        todo &= ~same;
      }
      (void)leader;
    } else {
      atomicOr(&bitmap[w], bit);
    }
  }
}

int main() {
  const long n = 10000000;
  const uint32_t range = 3500000;  // bits (C3: some 3.4M dense cells)
  std::vector<uint32_t> hk(n);
  uint32_t *key, *bitmap;
  hipMalloc(&key, n * 4);
  hipMalloc(&bitmap, 8 << 20);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  std::mt19937 rng(1);
  for (int mode = 0; mode < 3; mode++) {
    for (long i = 0; i < n; i++) {
      if (mode == 0) hk[i] = rng() % range;                                   // anywhere
      else if (mode == 1) hk[i] = (uint32_t)((i * (long)range) / n);          // in key order: 3 points per cell, neighbours
      else hk[i] = (uint32_t)(((i / 64) * 64 * (long)range) / n) + rng() % 4096;  // a wave's points within 4096 cells (a slab)
    }
    hipMemcpy(key, hk.data(), n * 4, hipMemcpyHostToDevice);
    for (int dedup = 0; dedup < 2; dedup++) {
      if (dedup && mode == 0) continue;
      float best = 1e9f;
      for (int rep = 0; rep < 4; rep++) {
        hipMemset(bitmap, 0, 8 << 20);
        hipEventRecord(e0);
        hipLaunchKernelGGL(set_bits, dim3(4096), dim3(256), 0, 0, key, n, bitmap, dedup);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
      }
      printf("%-40s %s: %7.1f us for 10M points (reading their keys: 40 MB)\n",
             mode == 0 ? "keys anywhere in 3.5M cells" : mode == 1 ? "keys in order" : "a wave's keys within 4096 cells",
             dedup ? "same-word lanes combined" : "one atomicOr per point  ", best * 1e3);
    }
  }
  return 0;
}
