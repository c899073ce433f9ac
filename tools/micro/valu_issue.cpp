// Microbenchmark: what a vector instruction of ONE wave costs on gfx950 when nothing else runs on its SIMD, and with a
// second wave of the same workgroup next to it -- the question behind the chain kernel's forward scan (840 instructions,
// ~2 us).  Streams: dependent v_add_u32; four independent v_add_u32 chains; v_cmp + v_cndmask pairs (a select whose
// mask the instruction in front of it made); DPP moves + adds (a scan step); v_med3 / v_max chains.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_issue.cpp -o tools/micro/valu_issue.bin && tools/micro/valu_issue.bin
#include <hip/hip_runtime.h>
#include <cstdio>

enum Kind { DEP_ADD, INDEP4_ADD, CMP_SELECT, DPP_ADD, MED3_MAX, kKinds };
static const int kInstrPerGroup[kKinds] = {4, 4, 16, 4, 8};  // vector instructions the compiler makes of a group (ISA read)
static const char *kNames[kKinds] = {"dependent v_add_u32", "4 independent v_add_u32", "v_cmp + v_cndmask (4 indep.)",
                                     "v_mov_dpp + v_add (4 indep.)", "v_med3 + v_max (4 indep.)"};

template <int kKind>
__global__ void stream(int n, unsigned seed, unsigned *out, long long *cycles, int timed_waves) {
  unsigned a = seed + threadIdx.x, b = a * 3u, c = a * 5u, d = a * 7u;
  const unsigned k1 = seed | 1u, k2 = seed | 2u;
  const int wave = threadIdx.x >> 6;
  const long long c0 = clock64();
  for (int i = 0; i < n; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
      // (the empty asm keeps the compiler from folding a chain of additions into one)
#define OPAQUE(x) asm volatile("" : "+v"(x))
      if (kKind == DEP_ADD) {
        a = a + k1; OPAQUE(a); a = a + k2; OPAQUE(a); a = a + k1; OPAQUE(a); a = a + k2; OPAQUE(a);
      } else if (kKind == INDEP4_ADD) {
        a = a + k1; b = b + k2; c = c + k1; d = d + k2;
        OPAQUE(a); OPAQUE(b); OPAQUE(c); OPAQUE(d);
      } else if (kKind == CMP_SELECT) {
        a = (b & 1u) ? a + 0u : k1 ^ a; b = (c & 2u) ? b : k2 ^ b; c = (d & 1u) ? c : k1 ^ c; d = (a & 2u) ? d : k2 ^ d;
      } else if (kKind == DPP_ADD) {
        a += (unsigned)__builtin_amdgcn_mov_dpp((int)a, 0x111, 0xf, 0xf, true);
        b += (unsigned)__builtin_amdgcn_mov_dpp((int)b, 0x112, 0xf, 0xf, true);
        c += (unsigned)__builtin_amdgcn_mov_dpp((int)c, 0x114, 0xf, 0xf, true);
        d += (unsigned)__builtin_amdgcn_mov_dpp((int)d, 0x118, 0xf, 0xf, true);
      } else {
        a = (unsigned)max((int)b, min(max((int)a, -(1 << 28)), 1 << 28));
        b = (unsigned)max((int)c, min(max((int)b, -(1 << 28)), 1 << 28));
        c = (unsigned)max((int)d, min(max((int)c, -(1 << 28)), 1 << 28));
        d = (unsigned)max((int)a, min(max((int)d, -(1 << 28)), 1 << 28));
      }
      if (kKind >= CMP_SELECT) { OPAQUE(a); OPAQUE(b); OPAQUE(c); OPAQUE(d); }
    }
  }
  const long long c1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0 && wave < timed_waves) cycles[wave] = c1 - c0;
}

template <int kKind>
void run(int threads, unsigned *d_out, long long *d_cyc) {
  const int n = 2048;
  hipLaunchKernelGGL((stream<kKind>), dim3(1), dim3(threads), 0, 0, n, 12345u, d_out, d_cyc, 8);
  hipDeviceSynchronize();
  long long cyc[8] = {0};
  hipMemcpy(cyc, d_cyc, sizeof cyc, hipMemcpyDeviceToHost);
  const double groups = (double)n * 16;
  const double per = 1.0 / groups / kInstrPerGroup[kKind];  // clock64() is s_memtime: shader cycles
  printf("%-32s %d waves in the workgroup (%d per SIMD): wave 0 %.2f cycles per instruction", kNames[kKind], threads / 64,
         (threads / 64 + 3) / 4, (double)cyc[0] * per);
  if (threads > 256) printf(", wave 4 %.2f", (double)cyc[4] * per);
  printf("\n");
}

int main() {
  unsigned *d_out;
  long long *d_cyc;
  hipMalloc(&d_out, 1 << 16);
  hipMalloc(&d_cyc, 64);
  for (int threads : {64, 256, 512}) {
    run<DEP_ADD>(threads, d_out, d_cyc);
    run<INDEP4_ADD>(threads, d_out, d_cyc);
    run<CMP_SELECT>(threads, d_out, d_cyc);
    run<DPP_ADD>(threads, d_out, d_cyc);
    run<MED3_MAX>(threads, d_out, d_cyc);
  }
  printf("(shader cycles from s_memtime; wave 0 is the older wave of its SIMD, wave 4 the younger one)\n");
  return 0;
}
