// Microbenchmark: a kernel whose tail is ONE wave (a last-workgroup fold, a serial plan) leaves seven XCDs idle; how long
// after that do their workgroups of the NEXT kernel begin?  hold(T): one wave busy for T us (optionally with one more
// sleeping wave on every XCD), then a 1024-workgroup kernel that stamps its workgroups' entry times.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/xcd_wake.cpp -o tools/micro/xcd_wake.bin && tools/micro/xcd_wake.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>

__device__ unsigned long long g_in[1024], g_hold_end;
__device__ unsigned int g_xcd[1024];

__global__ void hold(unsigned long long ticks, int sleepers_spin) {
  // block 0 is the "tail"; the others (if any) keep their XCD awake until it is through, sleeping between looks at the clock
  const unsigned long long t0 = wall_clock64();
  if (blockIdx.x != 0 && !sleepers_spin) return;
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
  if (blockIdx.x == 0) g_hold_end = wall_clock64();
}
__global__ __launch_bounds__(256) void next_kernel() {
  if (threadIdx.x == 0) {
    g_in[blockIdx.x] = wall_clock64();
    unsigned int xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g_xcd[blockIdx.x] = xcc & 15u;
  }
}

int main() {
  const int us[] = {0, 2, 5, 10, 20, 50, 200};
  for (int mode = 0; mode < 3; mode++)
    for (int t : us) {
      double first = 0, last = 0, per[8] = {};
      for (int rep = 0; rep < 6; rep++) {
        hipDeviceSynchronize();
        // mode 0: the tail alone; 1: 8 workgroups launched, seven return at once; 2: seven stay, asleep, until the tail is through
        hipLaunchKernelGGL(hold, dim3(mode == 0 ? 1 : 8), dim3(64), 0, 0, (unsigned long long)t * 100ull, mode == 2 ? 1 : 0);
        hipLaunchKernelGGL(next_kernel, dim3(1024), dim3(256), 0, 0);
        hipDeviceSynchronize();
        static unsigned long long hi[1024], he;
        static unsigned int hx[1024];
        hipMemcpyFromSymbol(hi, HIP_SYMBOL(g_in), sizeof hi);
        hipMemcpyFromSymbol(hx, HIP_SYMBOL(g_xcd), sizeof hx);
        hipMemcpyFromSymbol(&he, HIP_SYMBOL(g_hold_end), sizeof he);
        if (rep) {
          first += ((double)*std::min_element(hi, hi + 1024) - (double)he) / 100.0;
          last += ((double)*std::max_element(hi, hi + 1024) - (double)he) / 100.0;
          for (int x = 0; x < 8; x++) {
            unsigned long long m = ~0ull;
            for (int b = 0; b < 1024; b++) if (hx[b] == (unsigned)x) m = std::min(m, hi[b]);
            per[x] += ((double)m - (double)he) / 100.0;
          }
        }
      }
      printf("%-34s tail %3d us: next kernel's first workgroup in %5.1f us after it, last %5.1f;  by XCD:", mode == 0 ? "one wave" : mode == 1 ? "one wave, seven that return" : "one wave, seven asleep beside it", t, first / 5, last / 5);
      for (int x = 0; x < 8; x++) printf(" %5.1f", per[x] / 5);
      printf("\n");
    }
  return 0;
}
