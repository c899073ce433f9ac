// wg_stamps.h -- wall clock stamps per workgroup and phase, for measurements (off unless the library is built with
// -DPCGX_STAMPS: tools/mk_variant.sh stamps "-DPCGX_STAMPS", then tools/stamps.py with PCGX_LIB).
//
// A kernel's duration says nothing about what its workgroups wait for.  wall_clock64() (100 MHz, one counter for the
// whole device) read by thread 0 at a phase's boundary and stored -- a PLAIN store, a word of its own per workgroup and
// phase -- into a __device__ array that an exported function copies out; the host folds.  Never atomics: 2048
// atomicMax on one word are a 20 us kernel of their own (10 ns each, one after the other -- tools/micro/kernel_boundary.cpp
// found its own cost that way).  What the stamps showed in round 5: a last-workgroup ticket of 1024 returning atomics
// (10 us of the min/max launch), workgroups that waited 7 us of their 13.7 for words from workgroups dispatched in the
// same microsecond (the voxel filter's bucket kernel), workgroups of different XCDs starting 3-16 us apart because the
// kernel read its workgroup size from the dispatch packet in host memory (a run-time indexed register array moved to LDS).
//
//   PCGX_STAMPS_DECLARE(name, max_workgroups, stamps_per_workgroup)   at namespace scope of the .hip file
//   PCGX_STAMP(name, stamps_per_workgroup, workgroup, k)              in the kernel (thread 0 stores)
//   int pcgx_debug_stamps_<name>(unsigned long long *out, int workgroups)   exported: synchronises, copies out
#pragma once
#include <hip/hip_runtime.h>

#if defined(PCGX_STAMPS)
#define PCGX_STAMPS_DECLARE(NAME, WGS, PER)                                                                                  \
  __device__ unsigned long long g_stamps_##NAME[(size_t)(WGS) * (PER)];                                                       \
  extern "C" __attribute__((visibility("default"))) int pcgx_debug_stamps_##NAME(unsigned long long *out, int workgroups) {   \
    if (hipDeviceSynchronize() != hipSuccess) return 1;                                                                        \
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps_##NAME), (size_t)workgroups * (PER) * 8) == hipSuccess ? 0 : 1;       \
  }
#define PCGX_STAMP(NAME, PER, WG, K)                                                                         \
  do {                                                                                                       \
    if (threadIdx.x == 0) g_stamps_##NAME[(size_t)(WG) * (PER) + (K)] = (unsigned long long)wall_clock64(); \
  } while (0)
#define PCGX_STAMP_IF(COND, NAME, PER, WG, K) \
  do {                                       \
    if (COND) PCGX_STAMP(NAME, PER, WG, K);  \
  } while (0)
/* by whichever wave's first lane gets here (the caller says which wave) */
#define PCGX_STAMP_WAVE_IF(COND, NAME, PER, WG, K)                                                                   \
  do {                                                                                                               \
    if ((COND) && (threadIdx.x & 63) == 0) g_stamps_##NAME[(size_t)(WG) * (PER) + (K)] = (unsigned long long)wall_clock64(); \
  } while (0)
#else
#define PCGX_STAMPS_DECLARE(NAME, WGS, PER)
#define PCGX_STAMP(NAME, PER, WG, K) ((void)0)
#define PCGX_STAMP_IF(COND, NAME, PER, WG, K) ((void)0)
#define PCGX_STAMP_WAVE_IF(COND, NAME, PER, WG, K) ((void)0)
#endif
