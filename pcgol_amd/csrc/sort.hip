// sort.hip -- stable LSD radix sort of (key32, value32) pairs for gfx950, and
// the Morton ordering of query points built on it.
//
// Used by (a) the voxel filter: points sorted by dense voxel index keep their
// input order inside a voxel (stability), which is what makes the sequential
// float32 centroid sum of the reference (voxelgrid.go:157) reproducible;
// (b) the kNN / ICP paths: queries are walked in Morton order so the lanes of
// a wave traverse neighbouring sub-trees (results are written back in the
// caller's order).
//
// One pass = histogram kernel (LDS bins) -> scan kernel (per digit over the tiles) -> scatter
// kernel (which also scans the 256 digit totals).  The scatter kernel ranks a 4096-element tile with wave64 ballots
// (8 per element round: a match-any on the 8-bit digit), re-orders the tile in
// LDS by digit and writes each digit run contiguously, so global writes are
// coalesced runs rather than 4-byte scatters.
#include <stdlib.h>
#include <algorithm>
#include <string.h>

#include "voxel_key.h"

namespace pcgx {

constexpr int kRsThreads = 256;  // fixed: thread t owns digit t in the histogram / prefix steps
constexpr int kRsWaves = kRsThreads / 64;
constexpr int kRadix = 256;
static_assert(kRsThreads == kRadix, "one thread per digit");
// Elements per thread: 16 (4096-element tiles: long output runs per digit) for large inputs, 8
// (2048-element tiles: twice as many workgroups, each half as long) when 4096-element tiles would
// not even give every CU two workgroups -- a pass over 1M pairs is bound by the serial time of one
// workgroup (load, 16 ranking rounds, re-order, write), not by bandwidth.
constexpr int64_t kRsSmallInput = (int64_t)1 << 22;
inline int rs_items(int64_t n) { return n <= kRsSmallInput ? 8 : 16; }  // 4 measured no better at 1M

template <int kRsItems>
__global__ __launch_bounds__(kRsThreads) void rs_hist_kernel(const uint32_t *__restrict__ keys, int64_t n,
                                                             int shift, uint32_t *__restrict__ block_hist,
                                                             int nblocks) {
  constexpr int kRsTile = kRsThreads * kRsItems;
  __shared__ uint32_t hist[kRadix];
  hist[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kRsTile;
#pragma unroll
  for (int r = 0; r < kRsItems; r++) {
    int64_t i = base + r * kRsThreads + threadIdx.x;
    if (i < n) atomicAdd(&hist[(keys[i] >> shift) & (kRadix - 1)], 1u);
  }
  __syncthreads();
  block_hist[(int64_t)threadIdx.x * nblocks + blockIdx.x] = hist[threadIdx.x];
}

// Block b turns row b (one digit, all tiles) into its exclusive prefix and
// records the row total.
__global__ __launch_bounds__(kRsThreads) void rs_scan_rows_kernel(uint32_t *__restrict__ block_hist,
                                                                  int nblocks, uint32_t *__restrict__ totals) {
  __shared__ uint32_t wave_sum[kRsWaves];
  __shared__ uint32_t carry_s;
  uint32_t *row = block_hist + (int64_t)blockIdx.x * nblocks;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int start = 0; start < nblocks; start += kRsThreads) {
    int i = start + threadIdx.x;
    uint32_t v = i < nblocks ? row[i] : 0u;
    const uint32_t inc = wave_incl_scan_u32(v);
    if (lane == 63) wave_sum[wave] = inc;
    __syncthreads();
    uint32_t wbase = 0;
    for (int w = 0; w < wave; w++) wbase += wave_sum[w];
    uint32_t carry = carry_s;
    if (i < nblocks) row[i] = carry + wbase + inc - v;
    __syncthreads();
    if (threadIdx.x == kRsThreads - 1) carry_s = carry + wbase + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) totals[blockIdx.x] = carry_s;
}

template <int kRsItems>
__global__ __launch_bounds__(kRsThreads) void rs_scatter_kernel(
    const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in, int64_t n, int shift,
    const uint32_t *__restrict__ block_hist, int nblocks, const uint32_t *__restrict__ totals,
    uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out, int xcd_remap) {
  constexpr int kRsTile = kRsThreads * kRsItems;
  constexpr int kRsWaveChunk = kRsTile / kRsWaves;  // contiguous elements per wave
  __shared__ uint32_t cnt[kRsWaves][kRadix];
  __shared__ uint32_t tile_pref[kRadix];
  __shared__ uint32_t gbase[kRadix];
  __shared__ uint32_t wave_sum[kRsWaves];
  __shared__ uint32_t gwave_sum[kRsWaves];
  __shared__ uint32_t skeys[kRsTile];
  __shared__ uint32_t svals[kRsTile];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // XCD-contiguous placement: workgroups are dealt round-robin to the 8 XCDs (blockIdx % 8), so
  // XCD x takes the tiles [x * per, (x + 1) * per).  Consecutive tiles then write adjacent pieces
  // of every digit's output run through the SAME L2, where partial lines merge before they leave.
  const int per = (int)(gridDim.x >> 3);
  const int tile = xcd_remap ? (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  if (tile >= nblocks) return;
  const int64_t tile_base = (int64_t)tile * kRsTile;
  const int64_t wave_base = tile_base + (int64_t)wave * kRsWaveChunk;
  const uint64_t lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));

#pragma unroll
  for (int w = 0; w < kRsWaves; w++) cnt[w][threadIdx.x] = 0;
  __syncthreads();

  uint32_t key[kRsItems], val[kRsItems];
  uint32_t rank[kRsItems];
  volatile uint32_t *my_cnt = cnt[wave];
#pragma unroll
  for (int r = 0; r < kRsItems; r++) {
    const int64_t i = wave_base + r * 64 + lane;
    const bool valid = i < n;
    key[r] = valid ? keys_in[i] : 0u;
    val[r] = valid ? (vals_in ? vals_in[i] : (uint32_t)i) : 0u;  // no value array: the element's position
    const uint32_t d = (key[r] >> shift) & (kRadix - 1);
    uint64_t m = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; b++) {
      const bool bit = (d >> b) & 1u;
      const uint64_t bal = __ballot(bit);
      m &= bit ? bal : ~bal;
    }
    // m = lanes of this wave holding the same digit (valid lanes only)
    uint32_t prev = 0;
    if (valid) prev = my_cnt[d];
    rank[r] = prev + (uint32_t)__popcll(m & lt_mask);
    __builtin_amdgcn_wave_barrier();
    if (valid && (m >> lane) == 1ull) my_cnt[d] = prev + (uint32_t)__popcll(m);  // highest lane of the group
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();

  // digit t: per-wave exclusive offsets, tile total, exclusive prefix over digits; the global
  // start of digit t (exclusive prefix of the digit totals) is scanned here too, by every
  // workgroup for itself: 256 values, cheaper than a launch of its own
  {
    const int t = threadIdx.x;
    uint32_t run = 0;
#pragma unroll
    for (int w = 0; w < kRsWaves; w++) {
      uint32_t c = cnt[w][t];
      cnt[w][t] = run;
      run += c;
    }
    const uint32_t tot = totals[t];
    const uint32_t inc = wave_incl_scan_u32(run), ginc = wave_incl_scan_u32(tot);
    if (lane == 63) {
      wave_sum[wave] = inc;
      gwave_sum[wave] = ginc;
    }
    __syncthreads();
    uint32_t wbase = 0, gwbase = 0;
    for (int w = 0; w < wave; w++) {
      wbase += wave_sum[w];
      gwbase += gwave_sum[w];
    }
    const uint32_t excl = wbase + inc - run;
    tile_pref[t] = excl;
    gbase[t] = (gwbase + ginc - tot) + block_hist[(int64_t)t * nblocks + tile] - excl;  // dst = gbase[d] + pos
  }
  __syncthreads();

#pragma unroll
  for (int r = 0; r < kRsItems; r++) {
    const int64_t i = wave_base + r * 64 + lane;
    if (i < n) {
      const uint32_t d = (key[r] >> shift) & (kRadix - 1);
      const uint32_t pos = tile_pref[d] + cnt[wave][d] + rank[r];
      skeys[pos] = key[r];
      svals[pos] = val[r];
    }
  }
  __syncthreads();

  const int64_t rem = n - tile_base;
  const int count = rem < kRsTile ? (int)rem : kRsTile;
  for (int p = threadIdx.x; p < count; p += kRsThreads) {
    const uint32_t k = skeys[p];
    const uint32_t d = (k >> shift) & (kRadix - 1);
    const uint32_t dst = gbase[d] + (uint32_t)p;
    keys_out[dst] = k;
    vals_out[dst] = svals[p];
  }
}

// Measured and rejected (HISTORY.md 3.2): a single-kernel pass with decoupled look-back
// ("onesweep": global digit histograms up front, per-(tile, digit) status words walked back with
// agent-scope loads).  Correct, but 105 us per 10M-pair pass against 91 us for the four launches
// here: the pass is bound by the VALU work of the ballot ranking, and every look-back hop is a
// cross-XCD round trip (the per-XCD L2s are not coherent), so the chain costs more than the
// second read of the keys it saves; at 1M pairs (tree build, Morton presort) it was no faster either.

static bool sort_xcd_remap() {
  static const int v = [] {
    const char *e = getenv("PCGX_SORT_XCD");
    return (e && !strcmp(e, "0")) ? 0 : 1;
  }();
  return v != 0;
}

size_t radix_sort_workspace_bytes(int64_t n) {
  const int64_t tile = (int64_t)kRsThreads * rs_items(n);
  const int64_t nblocks = (n + tile - 1) / tile;
  return (size_t)(nblocks * kRadix + 2 * kRadix) * sizeof(uint32_t);
}

// Sorts pairs by key bits [0, key_bits).  keys[0]/vals[0] hold the input;
// *result is the index (0/1) of the buffers holding the output.
__global__ __launch_bounds__(256) void rs_iota_kernel(uint32_t *__restrict__ v, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) v[i] = (uint32_t)i;
}

RadixFirstHist radix_first_hist(int64_t n, void *workspace) {
  RadixFirstHist h;
  h.items = rs_items(n);
  h.nblocks = (int)((n + (int64_t)kRsThreads * h.items - 1) / ((int64_t)kRsThreads * h.items));
  h.hist = (uint32_t *)workspace;
  return h;
}

pcgx_status radix_sort_pairs(uint32_t *keys[2], uint32_t *vals[2], int64_t n, int key_bits,
                             void *workspace, int *result, hipStream_t st, bool iota_vals, bool first_hist_done) {
  *result = 0;
  if (n <= 1 || key_bits <= 0) {  // nothing to sort: the values as they are -- or as they would have been
    if (iota_vals && n > 0) {
      hipLaunchKernelGGL(rs_iota_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, vals[0], n);
      PCGX_HIP_TRY(hipGetLastError());
    }
    return PCGX_OK;
  }
  if (n > 0x7fffffffll) return fail(PCGX_E_INVALID, "radix sort: n too large");
  const int items = rs_items(n);
  const int64_t tile = (int64_t)kRsThreads * items;
  const int nblocks = (int)((n + tile - 1) / tile);
  uint32_t *block_hist = (uint32_t *)workspace;
  uint32_t *totals = block_hist + (int64_t)nblocks * kRadix;
  int cur = 0;
  for (int shift = 0; shift < key_bits; shift += 8) {
    if (shift == 0 && first_hist_done) {
      // (the kernel that made the keys counted the first digit on its way)
    } else if (items == 8)
      hipLaunchKernelGGL(rs_hist_kernel<8>, dim3(nblocks), dim3(kRsThreads), 0, st, keys[cur], n, shift, block_hist,
                         nblocks);
    else
      hipLaunchKernelGGL(rs_hist_kernel<16>, dim3(nblocks), dim3(kRsThreads), 0, st, keys[cur], n, shift, block_hist,
                         nblocks);
    hipLaunchKernelGGL(rs_scan_rows_kernel, dim3(kRadix), dim3(kRsThreads), 0, st, block_hist, nblocks,
                       totals);
    {
      ProfScope prof(PCGX_PROF_SORT_SCATTER, st);
      const int remap = sort_xcd_remap() && nblocks >= 64;
      const int grid = remap ? 8 * ((nblocks + 7) / 8) : nblocks;
      // iota_vals: the values are the positions 0 .. n-1, never stored before the first pass
      const uint32_t *vin = (iota_vals && shift == 0) ? nullptr : vals[cur];
      if (items == 8)
        hipLaunchKernelGGL(rs_scatter_kernel<8>, dim3(grid), dim3(kRsThreads), 0, st, keys[cur], vin, n, shift,
                           block_hist, nblocks, totals, keys[cur ^ 1], vals[cur ^ 1], remap);
      else
        hipLaunchKernelGGL(rs_scatter_kernel<16>, dim3(grid), dim3(kRsThreads), 0, st, keys[cur], vin, n, shift,
                           block_hist, nblocks, totals, keys[cur ^ 1], vals[cur ^ 1], remap);
    }
    cur ^= 1;
  }
  PCGX_HIP_TRY(hipGetLastError());
  *result = cur;
  return PCGX_OK;
}

// ---------------------------------------------------------------- min / max
// pc.MinMaxVec3 (pc/minmax.go:9-26): per-axis min and max with strict comparisons starting from point 0.  A NaN never
// replaces anything (its comparisons are false) but a NaN at index 0 sticks; of equal values the sequential loop keeps
// the FIRST -- and the only equal values with different bits are +0 and -0.  So the reduction is IEEE minNum / maxNum
// (a NaN operand loses: v_min_f32 / v_max_f32 behind the compiler's canonicalisation of the loaded value) over
// everything, plus, per axis, the index and sign of the FIRST zero of the cloud: when an axis' min (or max) is zero, it
// became zero at the first zero in index order (every running value before it was > 0, resp. < 0) and no later zero
// replaced it -- that zero's sign is the result's.  Four vector instructions per coordinate (canonicalise, min, max,
// compare with zero; the zero's bookkeeping runs only where a wave has seen one) where the (value, index) pairs of
// rounds 1-4 took sixteen: the pass is a stream of 12-byte records and was bound by its arithmetic (3.2 TB/s).
// Point 0 is folded in last with the reference's rule.
__device__ __forceinline__ float ld_f32_any(const uint8_t *p) {
  float v;
  __builtin_memcpy(&v, p, 4);  // records may be byte aligned (pc/iterator.go:71-76)
  return v;
}

struct MinMaxAcc {
  float mn[3], mx[3];
  uint32_t zf[3];  // (index << 1 | sign bit) of the first zero coordinate seen, 0xffffffff: none
};

__device__ __forceinline__ void minmax_init(MinMaxAcc &a) {
  const float qnan = __uint_as_float(0x7fc00000u);
  for (int k = 0; k < 3; k++) {
    a.mn[k] = qnan; a.mx[k] = qnan;  // (minNum / maxNum: the first number replaces it; all NaN: stays NaN)
    a.zf[k] = 0xffffffffu;
  }
}
__device__ __forceinline__ void mm_value(MinMaxAcc &a, int k, float v) {
  a.mn[k] = fminf(a.mn[k], v);
  a.mx[k] = fmaxf(a.mx[k], v);
}
__device__ __forceinline__ void mm_zero(MinMaxAcc &a, int k, float v, int64_t i) {
  if (v == 0.0f) a.zf[k] = min(a.zf[k], ((uint32_t)i << 1) | (__float_as_uint(v) >> 31));
}
__device__ __forceinline__ void mm_merge(MinMaxAcc &a, const MinMaxAcc &b) {
  for (int k = 0; k < 3; k++) {
    a.mn[k] = fminf(a.mn[k], b.mn[k]);
    a.mx[k] = fmaxf(a.mx[k], b.mx[k]);
    a.zf[k] = min(a.zf[k], b.zf[k]);
  }
}
// the reference's result out of the fold: a zero takes the sign of the cloud's first zero
__device__ __forceinline__ void mm_finish(MinMaxAcc &a) {
  for (int k = 0; k < 3; k++) {
    const float z = __uint_as_float((a.zf[k] & 1u) << 31);
    if (a.mn[k] == 0.0f) a.mn[k] = z;
    if (a.mx[k] == 0.0f) a.mx[k] = z;
  }
}

// What the launch's LAST workgroup does with the partials (a ticket: one returning atomic per workgroup): the fold,
// point 0's rule, out6 = {min xyz, max xyz} -- and, where the host waits for them, the six floats straight into its
// pinned mailbox, the sequence word last.  (A kernel of its own for this was 13.6 us of the filter's call: a launch
// behind a 120 MB stream, and 1024 partials read by one workgroup.)
struct MinMaxTail {
  MinMaxAcc *partials;
  unsigned int *ticket;
  const uint8_t *data;
  int32_t off, sticky_first;
  float *out6;
  volatile uint32_t *mailbox;
  uint32_t seq;
  VoxelPlanHook hook;  // the voxel filter's plan, made right behind the six floats (voxel_key.h); hook.dp == nullptr: none
};
// (a slice per workgroup of the words the filter wants cleared: 16 bytes per thread and round)
__device__ __forceinline__ void minmax_clear_slice(const MinMaxTail &T) {
  if (T.hook.zero == nullptr) return;
  const uint32_t quads = T.hook.zero_words >> 2, per = (quads + gridDim.x - 1u) / gridDim.x;
  const uint32_t q0 = blockIdx.x * per, q1 = min(q0 + per, quads);
  for (uint32_t q = q0 + threadIdx.x; q < q1; q += blockDim.x) reinterpret_cast<uint4 *>(T.hook.zero)[q] = make_uint4(0u, 0u, 0u, 0u);
}
__device__ __forceinline__ void minmax_block_fold(MinMaxAcc &a, const MinMaxTail &T) {
  __shared__ MinMaxAcc s_acc[4];
  __shared__ unsigned int s_last;
  auto wave_fold = [&](MinMaxAcc &x) {
    for (int o = 32; o > 0; o >>= 1) {
      MinMaxAcc y;
      for (int k = 0; k < 3; k++) {
        y.mn[k] = __shfl_down(x.mn[k], o);
        y.mx[k] = __shfl_down(x.mx[k], o);
        y.zf[k] = (uint32_t)__shfl_down((int)x.zf[k], o);
      }
      mm_merge(x, y);
    }
  };
  wave_fold(a);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) s_acc[wave] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; w++) mm_merge(a, s_acc[w]);
    // (write-through: the last workgroup reads the partials past its own XCD's L2)
    MinMaxAcc *dst = &T.partials[blockIdx.x];
    for (int k = 0; k < 3; k++) {
      __hip_atomic_store(&dst->mn[k], a.mn[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&dst->mx[k], a.mx[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&dst->zf[k], a.zf[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    s_last = last_workgroup_ticket(T.ticket) ? 1u : 0u;
  }
  __syncthreads();
  if (!s_last) return;  // uniform
  MinMaxAcc f;
  minmax_init(f);
  // (at most 1024 workgroups: every thread's four partials asked for at once, one round trip instead of four)
  MinMaxAcc p[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int b = (int)threadIdx.x + 256 * r;
    const MinMaxAcc *src = &T.partials[b < (int)gridDim.x ? b : 0];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      p[r].mn[k] = __hip_atomic_load(&src->mn[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      p[r].mx[k] = __hip_atomic_load(&src->mx[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      p[r].zf[k] = __hip_atomic_load(&src->zf[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; r++) mm_merge(f, p[r]);
  wave_fold(f);
  __syncthreads();  // (s_acc is read above by thread 0 only, before its ticket)
  if (lane == 0) s_acc[wave] = f;
  __syncthreads();
  if (threadIdx.x != 0) return;
  for (int w = 1; w < 4; w++) mm_merge(f, s_acc[w]);
  mm_finish(f);
  for (int k = 0; k < 3; k++) {
    // min, max := Vec3At(0): a NaN there is never replaced (minmax.go:13-23)
    // (sticky_first 0: `data` is a later slice of a cloud split over ranks -- its first point is no more
    // special than any other, a NaN there is skipped like everywhere else)
    const float p0 = ld_f32_any(T.data + T.off + 4 * k);
    if (T.sticky_first && p0 != p0) { f.mn[k] = p0; f.mx[k] = p0; }
    T.out6[k] = f.mn[k];
    T.out6[3 + k] = f.mx[k];
  }
  if (T.hook.dp) {
    const float mm6[6] = {f.mn[0], f.mn[1], f.mn[2], f.mx[0], f.mx[1], f.mx[2]};
    voxel_plan_on_device(mm6, T.hook);
  }
  if (T.mailbox) {  // the host waits for these: straight into its (pinned) memory, the sequence word last
    for (int k = 0; k < 3; k++) {
      T.mailbox[1 + k] = __float_as_uint(f.mn[k]);
      T.mailbox[4 + k] = __float_as_uint(f.mx[k]);
    }
    __threadfence_system();
    T.mailbox[0] = T.seq;
  }
}

__global__ __launch_bounds__(256) void minmax_partial_kernel(const uint8_t *__restrict__ data, int64_t n,
                                                             int32_t stride, int32_t off, MinMaxTail T) {
  MinMaxAcc a;
  minmax_init(a);
  minmax_clear_slice(T);
  // four independent loads in flight per thread
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += 4 * step) {
    float v[4][3];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int64_t i = i0 + u * step;
      const uint8_t *p = data + (i < n ? i : i0) * stride + off;
      v[u][0] = ld_f32_any(p); v[u][1] = ld_f32_any(p + 4); v[u][2] = ld_f32_any(p + 8);
    }
    bool zero = false;
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int k = 0; k < 3; k++) {  // (an index beyond n re-reads point i0: harmless for min / max, and for the zeros below)
        mm_value(a, k, v[u][k]);
        zero |= v[u][k] == 0.0f;
      }
    if (__ballot(zero) != 0ull) {  // uniform, rare: somebody's coordinate is a zero
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int64_t i = i0 + u * step < n ? i0 + u * step : i0;
        for (int k = 0; k < 3; k++) mm_zero(a, k, v[u][k], i);
      }
    }
  }
  minmax_block_fold(a, T);
}

// Packed xyz clouds (stride 12, offset 0, 16-byte aligned base) as a plain stream of 16-byte words: float j of word q is
// coordinate (q + j) mod 3 of point (4q + j) / 3.  A thread takes the words base + u T + t (u = 0, 1, 2; T threads,
// T = 1 mod 3, base a multiple of 3T), so that word u's float j has coordinate (t + u + j) mod 3: the accumulators are
// kept by (u + j) mod 3 and turned by t mod 3 once at the end.  (Ordinary loads: streaming ones -- `nt`, which leave the
// caches' dirty lines where they are instead of pushing them out, 5.5 against 2.7 TB/s in tools/micro/stream_read.cpp
// -- take 9 us off this pass and put 27 on the two behind it, which then find no cloud in the Infinity Cache.)
__global__ __launch_bounds__(256) void minmax_partial_packed_kernel(const float4 *__restrict__ data, int64_t n, MinMaxTail T) {
  MinMaxAcc a;
  minmax_init(a);
  minmax_clear_slice(T);
  const int64_t words = (3 * n) >> 2;  // whole 16-byte words; the floats behind them are taken by one thread below
  const int64_t threads = (int64_t)gridDim.x * 256, t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float mn[3], mx[3];
  for (int k = 0; k < 3; k++) { mn[k] = a.mn[k]; mx[k] = a.mx[k]; }
  for (int64_t base = 0; base < words; base += 3 * threads) {
    float4 r[3];
#pragma unroll
    for (int u = 0; u < 3; u++) {
      // (a word beyond the end: the thread's first word of the same u -- there is one, the launch sees to 3T <= words)
      const int64_t q = base + u * threads + t;
      r[u] = data[q < words ? q : u * threads + t];
    }
    bool zero = false;
#pragma unroll
    for (int u = 0; u < 3; u++) {
      const float f[4] = {r[u].x, r[u].y, r[u].z, r[u].w};
#pragma unroll
      for (int j = 0; j < 4; j++) {
        mn[(u + j) % 3] = fminf(mn[(u + j) % 3], f[j]);
        mx[(u + j) % 3] = fmaxf(mx[(u + j) % 3], f[j]);
        zero |= f[j] == 0.0f;
      }
    }
    if (__ballot(zero) != 0ull) {  // uniform, rare
#pragma unroll
      for (int u = 0; u < 3; u++) {
        const int64_t q0 = base + u * threads + t, q = q0 < words ? q0 : u * threads + t;
        const float f[4] = {r[u].x, r[u].y, r[u].z, r[u].w};
        for (int j = 0; j < 4; j++) {
          const int64_t fl = 4 * q + j;
          const int k = (int)(fl % 3);
          const uint32_t z = f[j] == 0.0f ? ((uint32_t)(fl / 3) << 1) | (__float_as_uint(f[j]) >> 31) : 0xffffffffu;
          // (no a.zf[k]: an array indexed at run time is moved to LDS, and finding one's slice there takes the workgroup's
          // size out of the dispatch packet -- host memory, 10 us before the last XCD's workgroups have it)
          a.zf[0] = min(a.zf[0], k == 0 ? z : 0xffffffffu);
          a.zf[1] = min(a.zf[1], k == 1 ? z : 0xffffffffu);
          a.zf[2] = min(a.zf[2], k == 2 ? z : 0xffffffffu);
        }
      }
    }
  }
  {
    const int turn = (int)(t % 3);  // slot s holds coordinate (s + t) mod 3
    for (int k = 0; k < 3; k++) {
      const int sl = (k + 3 - turn) % 3;
      a.mn[k] = sl == 0 ? mn[0] : sl == 1 ? mn[1] : mn[2];
      a.mx[k] = sl == 0 ? mx[0] : sl == 1 ? mx[1] : mx[2];
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {  // up to 3 floats behind the last whole word
    const float *f = reinterpret_cast<const float *>(data);
    for (int64_t fl = words << 2; fl < 3 * n; fl++) {
      MinMaxAcc one;
      minmax_init(one);
      for (int k = 0; k < 3; k++)
        if (fl % 3 == k) {
          mm_value(one, k, f[fl]);
          mm_zero(one, k, f[fl], fl / 3);
        }
      mm_merge(a, one);
    }
  }
  minmax_block_fold(a, T);
}

static pcgx_status launch_minmax_impl(const void *d_data, int64_t n, int32_t stride, int32_t off, float *d_out6,
                                      hipStream_t st, bool sticky_first, volatile uint32_t *mailbox, uint32_t seq,
                                      const VoxelPlanHook *hook = nullptr) {
  if (n <= 0) return fail(PCGX_E_NO_POINT, "no point");
  int blocks = (int)((n + 256 * 8 - 1) / (256 * 8));
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  MinMaxTail T;
  PCGX_TRY(ctx().arena.alloc_n(1024, &T.partials));
  T.ticket = ctx().tickets;  // (zero between launches: the last workgroup puts it back; one launch at a time per context)
  T.data = (const uint8_t *)d_data;
  T.off = off;
  T.sticky_first = sticky_first ? 1 : 0;
  T.out6 = d_out6;
  T.mailbox = mailbox;
  T.seq = seq;
  memset(&T.hook, 0, sizeof T.hook);
  if (hook) T.hook = *hook;
  // (the packed kernel wants a thread count that is 1 mod 3 -- 256 is -- and three words for every thread)
  int pblocks = (int)std::min<int64_t>(1024, ((3 * n) >> 2) / (3 * 256));
  pblocks -= (pblocks + 2) % 3;
  if (stride == 12 && off == 0 && (reinterpret_cast<uintptr_t>(d_data) & 15) == 0 && pblocks >= 1)
    hipLaunchKernelGGL(minmax_partial_packed_kernel, dim3(pblocks), dim3(256), 0, st, (const float4 *)d_data, n, T);
  else
    hipLaunchKernelGGL(minmax_partial_kernel, dim3(blocks), dim3(256), 0, st, (const uint8_t *)d_data, n, stride, off, T);
  PCGX_HIP_TRY(hipGetLastError());
  return PCGX_OK;
}

pcgx_status launch_minmax(const void *d_data, int64_t n, int32_t stride, int32_t off, float *d_out6,
                          hipStream_t st, bool sticky_first) {
  return launch_minmax_impl(d_data, n, stride, off, d_out6, st, sticky_first, nullptr, 0u);
}

pcgx_status launch_minmax_with_plan(const void *d_data, int64_t n, int32_t stride, int32_t off, float *d_out6,
                                    const VoxelPlanHook &hook, hipStream_t st) {
  return launch_minmax_impl(d_data, n, stride, off, d_out6, st, true, nullptr, 0u, &hook);
}

// A call's few result words on the host without a copy command and without waiting on the stream: a one-wave kernel
// behind the call's last kernel stores them into the context's pinned mailbox, the sequence word last, and the host
// polls that word.  (hipMemcpyAsync into pageable memory + hipStreamSynchronize: a staging copy, a blit kernel and two
// waits -- 20 us behind the voxel filter's last kernel.)  The kernel boundary in front of the one wave is what makes
// the call's output complete when the word arrives.
__global__ __launch_bounds__(64) void read_back_kernel(const uint32_t *__restrict__ src, int words, volatile uint32_t *mailbox, uint32_t seq) {
  for (int i = threadIdx.x; i < words; i += 64) mailbox[2 + i] = src[i];
  __threadfence_system();
  __builtin_amdgcn_s_barrier();  // (one wave: every lane's stores are out before lane 0's)
  if (threadIdx.x == 0) mailbox[0] = seq;
}

// the host's side of the mailbox: the words of sequence number `seq` (mailbox_next_seq) into host_dst
uint32_t mailbox_next_seq() {
  Context &c = ctx();
  return ++c.mailbox_seq ? c.mailbox_seq : ++c.mailbox_seq;  // (never 0: the mailbox starts zeroed)
}
pcgx_status mailbox_wait(uint32_t seq, size_t bytes, void *host_dst, hipStream_t st) {
  volatile uint32_t *mb = ctx().mailbox;
  for (long spins = 0;; spins++) {
    if (__atomic_load_n((const uint32_t *)mb, __ATOMIC_ACQUIRE) == seq) break;
    if ((spins & 0xfffff) == 0xfffff && hipStreamQuery(st) != hipErrorNotReady) {  // finished (or failed) without the word?
      if (__atomic_load_n((const uint32_t *)mb, __ATOMIC_ACQUIRE) == seq) break;
      PCGX_HIP_TRY(hipStreamSynchronize(st));
      if (__atomic_load_n((const uint32_t *)mb, __ATOMIC_ACQUIRE) == seq) break;
      return fail(PCGX_E_HIP, "the result words did not arrive in the mailbox");
    }
    __builtin_ia32_pause();
  }
  memcpy(host_dst, (const void *)(mb + 2), bytes);
  return PCGX_OK;
}

pcgx_status mailbox_wait_tagged(uint32_t seq, int words, uint32_t *host_dst, hipStream_t st) {
  const unsigned long long *mb = reinterpret_cast<const unsigned long long *>(const_cast<const uint32_t *>(ctx().mailbox) + 2);
  auto all_in = [&]() {
    for (int k = words - 1; k >= 0; k--)
      if ((uint32_t)(__atomic_load_n(&mb[k], __ATOMIC_ACQUIRE) >> 32) != seq) return false;
    return true;
  };
  for (long spins = 0;; spins++) {
    if (all_in()) break;
    if ((spins & 0xfffff) == 0xfffff && hipStreamQuery(st) != hipErrorNotReady) {  // finished (or failed) without the words?
      if (all_in()) break;
      PCGX_HIP_TRY(hipStreamSynchronize(st));
      if (all_in()) break;
      return fail(PCGX_E_HIP, "the result words did not arrive in the mailbox");
    }
    __builtin_ia32_pause();
  }
  for (int k = 0; k < words; k++) host_dst[k] = (uint32_t)__atomic_load_n(&mb[k], __ATOMIC_RELAXED);
  return PCGX_OK;
}

pcgx_status read_back_small(const void *d_src, size_t bytes, void *host_dst, hipStream_t st) {
  Context &c = ctx();
  volatile uint32_t *mb = c.mailbox;
  if (mb && bytes % 4 == 0 && bytes + 8 <= kMailboxBytes) {
    const uint32_t seq = mailbox_next_seq();
    hipLaunchKernelGGL(read_back_kernel, dim3(1), dim3(64), 0, st, (const uint32_t *)d_src, (int)(bytes / 4), mb, seq);
    PCGX_HIP_TRY(hipGetLastError());
    return mailbox_wait(seq, bytes, host_dst, st);
  }
  PCGX_HIP_TRY(hipMemcpyAsync(host_dst, d_src, bytes, hipMemcpyDeviceToHost, st));
  PCGX_HIP_TRY(hipStreamSynchronize(st));
  return PCGX_OK;
}

pcgx_status minmax_to_host(const void *d_data, int64_t n, int32_t stride, int32_t off, float *d_out6, float out6[6],
                           hipStream_t st, bool sticky_first) {
  Context &c = ctx();
  volatile uint32_t *mb = c.mailbox;
  if (mb) {
    const uint32_t seq = ++c.mailbox_seq ? c.mailbox_seq : ++c.mailbox_seq;  // (never 0: the mailbox starts zeroed)
    PCGX_TRY(launch_minmax_impl(d_data, n, stride, off, d_out6, st, sticky_first, mb, seq));
    // the kernels are a few tens of microseconds; a device in trouble is found by the stream instead
    for (long spins = 0;; spins++) {
      if (__atomic_load_n((const uint32_t *)mb, __ATOMIC_ACQUIRE) == seq) {
        for (int k = 0; k < 6; k++) {
          const uint32_t bits = mb[1 + k];
          memcpy(&out6[k], &bits, 4);
        }
        return PCGX_OK;
      }
      if ((spins & 0xfffff) == 0xfffff && hipStreamQuery(st) != hipErrorNotReady) break;  // finished (or failed) without the word
      __builtin_ia32_pause();
    }
    if (__atomic_load_n((const uint32_t *)mb, __ATOMIC_ACQUIRE) == seq) {
      for (int k = 0; k < 6; k++) {
        const uint32_t bits = mb[1 + k];
        memcpy(&out6[k], &bits, 4);
      }
      return PCGX_OK;
    }
  } else {
    PCGX_TRY(launch_minmax_impl(d_data, n, stride, off, d_out6, st, sticky_first, nullptr, 0u));
  }
  PCGX_HIP_TRY(hipMemcpyAsync(out6, d_out6, 6 * sizeof(float), hipMemcpyDeviceToHost, st));
  PCGX_HIP_TRY(hipStreamSynchronize(st));
  return PCGX_OK;
}

// ------------------------------------------------------------------ Morton
__device__ __forceinline__ uint32_t spread3(uint32_t v) {  // 10 bits -> every third bit
  v &= 0x3ffu;
  v = (v | (v << 16)) & 0x030000ffu;
  v = (v | (v << 8)) & 0x0300f00fu;
  v = (v | (v << 4)) & 0x030c30c3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}

struct MortonBox {
  float lo[3];
  float scale[3];  // cells per metre at `bits` bits per axis
  int bits;
};

__global__ __launch_bounds__(256) void morton_key_kernel(const float *__restrict__ q, int64_t n, MortonBox box,
                                                         uint32_t *__restrict__ keys) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float cmax = (float)((1u << box.bits) - 1u);
  uint32_t c[3];
  for (int k = 0; k < 3; k++) {
    float f = (q[3 * i + k] - box.lo[k]) * box.scale[k];
    f = fminf(fmaxf(f, 0.0f), cmax);  // points outside the box clamp to its faces; NaN -> 0
    c[k] = (uint32_t)f;
  }
  keys[i] = spread3(c[0]) | (spread3(c[1]) << 1) | (spread3(c[2]) << 2);
}

// Queries only need to be ordered coarsely (lanes of a wave should walk neighbouring sub-trees):
// 16-bit keys (2 radix passes) over the bounding box of the base cloud, which the tree already
// knows -- no min/max pass over the queries.  Axis bits 6/5/5 (x gets the spare bit).
constexpr int kMortonBitsPerAxis = 5;

// perm[pos] = index of the query visited at position pos (a permutation of 0..n-1).
// lo/hi: box the keys are taken over (the base cloud's bounding box).
pcgx_status morton_order(const float *d_q, int64_t n, const float lo[3], const float hi[3], int32_t *d_perm,
                         hipStream_t st) {
  if (n > 0x7fffffffll) return fail(PCGX_E_INVALID, "morton_order: n too large");
  Arena &ar = ctx().arena;
  const size_t nb = (size_t)n * 4;
  uint32_t *keys[2] = {nullptr, nullptr};
  uint32_t *vals[2] = {(uint32_t *)d_perm, nullptr};
  void *wsp = nullptr;
  PCGX_TRY(ar.alloc_n((size_t)n, &keys[0]));
  PCGX_TRY(ar.alloc_n((size_t)n, &keys[1]));
  PCGX_TRY(ar.alloc_n((size_t)n, &vals[1]));
  PCGX_TRY(ar.alloc(radix_sort_workspace_bytes(n), &wsp));
  MortonBox box;
  static const int bits_env = [] { const char *e = getenv("PCGX_MORTON_BITS"); return e ? atoi(e) : 0; }();
  const int bits_per_axis = (bits_env >= 1 && bits_env <= 10) ? bits_env : kMortonBitsPerAxis;
  box.bits = bits_per_axis;
  for (int k = 0; k < 3; k++) {
    const float ext = hi[k] - lo[k];
    box.lo[k] = lo[k] == lo[k] ? lo[k] : 0.0f;
    box.scale[k] = (ext > 0.0f && ext < 3.0e38f) ? (float)(1u << box.bits) / ext : 0.0f;
  }
  hipLaunchKernelGGL(morton_key_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_q, n, box, keys[0]);
  int res = 0;
  PCGX_TRY(radix_sort_pairs(keys, vals, n, bits_per_axis == kMortonBitsPerAxis ? 3 * kMortonBitsPerAxis + 1 : 3 * bits_per_axis, wsp, &res, st, true));
  if (res != 0) PCGX_HIP_TRY(hipMemcpyAsync(d_perm, vals[res], nb, hipMemcpyDeviceToDevice, st));
  return PCGX_OK;
}

}  // namespace pcgx
