// voxel_bucket.hip -- the VoxelGrid filter's bucket path: the points travel WITH their keys.
//
// Reference: pc/filter/voxelgrid/voxelgrid.go:136-187 (filterChunk).  The radix path (voxel.hip) sorts (key, index)
// pairs and then gathers every point once more through its index: on a randomly ordered cloud that gather costs a
// 64-byte sector per 12-byte point (0.8 GB of the 1.7 GB a C3 call moves, a third of its time).  Here the
// coordinates are moved by the sort itself, and only as far as needed:
//
//   keys + histograms   one pass over the cloud: the reference's cell of every point (voxel_key.h), the first
//                       pass's tile histograms, and the population of every BUCKET (bucket = key >> s: 2^s
//                       consecutive cells, s chosen so that a bucket's points fit a workgroup's LDS)
//   1-2 scatter passes  stable LSD partition by the bucket number (digits of <= 8 bits), 2048-element tiles
//                       re-ordered in LDS so that every digit's run leaves in one piece; the elements are 16-byte
//                       records {x, y, z, key} (+ the point's index, by itself, when records carry more than xyz)
//   bucket kernel       one workgroup per bucket: its points into LDS, counting sort by the key's low s bits, the
//                       points of a cell put into input order by their position (the partition is stable), the
//                       reference's sequential float32 sum per cell (voxelgrid.go:157), centroid, output record.
//                       Where in the output a bucket's cells go is the number of occupied cells in all buckets
//                       before it: the workgroups publish their counts and wait for the earlier ones inside the
//                       launch (the exchange of strict_sum_kernel, strict.hip), so nothing is staged and compacted.
//
// What does not fit (a bucket with more points than the LDS tile holds, a cell with more than kMaxCell points, keys of
// more than 24 bits, fewer points than a launch is worth) goes the radix path: the same bytes come out either way.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>

#include "voxel_key.h"

namespace pcgx {

constexpr int kVbThreads = 256, kVbItems = 8, kVbTile = kVbThreads * kVbItems;  // scatter tiles (36 KB of LDS: four per CU)
constexpr int kVbWaves = kVbThreads / 64;
constexpr int kVbMaxCell = 255;        // points per cell it puts in order by itself
#ifndef PCGX_VB_FINAL_THREADS
#define PCGX_VB_FINAL_THREADS 256
#endif
constexpr int kVbFinalThreads = PCGX_VB_FINAL_THREADS;
constexpr long long kVbWaitTicks = 500000;  // 5 ms of s_memrealtime (100 MHz): a bucket that waits longer gives up, the radix path answers
constexpr int kVbSampleEvery = 32;     // every 32nd point is counted per bucket before anything is moved

// ---- keys, first tile histograms, a sample of the bucket populations ---------------------------------------------
// (every kVbSampleEvery-th point is counted per bucket, with global atomics: a bucket that would hold more
// than 1.5 LDS tiles by that estimate stops the attempt before anything is moved, vb_scan_rows_kernel; the exact
// check is the bucket kernel's)
__global__ __launch_bounds__(256) void vb_key_hist_kernel(const uint8_t *__restrict__ data, int64_t n, int32_t stride,
                                                          int32_t off, const VoxelDevPlan *__restrict__ dp,
                                                          uint32_t *__restrict__ key_out, uint32_t *__restrict__ block_hist,
                                                          uint32_t *__restrict__ bucket_sample, int32_t *__restrict__ err,
                                                          int hstride, const int32_t *__restrict__ flags) {
  __shared__ uint32_t dh[256];
  if (*flags) return;  // uniform: not a call for this path
  const VoxelParams vp = dp->vp;
  const VbPlan plan = dp->plan;
  dh[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t m1 = (1u << plan.d_bits[0]) - 1u;
  const bool sampled = threadIdx.x % kVbSampleEvery == 0 && bucket_sample != nullptr;
  bool any_bad = false;
  const int64_t base = (int64_t)blockIdx.x * kVbTile;
#pragma unroll
  for (int r = 0; r < kVbItems; r++) {
    const int64_t i = base + r * 256 + threadIdx.x;
    if (i < n) {
      const uint8_t *rec = data + i * stride + off;
      const float pt[3] = {ld_f32(rec), ld_f32(rec + 4), ld_f32(rec + 8)};
      uint32_t cid, ka;
      bool bad;
      const uint32_t key = voxel_key_xyz(pt, vp, cid, ka, bad);
      any_bad |= bad;
      if (key_out) key_out[i] = key;
      const uint32_t bkt = key >> plan.low_bits;
      atomicAdd(&dh[bkt & m1], 1u);
      if (sampled) atomicAdd(&bucket_sample[bkt], 1u);  // (two lanes of a wave: spread over all workgroups, no stragglers)
    }
  }
  if (any_bad) atomicOr(err, 1);
  __syncthreads();
  block_hist[(int64_t)threadIdx.x * hstride + blockIdx.x] = dh[threadIdx.x];
}

// tile histograms of the second pass's digit, from the keys as the first pass left them.  Sixteen tiles per workgroup,
// a wave each: the counts of a digit for sixteen consecutive tiles leave as ONE 64-byte piece of the digit's row (rows
// begin on 128-byte lines: hstride).  A workgroup per tile stored 256 single words, each into another row: a million
// and a quarter partial lines per call, and the kernel took 26 us for 40 MB.
constexpr int kVbHistTiles = 16;
__global__ __launch_bounds__(1024) void vb_hist2_kernel(const uint32_t *__restrict__ keys, int64_t n,
                                                        const VoxelDevPlan *__restrict__ dp, uint32_t *__restrict__ block_hist,
                                                        int ntiles, int hstride, const int32_t *__restrict__ flags) {
  __shared__ uint32_t dh[kVbHistTiles][256];
  if (*flags || dp->plan.d_bits[1] == 0) return;  // uniform
  const int shift = dp->plan.low_bits + dp->plan.d_bits[0];
  const uint32_t mask = (1u << dp->plan.d_bits[1]) - 1u;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int l = lane; l < 256; l += 64) dh[wave][l] = 0;
  const int tile = blockIdx.x * kVbHistTiles + wave;
  const int64_t base = (int64_t)tile * kVbTile;
  __builtin_amdgcn_wave_barrier();
  if (tile < ntiles) {
    uint32_t k[kVbTile / 64];
#pragma unroll
    for (int r = 0; r < kVbTile / 64; r++) {
      const int64_t i = base + r * 64 + lane;
      k[r] = i < n ? keys[i] : 0xffffffffu;
    }
#pragma unroll
    for (int r = 0; r < kVbTile / 64; r++)
      if (base + r * 64 + lane < n) atomicAdd(&dh[wave][(k[r] >> shift) & mask], 1u);
  }
  __syncthreads();
  const int j = threadIdx.x & (kVbHistTiles - 1);
  if (blockIdx.x * kVbHistTiles + j < ntiles)
    for (int d = threadIdx.x / kVbHistTiles; d < 256; d += 1024 / kVbHistTiles)
      block_hist[(int64_t)d * hstride + blockIdx.x * kVbHistTiles + j] = dh[j][d];
}

// every digit's row of tile counts -> its exclusive prefix over the tiles, and the row's total (as rs_scan_rows_kernel,
// sort.hip, with 1024 threads per row: 4882 tiles at C3 are five rounds instead of twenty)
__global__ __launch_bounds__(1024) void vb_scan_rows_kernel(uint32_t *__restrict__ block_hist, int ntiles, int hstride,
                                                            uint32_t *__restrict__ totals, const VoxelDevPlan *__restrict__ dp,
                                                            int pass, const uint32_t *__restrict__ bucket_sample,
                                                            int32_t *__restrict__ flags) {
  __shared__ uint32_t wave_sum[16];
  __shared__ uint32_t carry_s;
  __shared__ int32_t s_flags;  // (read once per workgroup: the sample's verdict below changes the word while others start)
  if (threadIdx.x == 0) s_flags = *flags;
  __syncthreads();
  if (s_flags || (pass == 1 && dp->plan.d_bits[1] == 0)) return;  // uniform
  if (bucket_sample) {  // (first pass: the sample's verdict rides along -- 256 x 1024 threads, at most 65536 buckets)
    const int b = blockIdx.x * 1024 + threadIdx.x;
    if (b < dp->plan.nbuckets && (uint64_t)bucket_sample[b] * kVbSampleEvery > (uint64_t)kVbCap * 3 / 2) atomicOr(flags, 1);
  }
  uint32_t *row = block_hist + (int64_t)blockIdx.x * hstride;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int start = 0; start < ntiles; start += 1024) {
    const int i = start + threadIdx.x;
    const uint32_t v = i < ntiles ? row[i] : 0u;
    const uint32_t inc = wave_incl_scan_u32(v);
    if (lane == 63) wave_sum[wave] = inc;
    __syncthreads();
    uint32_t wbase = 0;
    for (int w = 0; w < wave; w++) wbase += wave_sum[w];
    const uint32_t carry = carry_s;
    if (i < ntiles) row[i] = carry + wbase + inc - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = carry + wbase + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) totals[blockIdx.x] = carry_s;
}

// ---- one stable partition pass over {x, y, z} + key (+ index) -----------------------------------------------------
// kFirst: the coordinates come out of the caller's records (any stride / offset, voxel_key.h), the index is the
// point's position; else out of the previous pass's arrays.  Ranking as rs_scatter_kernel (sort.hip): wave64 ballots
// on the digit's bits, per-wave counts, the tile re-ordered in LDS so that a digit's run is written in one piece.
template <bool kFirst, bool kIdx>
__global__ __launch_bounds__(kVbThreads) void vb_scatter_kernel(
    const uint8_t *__restrict__ data, int32_t stride, int32_t off, const float4 *__restrict__ rec_in,
    const uint32_t *__restrict__ key_in, const uint32_t *__restrict__ idx_in, int64_t n, const VoxelDevPlan *__restrict__ dp,
    const uint32_t *__restrict__ block_hist, int ntiles, int hstride, const uint32_t *__restrict__ totals, float4 *__restrict__ rec_out,
    uint32_t *__restrict__ key_out, uint32_t *__restrict__ idx_out, uint32_t *__restrict__ inv_start,
    const int32_t *__restrict__ flags) {
  __shared__ uint32_t cnt[kVbWaves][256];
  __shared__ uint32_t tile_pref[256];
  __shared__ uint32_t gbase[256];
  __shared__ uint32_t wave_sum[kVbWaves];
  __shared__ uint32_t gwave_sum[kVbWaves];
  __shared__ float4 srec[kVbTile];  // {x, y, z, bits(key)}
  __shared__ uint32_t sidx[kIdx ? kVbTile : 1];
  if (*flags) return;  // uniform: the call goes the radix path
  const int shift = kFirst ? dp->plan.low_bits : dp->plan.low_bits + dp->plan.d_bits[0];
  const int dbits = dp->plan.d_bits[kFirst ? 0 : 1];
  if (dbits == 0) return;  // uniform: one pass was enough
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int per = (int)(gridDim.x >> 3);
  const int tile = (gridDim.x & 7u) == 0u ? (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3) : (int)blockIdx.x;  // XCD-contiguous (sort.hip)
  if (tile >= ntiles) return;
  const int64_t tile_base = (int64_t)tile * kVbTile;
  const int64_t wave_base = tile_base + (int64_t)wave * (kVbTile / kVbWaves);
  const uint64_t lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  const uint32_t mask = (1u << dbits) - 1u;
#pragma unroll
  for (int w = 0; w < kVbWaves; w++) cnt[w][threadIdx.x] = 0;
  __syncthreads();

  uint32_t key[kVbItems], rank[kVbItems];
  float4 rec[kVbItems];  // (every load of the tile in flight before the ranking: one round trip, not one per phase)
  volatile uint32_t *my_cnt = cnt[wave];
#pragma unroll
  for (int r = 0; r < kVbItems; r++) {
    const int64_t i = wave_base + r * 64 + lane;
    if (kFirst) {
      const uint8_t *src = data + (i < n ? i : 0) * stride + off;
      const float pt[3] = {ld_f32(src), ld_f32(src + 4), ld_f32(src + 8)};
      if (key_in) {
        key[r] = i < n ? key_in[i] : 0u;
      } else {  // the key formed again from the point (vb_key_hist_kernel counted it and did not write it)
        uint32_t cid, ka;
        bool bad;
        key[r] = voxel_key_xyz(pt, dp->vp, cid, ka, bad);
        if (i >= n) key[r] = 0u;
      }
      rec[r] = make_float4(pt[0], pt[1], pt[2], __uint_as_float(key[r]));
    } else {
      rec[r] = i < n ? rec_in[i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      key[r] = __float_as_uint(rec[r].w);
    }
  }
#pragma unroll
  for (int r = 0; r < kVbItems; r++) {
    const int64_t i = wave_base + r * 64 + lane;
    const bool valid = i < n;
    const uint32_t d = (key[r] >> shift) & mask;
    uint64_t m = __ballot(valid);
    for (int b = 0; b < dbits; b++) {  // uniform
      const bool bit = (d >> b) & 1u;
      const uint64_t bal = __ballot(bit);
      m &= bit ? bal : ~bal;
    }
    uint32_t prev = 0;
    if (valid) prev = my_cnt[d];
    rank[r] = prev + (uint32_t)__popcll(m & lt_mask);
    __builtin_amdgcn_wave_barrier();
    if (valid && (m >> lane) == 1ull) my_cnt[d] = prev + (uint32_t)__popcll(m);
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  {
    const int t = threadIdx.x;
    uint32_t run = 0;
#pragma unroll
    for (int w = 0; w < kVbWaves; w++) {
      const uint32_t c = cnt[w][t];
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
    gbase[t] = (gwbase + ginc - tot) + block_hist[(int64_t)t * hstride + tile] - excl;  // dst = gbase[d] + pos
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < kVbItems; r++) {
    const int64_t i = wave_base + r * 64 + lane;
    if (i < n) {
      const uint32_t d = (key[r] >> shift) & mask;
      const uint32_t pos = tile_pref[d] + cnt[wave][d] + rank[r];
      srec[pos] = rec[r];
      if (kIdx) sidx[pos] = kFirst ? (uint32_t)i : idx_in[i];
    }
  }
  __syncthreads();
  const int64_t rem = n - tile_base;
  const int count = rem < kVbTile ? (int)rem : kVbTile;
  // Where the buckets begin.  Behind the LAST pass the tile in LDS is grouped by bucket: this pass's digit stably, the
  // earlier digit ascending inside it (the tile came sorted by that one).  The first element of a bucket's run here
  // lands at the lowest place any of this tile's elements of the bucket get; the lowest over all tiles is the
  // bucket's start: one atomic per run (a hundred per tile), kept as the complement's maximum so that the cleared
  // word means "no points" (a kernel of its own that found the bounds in the sorted keys: 11 us per C3 call).
  const bool last_pass = inv_start != nullptr && (kFirst ? dp->plan.d_bits[1] == 0 : true);
  const int low_bits = dp->plan.low_bits;
  // The elements travel as ONE 16-byte record {x, y, z, key}: a load or store of sixteen bytes per lane costs a CU's
  // texture path twice what one of four bytes does (tools/micro/vmem_shape.cpp: 70 against 36 cycles per instruction,
  // consecutive lanes) and moves four times as much -- with x[] y[] z[] key[] as four arrays the two passes were bound
  // by that path (eight 4-byte instructions per sixty-four elements: 85 and 104 us).  Only the pass in front of the
  // second histograms writes the keys a second time, by themselves (vb_hist2_kernel reads 40 MB instead of 160).
  for (int p = threadIdx.x; p < count; p += kVbThreads) {
    const float4 e = srec[p];
    const uint32_t k = __float_as_uint(e.w);
    const uint32_t d = (k >> shift) & mask;
    const int64_t dst = (int64_t)gbase[d] + p;
    if (last_pass && (p == 0 || (__float_as_uint(srec[p - 1].w) >> low_bits) != (k >> low_bits)))
      atomicMax(&inv_start[k >> low_bits], ~(uint32_t)dst);
    rec_out[dst] = e;  // (streaming stores here: +35 us on this pass, +17 on the bucket kernel)
    if (key_out) key_out[dst] = k;
    if (kIdx) idx_out[dst] = sidx[p];
  }
}

// ---- the bucket kernel --------------------------------------------------------------------------------------------
// Where a bucket's cells go in the output = the occupied cells of all buckets before it.  Tagged words (value | 2^31),
// each written once with a write-through store and read with loads that bypass the vector L1 -- no atomics:
//   count[b]      by every bucket, as soon as it has counted its cells (nothing to wait for);
//   group_tot[g]  the 32 buckets of group g, by the group's LAST bucket once it has read the other 31 counts -- which
//                 it does anyway, for its own place;
//   super_tot[G]  the 32 groups (1024 buckets) of G, by G's last bucket, the same way from the group totals.
// A bucket that needs its place reads super_tot of the 1024-groups before its own, group_tot of the groups before
// its own inside its 1024-group and count of the buckets before it in its group: three loads per lane, a handful of
// cache lines, until every word carries its tag.  It waits for lower block indices only, which were dispatched
// before it (strict.hip, strict_sum_kernel: why that ends), and the totals it waits for depend on counts alone.
// (Measured on the way: arrival bits + a returning atomic per bucket to find out who completes a group -- two round
// trips on every workgroup's path, 50 us per C3 call; counts added into per-group 64-bit accumulators with
// fire-and-forget atomics and read by every later workgroup -- the accumulators' lines ping-pong between the adding
// and the polling XCDs, 0.25-0.6 ms of waiting per call.)
struct VbExchange {
  uint32_t *count;      // [nbuckets]
  uint32_t *group_tot;  // [ceil(nbuckets / 32)]
  uint32_t *super_tot;  // [ceil(nbuckets / 1024)] (<= 64)
};

__device__ __forceinline__ uint32_t ld_sc1_u32(const uint32_t *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool kIdx>
__global__ __launch_bounds__(kVbFinalThreads) void vb_bucket_kernel(
    const float4 *__restrict__ rec0, const float4 *__restrict__ rec1, int64_t n, const uint32_t *__restrict__ idx0,
    const uint32_t *__restrict__ idx1,
    const uint32_t *__restrict__ bucket_start, const VoxelDevPlan *__restrict__ dp, const uint8_t *__restrict__ data, int32_t stride,
    int32_t off, uint8_t *__restrict__ out, VbExchange ex, int64_t *__restrict__ total, int32_t *__restrict__ flags) {
  constexpr int kBins = 1 << kVbMaxLowBits, kWaves = kVbFinalThreads / 64;
  constexpr int kPer = kVbCap / kVbFinalThreads;       // points per thread
  constexpr int kBinsPer = kBins / kVbFinalThreads;    // cells per thread
  __shared__ float sx[kVbCap], sy[kVbCap], sz[kVbCap];
  __shared__ uint16_t order[kVbCap];    // positions cell after cell, as they arrived
  __shared__ uint32_t cnt[kBins];       // points of the cell; then: its first place in order[]
  __shared__ uint16_t ccount[kBins];    // points of the cell (kept)
  __shared__ uint16_t vrank[kBins];     // occupied cells before it in the bucket
  __shared__ uint32_t wsum[kWaves], wocc[kWaves];
  __shared__ uint32_t s_prefix;
  if (*flags & 9) return;  // uniform: not a call for this path / a bucket does not fit (an earlier kernel's finding): the radix path
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (b >= dp->plan.nbuckets) return;  // uniform (the grid is the most buckets a plan can have: the host does not know the plan)
  const VoxelParams vp = dp->vp;
  const VbPlan plan = dp->plan;
  const bool two = plan.d_bits[1] != 0;  // the arrays the last pass wrote
  const float4 *__restrict__ rec = two ? rec1 : rec0;  // {x, y, z, bits(key)} (vb_scatter_kernel)
  const uint32_t *__restrict__ idx = two ? idx1 : idx0;
  const int nbins = 1 << plan.low_bits;
  // the bucket's points: from its start (the last pass's finding, kept as the complement; a cleared word: no points) to
  // the start of the next bucket that has any
  const uint32_t inv = bucket_start[b];
  uint32_t start = 0, end = 0;
  if (inv != 0u) {  // uniform
    start = ~inv;
    for (int base = b + 1;; base += 64) {
      const int j = base + lane;
      const uint32_t v = j < plan.nbuckets ? bucket_start[j] : ~(uint32_t)n;  // (behind the last bucket: n)
      const uint64_t found = __ballot(v != 0u);
      if (found) {
        end = ~(uint32_t)__shfl((int)v, __ffsll((long long)found) - 1);
        break;
      }
    }
  }
  const bool fits = end - start <= (uint32_t)kVbCap;
  const int P = fits ? (int)(end - start) : 0;
  for (int l = threadIdx.x; l < nbins; l += kVbFinalThreads) cnt[l] = 0;
  __syncthreads();
  // ---- the bucket's points, in input order (the partition is stable); arrival order inside a cell is arbitrary
  uint16_t arr[kPer], low[kPer];
  const uint32_t lmask = (uint32_t)nbins - 1u;
#pragma unroll
  for (int r = 0; r < kPer; r++) {
    const int i = r * kVbFinalThreads + threadIdx.x;
    arr[r] = low[r] = 0;
    if (i < P) {
      const int64_t g = (int64_t)start + i;
      const float4 e = rec[g];
      const uint32_t l = __float_as_uint(e.w) & lmask;
      sx[i] = e.x;
      sy[i] = e.y;
      sz[i] = e.z;
      low[r] = (uint16_t)l;
      arr[r] = (uint16_t)atomicAdd(&cnt[l], 1u);
    }
  }
  __syncthreads();
  // ---- cells -> first place (exclusive scan of the counts) and rank among the occupied ones; thread t: cells
  // t * kBinsPer .. (consecutive, for the scan)
  uint32_t occupied_total = 0;
  bool crowded = false;
  {
    uint32_t c[kBinsPer], sum = 0, occ = 0;
#pragma unroll
    for (int k = 0; k < kBinsPer; k++) {
      const int l = threadIdx.x * kBinsPer + k;
      c[k] = l < nbins ? cnt[l] : 0u;
      sum += c[k];
      occ += c[k] ? 1u : 0u;
      crowded |= c[k] > (uint32_t)kVbMaxCell;
    }
    const uint32_t inc = wave_incl_scan_u32(sum), oinc = wave_incl_scan_u32(occ);
    if (lane == 63) {
      wsum[wave] = inc;
      wocc[wave] = oinc;
    }
    __syncthreads();
    uint32_t wb = 0, ob = 0;
    for (int w = 0; w < kWaves; w++) {
      if (w < wave) {
        wb += wsum[w];
        ob += wocc[w];
      }
      occupied_total += wocc[w];
    }
    uint32_t run = wb + inc - sum, orun = ob + oinc - occ;
#pragma unroll
    for (int k = 0; k < kBinsPer; k++) {
      const int l = threadIdx.x * kBinsPer + k;
      if (l < nbins) {
        cnt[l] = run;
        ccount[l] = (uint16_t)c[k];
        vrank[l] = (uint16_t)orun;
      }
      run += c[k];
      orun += c[k] ? 1u : 0u;
    }
  }
  const bool skip = __syncthreads_or((crowded || !fits) ? 1 : 0) != 0;  // (also the barrier behind cnt[] / ccount[])
  if (skip && threadIdx.x == 0) atomicOr(flags, fits ? 2 : 1);  // more points than the LDS tile / than a cell is ordered for: radix path
  // ---- my count out at once, nothing to wait for (the later buckets read it); my own wait comes last
  if (threadIdx.x == 0)
    __hip_atomic_store(&ex.count[b], occupied_total | 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int n2 = b >> 10, n1 = (b >> 5) & 31, n0 = b & 31;
  const bool closes_group = n0 == 31;  // (this bucket publishes its group's total, and its 1024-group's if it closes that too)
  // The totals this bucket owes the later ones go out NOW, as soon as their parts are in -- a group's total needs its
  // buckets' counts (published at this same point of their workgroups, a moment ago) and nothing else.  Published at
  // the end of the workgroup instead, every bucket's own wait met totals that were still being made (two polls
  // instead of one: 40 us per C3 call); published only behind the wait for the earlier groups, the groups chained
  // one behind the other (0.28 ms).
  if (closes_group && wave == 0) {
    bool pub_group = false, pub_super = n1 != 31;
    long long t_first = 0;
    for (int spins = 0; !(pub_group && pub_super); spins++) {
      if ((spins & 63) == 63) {
        const long long now = (long long)wall_clock64();
        if (t_first == 0) t_first = now;
        if (now - t_first > kVbWaitTicks) break;  // (the buckets behind this group then give up as well)
      }
      const uint32_t w1 = (lane < n1 && !pub_super) ? ld_sc1_u32(&ex.group_tot[(n2 << 5) + lane]) : 0x80000000u;
      const uint32_t w0 = lane < n0 ? ld_sc1_u32(&ex.count[((b >> 5) << 5) + lane]) : 0x80000000u;
      const bool ok0 = __ballot((w0 >> 31) == 0u) == 0ull, ok1 = __ballot((w1 >> 31) == 0u) == 0ull;
      uint32_t t0 = w0 & 0x7fffffffu, t1 = w1 & 0x7fffffffu;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        t0 += __shfl_xor(t0, o);
        t1 += __shfl_xor(t1, o);
      }
      if (ok0 && !pub_group) {
        if (lane == 0)
          __hip_atomic_store(&ex.group_tot[b >> 5], (t0 + occupied_total) | 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pub_group = true;
      }
      if (ok0 && ok1 && !pub_super) {
        if (lane == 0)
          __hip_atomic_store(&ex.super_tot[n2], (t1 + t0 + occupied_total) | 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pub_super = true;
      }
      if (!(pub_group && pub_super)) __builtin_amdgcn_s_sleep(4);
    }
  }
  const bool last = b == plan.nbuckets - 1;
  if ((P == 0 || skip) && !last) return;  // uniform: nothing to write (the last bucket also reports the total)
  // ---- positions cell after cell (arrival order inside a cell)
#pragma unroll
  for (int r = 0; r < kPer; r++) {
    const int i = r * kVbFinalThreads + threadIdx.x;
    if (i < P) order[cnt[low[r]] + arr[r]] = (uint16_t)i;
  }
  __syncthreads();
  // ---- cell by cell: its points into input order (ascending position: a sorting network over eight registers; a
  // cell with more points -- one in a thousand at three points per cell -- picks the next position by scanning), the
  // reference's sequential float32 sum, centroid (kept in registers until the place is known); thread t: cells t,
  // t + 256, ... (neighbouring lanes neighbouring cells: their stores are neighbours too)
  float o0[kBinsPer], o1[kBinsPer], o2[kBinsPer];
  int head[kBinsPer], cc[kBinsPer];
#pragma unroll
  for (int k = 0; k < kBinsPer; k++) {
    const int l = k * kVbFinalThreads + threadIdx.x;
    cc[k] = (l < nbins && !skip && !(plan.dbg & 2)) ? ccount[l] : 0;
    head[k] = 0;
    o0[k] = o1[k] = o2[k] = 0.0f;
    if (cc[k] == 0) continue;
    const int first = (int)cnt[l], c = cc[k];
    const uint32_t key = ((uint32_t)b << plan.low_bits) | (uint32_t)l;
    float origin[3] = {vp.vmin[0], vp.vmin[1], vp.vmin[2]};
    if (vp.chunked) chunk_origin(vp, vp.combined ? key / (uint32_t)vp.n_voxels : 0u, origin);
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
    if (c <= 8) {
      uint32_t e[8];
#pragma unroll
      for (int j = 0; j < 8; j++) e[j] = j < c ? (uint32_t)order[first + j] : 0xffffu + (uint32_t)j;  // (padding sorts behind every position)
      auto cas = [&](int x, int y) {
        const uint32_t lo = e[x] < e[y] ? e[x] : e[y], hi = e[x] < e[y] ? e[y] : e[x];
        e[x] = lo;
        e[y] = hi;
      };
      // 19 compare-exchanges (Batcher's odd-even merge sort of eight)
      cas(0, 1); cas(2, 3); cas(4, 5); cas(6, 7);
      cas(0, 2); cas(1, 3); cas(4, 6); cas(5, 7);
      cas(1, 2); cas(5, 6);
      cas(0, 4); cas(1, 5); cas(2, 6); cas(3, 7);
      cas(2, 4); cas(3, 5);
      cas(1, 2); cas(3, 4); cas(5, 6);
      float px[8], py[8], pz[8];
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int m = j < c ? (int)e[j] : (int)e[0];
        px[j] = sx[m]; py[j] = sy[m]; pz[j] = sz[m];
      }
#pragma unroll
      for (int j = 0; j < 8; j++) {
        if (j < c) {  // p := it.Vec3().Sub(vMin); v.sum = v.sum.Add(p)   (voxelgrid.go:149,157)
          s0 = s0 + (px[j] - origin[0]);
          s1 = s1 + (py[j] - origin[1]);
          s2 = s2 + (pz[j] - origin[2]);
        }
      }
      head[k] = (int)e[0];
      o0[k] = px[0]; o1[k] = py[0]; o2[k] = pz[0];
    } else {
      int last_pos = -1;
      for (int t = 0; t < c; t++) {  // the next position above `last_pos`
        int m = 0x7fffffff;
        for (int j = 0; j < c; j++) {
          const int v = order[first + j];
          m = (v > last_pos && v < m) ? v : m;
        }
        if (t == 0) head[k] = m;
        s0 = s0 + (sx[m] - origin[0]);
        s1 = s1 + (sy[m] - origin[1]);
        s2 = s2 + (sz[m] - origin[2]);
        last_pos = m;
      }
      o0[k] = sx[head[k]]; o1[k] = sy[head[k]]; o2[k] = sz[head[k]];
    }
    if (c > 1) {  // jt.SetVec3(v.sum.Mul(1.0 / float32(n)).Add(vMin))  (voxelgrid.go:178-180)
      const float inv = 1.0f / (float)c;
      o0[k] = s0 * inv + origin[0];
      o1[k] = s1 * inv + origin[1];
      o2[k] = s2 * inv + origin[2];
    }
  }
  // ---- where the bucket's cells go: the occupied cells of all buckets before it (they were dispatched before this
  // one and have published their counts long since: one round of loads as a rule)
  if (wave == 0) {
    bool gave_up = false;
    uint32_t s2 = 0, s1 = 0, s0 = 0;
    long long t_first = 0;
    for (int spins = 0; !(plan.dbg & 1); spins++) {
      // (issued before the cell phase and looked at here, the first poll was slower: 138 us against 124 for the kernel)
      const uint32_t w2 = lane < n2 ? ld_sc1_u32(&ex.super_tot[lane]) : 0x80000000u;
      const uint32_t w1 = lane < n1 ? ld_sc1_u32(&ex.group_tot[(n2 << 5) + lane]) : 0x80000000u;
      const uint32_t w0 = lane < n0 ? ld_sc1_u32(&ex.count[((b >> 5) << 5) + lane]) : 0x80000000u;
      s2 = w2 & 0x7fffffffu;
      s1 = w1 & 0x7fffffffu;
      s0 = w0 & 0x7fffffffu;
      if (__ballot(((w2 & w1 & w0) >> 31) == 0u) == 0ull) break;
      if ((spins & 63) == 63) {
        const long long now = (long long)wall_clock64();
        if (t_first == 0) t_first = now;
        if (now - t_first > kVbWaitTicks) {
          gave_up = true;
          break;
        }
      }
      __builtin_amdgcn_s_sleep(4);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      s2 += __shfl_xor(s2, o);
      s1 += __shfl_xor(s1, o);
      s0 += __shfl_xor(s0, o);
    }
    if (lane == 0) {
      const uint32_t v = s2 + s1 + s0;
      s_prefix = v;
      if (gave_up) atomicOr(flags, 4);
      if (last && !gave_up) *total = (int64_t)v + (int64_t)occupied_total;
    }
  }
  __syncthreads();
  const uint32_t prefix = s_prefix;
#pragma unroll
  for (int k = 0; k < kBinsPer; k++) {
    if (cc[k] == 0) continue;
    const int l = k * kVbFinalThreads + threadIdx.x;
    const int64_t slot = (int64_t)prefix + vrank[l];
    uint8_t *dst = out + slot * stride;
    if (!kIdx) {  // records are xyz and nothing else, 4-byte aligned
      float *d = reinterpret_cast<float *>(dst);
      d[0] = o0[k]; d[1] = o1[k]; d[2] = o2[k];  // (streaming stores: nothing, 393 against 394 us a call)
    } else {  // v.index: the first point in input order; its whole record is copied (voxelgrid.go:152-155,173-177)
      const uint8_t *src = data + (int64_t)idx[(int64_t)start + head[k]] * stride;
      if ((stride & 3) == 0 && ((reinterpret_cast<uintptr_t>(data) | reinterpret_cast<uintptr_t>(out)) & 3) == 0) {
        for (int q = 0; q < stride; q += 4) *reinterpret_cast<uint32_t *>(dst + q) = *reinterpret_cast<const uint32_t *>(src + q);
      } else {
        for (int q = 0; q < stride; q++) dst[q] = src[q];
      }
      __builtin_memcpy(dst + off, &o0[k], 4);
      __builtin_memcpy(dst + off + 4, &o1[k], 4);
      __builtin_memcpy(dst + off + 8, &o2[k], 4);
    }
  }
}

static std::atomic<long long> g_vb_taken{0}, g_vb_given_up{0}, g_vb_last_flags{0}, g_vb_last_low{0};

static int vb_knob(const char *name, int def) {
  const char *e = getenv(name);
  return e ? atoi(e) : def;
}

// The bucket path of one filter call on one GPU (voxel_key.h).  The host does not wait for the cloud's min / max:
// the plan is made on the device (vb_plan_kernel), every kernel reads it from there, buffers and grids are sized for
// the largest plan there is (the bucket kernel's grid: kVbMaxBuckets workgroups, those without a bucket return at
// once), and what came of it all -- plan, flags, the number of output points -- is read back in one copy at the end.
// Before: 34 us of a C3 call in which the GPU waited for the host (six floats back, the plan, a dozen launches).
pcgx_status voxel_bucket_filter(const void *d_data, int64_t n, int32_t stride, int32_t xyz_off, const float leaf[3],
                                const int32_t chunk[3], void *d_out, int64_t *out_n, bool *attempted, bool *taken,
                                VoxelDevPlan *dp_host, hipStream_t st) {
  *attempted = *taken = false;
  const int enabled = vb_knob("PCGX_VOXEL_BUCKET", 1), min_n = vb_knob("PCGX_VOXEL_BUCKET_MIN_N", 400000);
  if (!enabled || n < min_n) return PCGX_OK;
  *attempted = true;
  VoxelPlanHook kn;
  memset(&kn, 0, sizeof kn);
  kn.n = n;
  for (int k = 0; k < 3; k++) {
    kn.leaf[k] = leaf[k];
    kn.chunk[k] = chunk[k];
  }
  const char *force_two = getenv("PCGX_VOXEL_TWO_SORTS");  // tests: keep the two-sort path covered
  kn.force_two_sorts = force_two && force_two[0] == '1';
  kn.dbg = vb_knob("PCGX_VOXEL_BUCKET_DBG", 0);
  const int ntiles = kn.ntiles = (int32_t)((n + kVbTile - 1) / kVbTile);
  kn.grid = vb_knob("PCGX_VOXEL_BUCKET_GRID", kVbMaxBuckets);  // (measurement aid: a plan with more buckets goes the radix path)
  if (kn.grid < 1 || kn.grid > kVbMaxBuckets) kn.grid = kVbMaxBuckets;
  const bool with_idx = stride != 12 || xyz_off != 0 || ((reinterpret_cast<uintptr_t>(d_data) | reinterpret_cast<uintptr_t>(d_out)) & 3) != 0;

  Arena &ar = ctx().arena;
  uint32_t *key0 = nullptr, *key1 = nullptr, *idxb[2] = {nullptr, nullptr};  // key1: the first pass's keys again, for the second histograms
  float4 *recb[2] = {nullptr, nullptr};                                       // {x, y, z, bits(key)} behind each pass
  uint32_t *block_hist = nullptr, *totals = nullptr, *bucket_sample = nullptr, *inv_start = nullptr;
  float *d_mm6 = nullptr;
  VoxelDevPlan *d_plan = nullptr;
  int32_t *d_flags = nullptr;  // [0] flags (1 crowded bucket, 2 crowded cell, 4 the exchange gave up, 8 no plan), [1] key out of range
  int64_t *d_total = nullptr;
  VbExchange ex;
  constexpr int64_t kGroups = kVbMaxBuckets / 32, kSuper = kVbMaxBuckets / 1024;
  PCGX_TRY(ar.alloc_n((size_t)n, &key0));
  PCGX_TRY(ar.alloc_n((size_t)n, &key1));
  for (int k = 0; k < 2; k++) {
    PCGX_TRY(ar.alloc_n((size_t)n, &recb[k]));
    if (with_idx) PCGX_TRY(ar.alloc_n((size_t)n, &idxb[k]));
  }
  const int hstride = (ntiles + 31) & ~31;  // (a digit's row of tile counts begins on a 128-byte line)
  PCGX_TRY(ar.alloc_n((size_t)hstride * 256 + 32, &block_hist));
  block_hist = reinterpret_cast<uint32_t *>((reinterpret_cast<uintptr_t>(block_hist) + 127) & ~(uintptr_t)127);
  PCGX_TRY(ar.alloc_n(256, &totals));
  PCGX_TRY(ar.alloc_n(8, &d_mm6));
  // one block, zeroed at once: flags, the exchange's words, the sample of the bucket populations.  What the host
  // reads back at the end is at its start: flags, err, total, the plan
  struct Readback {
    int32_t flags, err;
    int64_t total;
    VoxelDevPlan plan;
  };
  static_assert(sizeof(Readback) % 8 == 0, "words");
  constexpr size_t kHeadWords = (sizeof(Readback) + 127) / 128 * 32;
  const size_t zero_words = kHeadWords + (size_t)kVbMaxBuckets * 3 + (size_t)kGroups + (size_t)kSuper;
  uint32_t *zero_block = nullptr;
  PCGX_TRY(ar.alloc_n(zero_words + 64, &zero_block));
  zero_block = reinterpret_cast<uint32_t *>((reinterpret_cast<uintptr_t>(zero_block) + 127) & ~(uintptr_t)127);
  Readback *d_rb = reinterpret_cast<Readback *>(zero_block);
  d_flags = &d_rb->flags;
  d_total = &d_rb->total;
  d_plan = &d_rb->plan;
  bucket_sample = zero_block + kHeadWords;
  ex.count = bucket_sample + kVbMaxBuckets;
  ex.group_tot = ex.count + kVbMaxBuckets;
  ex.super_tot = ex.group_tot + kGroups;
  inv_start = ex.super_tot + kSuper;
  // (no memset, no kernel for the plan: the min/max launch clears the words behind the head as it goes, and its last
  // workgroup writes the head -- flags, the plan -- behind the six floats)
  static_assert(kHeadWords % 4 == 0 && (kVbMaxBuckets * 3 + kGroups + kSuper) % 4 == 0, "cleared 16 bytes at a time");
  kn.dp = d_plan;
  kn.head = d_flags;
  kn.zero = zero_block + kHeadWords;
  kn.zero_words = (uint32_t)(zero_words - kHeadWords);

  const uint8_t *data = (const uint8_t *)d_data;
  PCGX_TRY(launch_minmax_with_plan(d_data, n, stride, xyz_off, d_mm6, kn, st));
  const int sample = vb_knob("PCGX_VOXEL_BUCKET_SAMPLE", 1);
  const bool rekey = vb_knob("PCGX_VOXEL_BUCKET_REKEY", 1) != 0;  // the first pass forms the keys again instead of reading an array of them
  uint32_t *key0_arg = rekey ? nullptr : key0;
  uint32_t *scatter_bounds = inv_start;
  hipLaunchKernelGGL(vb_key_hist_kernel, dim3(ntiles), dim3(256), 0, st, data, n, stride, xyz_off, (const VoxelDevPlan *)d_plan, key0_arg,
                     block_hist, sample ? bucket_sample : (uint32_t *)nullptr, d_flags + 1, hstride, (const int32_t *)d_flags);
  hipLaunchKernelGGL(vb_scan_rows_kernel, dim3(256), dim3(1024), 0, st, block_hist, ntiles, hstride, totals, (const VoxelDevPlan *)d_plan, 0,
                     sample ? (const uint32_t *)bucket_sample : (const uint32_t *)nullptr, d_flags);
  const int grid = ntiles >= 64 ? 8 * ((ntiles + 7) / 8) : ntiles;
  if (with_idx)
    hipLaunchKernelGGL((vb_scatter_kernel<true, true>), dim3(grid), dim3(kVbThreads), 0, st, data, stride, xyz_off,
                       (const float4 *)nullptr, (const uint32_t *)key0_arg, (const uint32_t *)nullptr, n, (const VoxelDevPlan *)d_plan,
                       (const uint32_t *)block_hist, ntiles, hstride, (const uint32_t *)totals, recb[0], key1, idxb[0], scatter_bounds,
                       (const int32_t *)d_flags);
  else
    hipLaunchKernelGGL((vb_scatter_kernel<true, false>), dim3(grid), dim3(kVbThreads), 0, st, data, stride, xyz_off,
                       (const float4 *)nullptr, (const uint32_t *)key0_arg, (const uint32_t *)nullptr, n, (const VoxelDevPlan *)d_plan,
                       (const uint32_t *)block_hist, ntiles, hstride, (const uint32_t *)totals, recb[0], key1, idxb[0], scatter_bounds,
                       (const int32_t *)d_flags);
  // the second digit (every plan above some ten thousand points has one; a plan without returns from these at once)
  hipLaunchKernelGGL(vb_hist2_kernel, dim3((ntiles + kVbHistTiles - 1) / kVbHistTiles), dim3(1024), 0, st, (const uint32_t *)key1, n,
                     (const VoxelDevPlan *)d_plan, block_hist, ntiles, hstride, (const int32_t *)d_flags);
  hipLaunchKernelGGL(vb_scan_rows_kernel, dim3(256), dim3(1024), 0, st, block_hist, ntiles, hstride, totals, (const VoxelDevPlan *)d_plan, 1,
                     (const uint32_t *)nullptr, d_flags);
  if (with_idx)
    hipLaunchKernelGGL((vb_scatter_kernel<false, true>), dim3(grid), dim3(kVbThreads), 0, st, data, stride, xyz_off,
                       (const float4 *)recb[0], (const uint32_t *)nullptr, (const uint32_t *)idxb[0], n, (const VoxelDevPlan *)d_plan,
                       (const uint32_t *)block_hist, ntiles, hstride, (const uint32_t *)totals, recb[1], (uint32_t *)nullptr, idxb[1], scatter_bounds,
                       (const int32_t *)d_flags);
  else
    hipLaunchKernelGGL((vb_scatter_kernel<false, false>), dim3(grid), dim3(kVbThreads), 0, st, data, stride, xyz_off,
                       (const float4 *)recb[0], (const uint32_t *)nullptr, (const uint32_t *)idxb[0], n, (const VoxelDevPlan *)d_plan,
                       (const uint32_t *)block_hist, ntiles, hstride, (const uint32_t *)totals, recb[1], (uint32_t *)nullptr, idxb[1], scatter_bounds,
                       (const int32_t *)d_flags);
  const int bucket_grid = kn.grid;
  if (with_idx)
    hipLaunchKernelGGL(vb_bucket_kernel<true>, dim3(bucket_grid), dim3(kVbFinalThreads), 0, st, (const float4 *)recb[0],
                       (const float4 *)recb[1], n, (const uint32_t *)idxb[0], (const uint32_t *)idxb[1], (const uint32_t *)inv_start, (const VoxelDevPlan *)d_plan, data, stride, xyz_off,
                       (uint8_t *)d_out, ex, d_total, d_flags);
  else
    hipLaunchKernelGGL(vb_bucket_kernel<false>, dim3(bucket_grid), dim3(kVbFinalThreads), 0, st, (const float4 *)recb[0],
                       (const float4 *)recb[1], n, (const uint32_t *)idxb[0], (const uint32_t *)idxb[1], (const uint32_t *)inv_start, (const VoxelDevPlan *)d_plan, data, stride, xyz_off,
                       (uint8_t *)d_out, ex, d_total, d_flags);
  PCGX_HIP_TRY(hipGetLastError());
  Readback h;
  PCGX_TRY(read_back_small(d_rb, sizeof h, &h, st));
  *dp_host = h.plan;
  if (h.plan.status == 1 || h.plan.status == 2) {  // the grid itself is not to be had: the host's words for it
    VoxelParams vp;
    return voxel_grid_params_or_fail(h.plan.mm6, leaf, chunk, vp);
  }
  if (h.err)
    return fail(PCGX_E_OUT_OF_RANGE, "voxel filter: a point falls outside the dense grid (the reference panics: index out of range)");
  if (h.plan.status) return PCGX_OK;  // keys too wide, two sorts, too few keys: the radix path
  g_vb_last_low = h.plan.plan.low_bits;
  if (h.flags) {  // crowded bucket / cell (or the exchange gave up): the radix path does the call
    g_vb_given_up++;
    g_vb_last_flags = h.flags;
    if (getenv("PCGX_VOXEL_BUCKET_TRACE"))
      fprintf(stderr, "pcgx voxel bucket path: flags %d (1 bucket over %d points, 2 cell over %d points, 4 exchange gave up); low bits %d, %d buckets, digits %d + %d\n",
              h.flags, kVbCap, kVbMaxCell, h.plan.plan.low_bits, h.plan.plan.nbuckets, h.plan.plan.d_bits[0], h.plan.plan.d_bits[1]);
    return PCGX_OK;
  }
  *out_n = h.total;
  *taken = true;
  g_vb_taken++;
  return PCGX_OK;
}

}  // namespace pcgx

extern "C" pcgx_status pcgx_debug_voxel_stats(int64_t out[4], int32_t reset) {
  using namespace pcgx;
  if (!out) return fail(PCGX_E_INVALID, "pcgx_debug_voxel_stats: NULL argument");
  out[0] = g_vb_taken;
  out[1] = g_vb_given_up;
  out[2] = g_vb_last_flags;
  out[3] = g_vb_last_low;
  if (reset) g_vb_taken = g_vb_given_up = g_vb_last_flags = 0;
  return PCGX_OK;
}
