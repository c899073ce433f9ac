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
//   bucket kernel       the device's workgroups walk the buckets: a bucket's points sorted by the key's low s bits in
//                       LDS -- stably, so that a cell's points lie in input order (the partition is stable too) -- the
//                       reference's sequential float32 sum per cell (voxelgrid.go:157), centroid; the cell's result is
//                       left at the bucket's first point + the cell's rank among the bucket's occupied cells
//   placing kernel      where in the output a bucket's cells go is the number of occupied cells in all buckets before
//                       it: every workgroup adds up the counts before its sixteen buckets and copies their cells
//
// What does not fit (a bucket with more points than the LDS tile holds, keys of more than 26 bits, fewer points than a
// launch is worth) goes the radix path: the same bytes come out either way.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>

#include "voxel_key.h"
#include "wg_stamps.h"

namespace pcgx {

constexpr int kVbThreads = 256, kVbItems = 8, kVbTile = kVbThreads * kVbItems;  // scatter tiles (36 KB of LDS: four per CU)
constexpr int kVbWaves = kVbThreads / 64;
#ifndef PCGX_VB_FINAL_THREADS
#define PCGX_VB_FINAL_THREADS 256
#endif
constexpr int kVbFinalThreads = PCGX_VB_FINAL_THREADS;
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

// every digit's row of tile counts -> its exclusive prefix over the tiles, and the row's total (1024 threads per row)
__global__ __launch_bounds__(1024) void vb_scan_rows_kernel(uint32_t *__restrict__ block_hist, int ntiles, int hstride,
                                                            uint32_t *__restrict__ totals, const VoxelDevPlan *__restrict__ dp,
                                                            int pass, const uint32_t *__restrict__ bucket_sample,
                                                            int32_t *__restrict__ flags) {
  __shared__ uint32_t wave_sum[16];
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
  // a run of consecutive tiles per thread, ONE scan over the threads (4883 tiles at C3: five a thread; 1024 tiles at a
  // time, a scan and three barriers each, the kernel took 7.5 us)
  constexpr int kKeep = 8;  // (a run's counts stay in registers up to this length; longer runs are read twice)
  const int per = (ntiles + 1023) / 1024, t0 = (int)threadIdx.x * per, t1 = min(t0 + per, ntiles);
  uint32_t v[kKeep], sum = 0;
#pragma unroll
  for (int j = 0; j < kKeep; j++) {
    v[j] = t0 + j < t1 ? row[t0 + j] : 0u;
    sum += v[j];
  }
  for (int i = t0 + kKeep; i < t1; i++) sum += row[i];
  const uint32_t inc = wave_incl_scan_u32(sum);
  if (lane == 63) wave_sum[wave] = inc;
  __syncthreads();
  uint32_t run = inc - sum, all = 0;
  for (int w = 0; w < 16; w++) {
    if (w < wave) run += wave_sum[w];
    all += wave_sum[w];
  }
#pragma unroll
  for (int j = 0; j < kKeep; j++) {
    if (t0 + j < t1) row[t0 + j] = run;
    run += v[j];
  }
  for (int i = t0 + kKeep; i < t1; i++) {
    const uint32_t c = row[i];
    row[i] = run;
    run += c;
  }
  if (threadIdx.x == 0) totals[blockIdx.x] = all;
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

// ---- the bucket kernel and the placing kernel -----------------------------------------------------------------------
// Where a bucket's cells go in the output = the occupied cells of all buckets before it -- which no bucket knows
// while it works.  Two kernels: the bucket kernel does everything but the placing -- a cell's result {x, y, z, the
// first point's index} goes to cells[bucket's first point + the cell's rank among the bucket's occupied ones], an array
// the passes are through with, and the bucket's count of occupied cells into count[b] and, with fire-and-forget
// atomics, into its group's (32 buckets) and its 1024-group's totals; the placing kernel, a wave per bucket, adds up
// what is before its bucket (three loads, final values: nothing to wait for) and copies the bucket's cells to their
// place.  38 MB written and read once more at C3, 20 us.
//
// One kernel with the buckets WAITING for their place took 118 us.  Workgroup by workgroup (wall clock stamps, C3, 13010
// buckets): 6.5 us of work -- bounds 0.8, points into LDS 2.3, cells scanned 0.9, order[] 0.4, cell phase 1.8 --
// and 7 us of waiting with 26 KB of LDS held, ten looks at words that were not there yet: a count is out 4.3 us into
// its workgroup's life, visible to another XCD 2 us later, and the buckets right before a bucket were dispatched
// within the same microsecond (up to 3 us AFTER it on another XCD); a look (agent-scope loads, past the L2) takes 3 us
// under the kernel's traffic.  Reading the last 14 groups' counts directly instead of their totals (which their last
// bucket publishes another poll later): 4.4 us of waiting, 2 looks, 113 us.  (Earlier forms of the exchange, all
// slower: arrival bits + a returning atomic per bucket; per-group accumulators polled by every later workgroup; totals
// published at the end of a workgroup or behind its own wait.)

// the bucket's points: from its start (the last pass's finding, kept as the complement; a cleared word: no points) to
// the start of the next bucket that has any.  One wave's worth of words at once: the bucket's own and the 63 behind it.
__device__ __forceinline__ void vb_bucket_bounds(const uint32_t *__restrict__ bucket_start, int b, int nbuckets, int64_t n, int lane,
                                                 uint32_t &start, uint32_t &end) {
  start = end = 0;
  for (int base = b;; base += 64) {
    const int j = base + lane;
    uint32_t v = j < nbuckets ? bucket_start[j] : ~(uint32_t)n;  // (behind the last bucket: n)
    if (base == b) {
      const uint32_t own = (uint32_t)__shfl((int)v, 0);
      if (own == 0u) return;  // uniform: no points
      start = ~own;
      if (lane == 0) v = 0u;
    }
    const uint64_t found = __ballot(v != 0u);
    if (found) {
      end = ~(uint32_t)__shfl((int)v, __ffsll((long long)found) - 1);
      return;
    }
  }
}

// (measurements only, -DPCGX_STAMPS: a bucket's phases by the wall clock -- tools/stamps.py vb_bucket)
PCGX_STAMPS_DECLARE(vb_bucket, kVbMaxBuckets, 8)
#ifndef PCGX_VB_MIN_WAVES
#define PCGX_VB_MIN_WAVES 5
#endif
template <bool kIdx>
__global__ __launch_bounds__(kVbFinalThreads, PCGX_VB_MIN_WAVES) void vb_bucket_kernel(
    const float4 *__restrict__ rec0, const float4 *__restrict__ rec1, int64_t n, const uint32_t *__restrict__ idx0,
    const uint32_t *__restrict__ idx1, const uint32_t *__restrict__ bucket_start, const VoxelDevPlan *__restrict__ dp,
    float4 *cells0, float4 *cells1, uint32_t *__restrict__ count, int32_t *__restrict__ flags) {
  constexpr int kBins = 1 << kVbMaxLowBits, kWaves = kVbFinalThreads / 64;
  constexpr int kRounds = kVbCap / kVbFinalThreads;    // points per thread: a wave takes a quarter of the bucket, 64 at a time
  constexpr int kBinsPer = kBins / kVbFinalThreads;    // cells per thread
  static_assert(kVbCap % kVbFinalThreads == 0 && kVbCap < 65536, "whole rounds; places are 16 bits");
  __shared__ float sx[kVbCap], sy[kVbCap], sz[kVbCap];   // the bucket's points, cell after cell, input order inside a cell
  __shared__ uint16_t sfrom[kIdx ? kVbCap : 1];          // (records with more than xyz: the point's place in the bucket's input order)
  // (rows of kBins + 4: the word behind a row's last cell is its end marker, and the rows begin on 8-byte words -- a
  // thread's four consecutive cells are one 8-byte access.  The kernel is bound by what a CU's LDS pipe and VALUs get
  // through, a bucket every 1.3 us per CU: 16-bit accesses cost what wider ones do)
  constexpr int kRow = kBins + 4;
  __shared__ __attribute__((aligned(16))) uint16_t cnt[kWaves][kRow];  // a wave's points of the cell; then: where they begin in sx / sy / sz
  __shared__ __attribute__((aligned(8))) uint16_t vrank[kBins];         // occupied cells before it in the bucket
  static_assert(kBinsPer == 4 && (kWaves * kRow * 2) % 16 == 0, "four cells per thread: 8-byte words; cleared 16 bytes at a time");
  __shared__ uint32_t wsum[kWaves], wocc[kWaves];
  if (*flags & 9) return;  // uniform: not a call for this path / a bucket does not fit (an earlier kernel's finding): the radix path
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const VoxelParams vp = dp->vp;
  const VbPlan plan = dp->plan;
  const bool two = plan.d_bits[1] != 0;  // the arrays the last pass wrote
  const float4 *__restrict__ rec = two ? rec1 : rec0;  // {x, y, z, bits(key)} (vb_scatter_kernel)
  const uint32_t *__restrict__ idx = two ? idx1 : idx0;
  float4 *cells = two ? cells0 : cells1;  // (the array the last pass did not write)
  const int nbins = 1 << plan.low_bits;
  const uint32_t lmask = (uint32_t)nbins - 1u;
  const uint64_t lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  // (the grid is what fits the device at once, whatever the plan: the host does not know the plan)
  for (int b = blockIdx.x; b < plan.nbuckets; b += gridDim.x) {
    PCGX_STAMP(vb_bucket, 8, b, 0);
    uint32_t start, end;
    vb_bucket_bounds(bucket_start, b, plan.nbuckets, n, lane, start, end);
    PCGX_STAMP(vb_bucket, 8, b, 1);  // bounds
    if (end - start > (uint32_t)kVbCap) {  // uniform: more points than the LDS tile -- the radix path does the call
      if (threadIdx.x == 0) atomicOr(flags, 1);
      return;
    }
    const int P = (int)(end - start);
    if (P == 0) continue;  // uniform: no points, no cells, nothing to count (the words are cleared)
    // ---- the bucket's points: wave w takes the w-th quarter, 64 at a time -- (wave, round, lane) is input order (the
    // partition passes are stable)
    const int quarter = (P + kWaves - 1) / kWaves, w_begin = wave * quarter, w_end = min(P, w_begin + quarter);
    float4 e[kRounds];
#pragma unroll
    for (int r = 0; r < kRounds; r++) {
      const int i = w_begin + r * 64 + lane;
      e[r] = i < w_end ? rec[(int64_t)start + i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    for (int l = threadIdx.x; l < kWaves * kRow * 2 / 16; l += kVbFinalThreads) reinterpret_cast<uint4 *>(&cnt[0][0])[l] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    PCGX_STAMP(vb_bucket, 8, b, 2);  // points asked for, counts cleared
    // ---- a point's rank among its wave's points of the same cell, in input order: the lanes of a round with the same
    // cell by ballots over the cell number's bits (vb_scatter_kernel's ranking), the earlier rounds by the wave's count
    uint16_t rank[kRounds];
    {
      // (an LDS pointer by type: as a generic one the compiler (ROCm 7.2) tests it for the LDS aperture with an
      // instruction it then fails to select)
      typedef __attribute__((address_space(3))) volatile uint16_t lds_vu16;
      lds_vu16 *my_cnt = (lds_vu16 *)cnt[wave];
      // (all kVbMaxLowBits bits, whatever the plan's: a bit above them is zero in every lane and leaves the mask as it
      // is.  As a loop over the plan's bits it was eight instructions and a branch per bit, 2 us of a bucket's 6; all
      // rounds' masks first, fifty independent steps, took 114 registers and a workgroup per CU with them)
#pragma unroll
      for (int r = 0; r < kRounds; r++) {
        const bool valid = w_begin + r * 64 + lane < w_end;
        const uint32_t d = __float_as_uint(e[r].w) & lmask;
        const uint64_t v = __ballot(valid);
        if (v == 0ull) {  // uniform
          rank[r] = 0;
          continue;
        }
        uint32_t m_lo = (uint32_t)v, m_hi = (uint32_t)(v >> 32);
#pragma unroll
        for (int bit = 0; bit < kVbMaxLowBits; bit++) {
          const uint32_t sel = (uint32_t)(((int32_t)(d << (31 - bit))) >> 31);  // all ones: my bit is set
          const uint64_t bal = __ballot(sel != 0u);
          m_lo &= ~((uint32_t)bal ^ sel);        // the lanes whose bit is mine
          m_hi &= ~((uint32_t)(bal >> 32) ^ sel);
        }
        const uint64_t m = (uint64_t)m_hi << 32 | m_lo;
        uint32_t prev = 0;
        if (valid) prev = my_cnt[d];
        rank[r] = (uint16_t)(prev + (uint32_t)__popcll(m & lt_mask));
        __builtin_amdgcn_wave_barrier();
        if (valid && (m >> lane) == 1ull) my_cnt[d] = (uint16_t)(prev + (uint32_t)__popcll(m));
        __builtin_amdgcn_wave_barrier();
      }
    }
    __syncthreads();
    PCGX_STAMP(vb_bucket, 8, b, 3);  // ranks
    // ---- cells -> where their points begin (exclusive scan of the counts, wave after wave inside a cell) and their rank
    // among the occupied ones; thread t: cells t * kBinsPer .. (consecutive, for the scan)
    uint32_t occupied_total = 0;
    {
      uint32_t c[kBinsPer][kWaves], sum = 0, occ = 0;
      const bool mine = (int)threadIdx.x * kBinsPer < nbins;  // (nbins is a power of two: all four cells or none)
#pragma unroll
      for (int w = 0; w < kWaves; w++) {
        const uint2 q = mine ? *reinterpret_cast<const uint2 *>(&cnt[w][threadIdx.x * kBinsPer]) : make_uint2(0u, 0u);
        c[0][w] = q.x & 0xffffu; c[1][w] = q.x >> 16; c[2][w] = q.y & 0xffffu; c[3][w] = q.y >> 16;
      }
#pragma unroll
      for (int k = 0; k < kBinsPer; k++) {
        uint32_t t = 0;
#pragma unroll
        for (int w = 0; w < kWaves; w++) t += c[k][w];
        sum += t;
        occ += t ? 1u : 0u;
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
      uint32_t first[kBinsPer][kWaves], vr[kBinsPer];
#pragma unroll
      for (int k = 0; k < kBinsPer; k++) {
        uint32_t t = 0;
#pragma unroll
        for (int w = 0; w < kWaves; w++) {
          first[k][w] = run + t;
          t += c[k][w];
        }
        vr[k] = orun;
        run += t;
        orun += t ? 1u : 0u;
      }
      if (mine) {
#pragma unroll
        for (int w = 0; w < kWaves; w++)
          *reinterpret_cast<uint2 *>(&cnt[w][threadIdx.x * kBinsPer]) = make_uint2(first[0][w] | first[1][w] << 16, first[2][w] | first[3][w] << 16);
        *reinterpret_cast<uint2 *>(&vrank[threadIdx.x * kBinsPer]) = make_uint2(vr[0] | vr[1] << 16, vr[2] | vr[3] << 16);
      }
      if (threadIdx.x == kVbFinalThreads - 1) cnt[0][nbins] = (uint16_t)P;  // (a cell's points end where the next one's begin)
    }
    __syncthreads();
    PCGX_STAMP(vb_bucket, 8, b, 4);  // cells scanned
    if (threadIdx.x == 0) count[b] = occupied_total;
    // ---- the points to their places
#pragma unroll
    for (int r = 0; r < kRounds; r++) {
      const int i = w_begin + r * 64 + lane;
      if (i < w_end) {
        const int pos = (int)cnt[wave][__float_as_uint(e[r].w) & lmask] + (int)rank[r];
        sx[pos] = e[r].x;
        sy[pos] = e[r].y;
        sz[pos] = e[r].z;
        if (kIdx) sfrom[pos] = (uint16_t)i;
      }
    }
    __syncthreads();
    PCGX_STAMP(vb_bucket, 8, b, 5);  // points at their places
    // ---- cell by cell: the reference's sequential float32 sum over its points in input order, the centroid; thread t:
    // cells t, t + 256, ... (neighbouring lanes neighbouring cells: their points are neighbours in LDS, their stores too)
#pragma unroll
    for (int k = 0; k < kBinsPer; k++) {
      const int l = k * kVbFinalThreads + threadIdx.x;
      if (l >= nbins || (plan.dbg & 2)) continue;
      const int first = (int)cnt[0][l], c = (int)cnt[0][l + 1] - first;
      if (c == 0) continue;
      float o0 = sx[first], o1 = sy[first], o2 = sz[first];
      if (c > 1) {
        const uint32_t key = ((uint32_t)b << plan.low_bits) | (uint32_t)l;
        float origin[3] = {vp.vmin[0], vp.vmin[1], vp.vmin[2]};
        if (vp.chunked) chunk_origin(vp, vp.combined ? key / (uint32_t)vp.n_voxels : 0u, origin);
        // p := it.Vec3().Sub(vMin); v.sum = v.sum.Add(p)   (voxelgrid.go:149,157)
        float s0 = 0.0f + (o0 - origin[0]), s1 = 0.0f + (o1 - origin[1]), s2 = 0.0f + (o2 - origin[2]);
        for (int j = 1; j < c; j++) {
          s0 = s0 + (sx[first + j] - origin[0]);
          s1 = s1 + (sy[first + j] - origin[1]);
          s2 = s2 + (sz[first + j] - origin[2]);
        }
        const float inv = 1.0f / (float)c;  // jt.SetVec3(v.sum.Mul(1.0 / float32(n)).Add(vMin))  (voxelgrid.go:178-180)
        o0 = s0 * inv + origin[0];
        o1 = s1 * inv + origin[1];
        o2 = s2 * inv + origin[2];
      }
      // v.index: the cell's first point in input order (voxelgrid.go:152-155); the placing kernel copies its record
      const uint32_t src = kIdx ? idx[(int64_t)start + sfrom[first]] : 0u;
      cells[(int64_t)start + vrank[l]] = make_float4(o0, o1, o2, __uint_as_float(src));
    }
    PCGX_STAMP(vb_bucket, 8, b, 6);  // cell phase, stores issued
    __syncthreads();  // (the next bucket clears cnt[] and fills sx / sy / sz)
  }
}

// Sixteen buckets per workgroup, four per wave: their cells from where the bucket kernel left them to their place in the
// output = the occupied cells of all buckets before.  Every workgroup adds up the counts before its sixteen by itself
// -- ordinary loads, 52 KB at most at C3, out of the L2: 21 MB over the whole launch.  (A kernel of its own for the
// counts' prefix, one workgroup: 12 us of the call.  The bucket kernel's last workgroup -- a ticket, then loads past
// the L2, 3 us a round trip under that kernel's traffic: 12 us as well.  Totals per 32 and per 1024 buckets added up by
// the bucket kernel with fire-and-forget atomics: atomics on words of ONE 128-byte line are served one after the
// other, 10 ns each -- 13010 of them on the thirteen 1024-bucket totals made the bucket kernel 189 us long.)
constexpr int kVbPlaceBuckets = 16;
template <bool kIdx>
__global__ __launch_bounds__(256) void vb_place_kernel(const float4 *__restrict__ cells0, const float4 *__restrict__ cells1,
                                                       const uint32_t *__restrict__ bucket_start, const VoxelDevPlan *__restrict__ dp,
                                                       const uint8_t *__restrict__ data, int32_t stride, int32_t off,
                                                       uint8_t *__restrict__ out, const uint32_t *__restrict__ count,
                                                       int64_t *__restrict__ total, const int32_t *__restrict__ flags) {
  __shared__ uint32_t wsum[4];
  if (*flags) return;  // uniform: the radix path does the call
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nbuckets = dp->plan.nbuckets;
  const float4 *__restrict__ cells = dp->plan.d_bits[1] != 0 ? cells0 : cells1;
  const bool words = (stride & 3) == 0 && ((reinterpret_cast<uintptr_t>(data) | reinterpret_cast<uintptr_t>(out)) & 3) == 0;
  for (int b0 = (int)blockIdx.x * kVbPlaceBuckets; b0 < nbuckets; b0 += (int)gridDim.x * kVbPlaceBuckets) {  // uniform
    uint32_t s = 0;
    for (int i = (int)threadIdx.x * 4; i < b0; i += 1024) {  // (b0 is a multiple of 16: whole 16-byte pieces)
      const uint4 v = *reinterpret_cast<const uint4 *>(count + i);
      s += (v.x + v.y) + (v.z + v.w);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) wsum[wave] = s;
    __syncthreads();
    const uint32_t before = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
    const uint32_t c = lane < kVbPlaceBuckets && b0 + lane < nbuckets ? count[b0 + lane] : 0u;
    const uint32_t inc = wave_incl_scan_u32(c);
    if (wave == 0 && b0 + kVbPlaceBuckets >= nbuckets && lane == 63) *total = (int64_t)before + (int64_t)inc;  // the call's total
#pragma unroll
    for (int q = 0; q < kVbPlaceBuckets / 4; q++) {
      const int slot = wave * (kVbPlaceBuckets / 4) + q, b = b0 + slot;
      const uint32_t mine = (uint32_t)__shfl((int)c, slot), to0 = before + (uint32_t)__shfl((int)(inc - c), slot);
      if (b >= nbuckets || mine == 0u) continue;  // uniform in the wave
      const int64_t from = (int64_t)(~bucket_start[b]), to = (int64_t)to0;
      for (uint32_t j = lane; j < mine; j += 64) {
        const float4 cell = cells[from + j];
        uint8_t *dst = out + (to + j) * stride;
        if (!kIdx) {  // records are xyz and nothing else, 4-byte aligned
          float *d = reinterpret_cast<float *>(dst);
          d[0] = cell.x; d[1] = cell.y; d[2] = cell.z;
        } else {  // the first point's whole record, its xyz replaced (voxelgrid.go:173-184)
          const uint8_t *src = data + (int64_t)__float_as_uint(cell.w) * stride;
          if (words) {
            for (int k = 0; k < stride; k += 4) *reinterpret_cast<uint32_t *>(dst + k) = *reinterpret_cast<const uint32_t *>(src + k);
          } else {
            for (int k = 0; k < stride; k++) dst[k] = src[k];
          }
          __builtin_memcpy(dst + off, &cell.x, 4);
          __builtin_memcpy(dst + off + 4, &cell.y, 4);
          __builtin_memcpy(dst + off + 8, &cell.z, 4);
        }
      }
    }
    __syncthreads();  // (wsum)
  }
}

static std::atomic<long long> g_vb_taken{0}, g_vb_given_up{0}, g_vb_last_flags{0}, g_vb_last_low{0};

// the bucket kernel's grid: the workgroups the device holds at once (each walks the buckets blockIdx.x, + grid, ...)
static int vb_resident_grid(bool with_idx) {
  static int grid[2] = {0, 0};
  int &g = grid[with_idx ? 1 : 0];
  if (g == 0) {
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    hipError_t e = with_idx ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, vb_bucket_kernel<true>, kVbFinalThreads, 0)
                            : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, vb_bucket_kernel<false>, kVbFinalThreads, 0);
    if (e == hipSuccess) e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, dev);
    g = (e == hipSuccess && per_cu > 0) ? per_cu * prop.multiProcessorCount : 1024;
    (void)hipGetLastError();
  }
  return g;
}

static int vb_knob(const char *name, int def) {
  const char *e = getenv(name);
  return e ? atoi(e) : def;
}

// The bucket path of one filter call on one GPU (voxel_key.h).  The host does not wait for the cloud's min / max:
// the plan is made on the device (the min/max launch's last workgroup, sort.hip), every kernel reads it from there,
// buffers are sized for the largest plan there is, the grids for the device (the bucket and the placing kernel's
// workgroups walk the buckets), and what came of it all -- plan, flags, the number of output points -- is read back in one copy at the end.
// Before: 34 us of a C3 call in which the GPU waited for the host (six floats back, the plan, a dozen launches).
pcgx_status voxel_bucket_filter(const void *d_data, int64_t n, int32_t stride, int32_t xyz_off, const float leaf[3],
                                const int32_t chunk[3], void *d_out, int64_t *out_n, bool *attempted, bool *taken,
                                VoxelDevPlan *dp_host, hipStream_t st) {
  *attempted = *taken = false;
  const int enabled = vb_knob("PCGX_VOXEL_BUCKET", 1), min_n = vb_knob("PCGX_VOXEL_BUCKET_MIN_N", 1024);  // (0.067 against 0.112 ms at a thousand points, 0.091 against 0.142 at 400k)
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
  int32_t *d_flags = nullptr;  // [0] flags (1 crowded bucket, 8 no plan), [1] key out of range
  int64_t *d_total = nullptr;
  uint32_t *cell_count = nullptr;  // [buckets] occupied cells; then: where the bucket's cells begin in the output
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
  // one block, zeroed at once: flags, the buckets' cell counts and starts, the sample of the bucket populations.  What the host
  // reads back at the end is at its start: flags, err, total, the plan
  struct Readback {
    int32_t flags, err;
    int64_t total;
    VoxelDevPlan plan;
  };
  static_assert(sizeof(Readback) % 8 == 0, "words");
  constexpr size_t kHeadWords = (sizeof(Readback) + 127) / 128 * 32;
  const size_t zero_words = kHeadWords + (size_t)kVbMaxBuckets * 3;
  uint32_t *zero_block = nullptr;
  PCGX_TRY(ar.alloc_n(zero_words + 64, &zero_block));
  zero_block = reinterpret_cast<uint32_t *>((reinterpret_cast<uintptr_t>(zero_block) + 127) & ~(uintptr_t)127);
  Readback *d_rb = reinterpret_cast<Readback *>(zero_block);
  d_flags = &d_rb->flags;
  d_total = &d_rb->total;
  d_plan = &d_rb->plan;
  bucket_sample = zero_block + kHeadWords;
  cell_count = bucket_sample + kVbMaxBuckets;
  inv_start = cell_count + kVbMaxBuckets;
  // (no memset, no kernel for the plan: the min/max launch clears the words behind the head as it goes, and its last
  // workgroup writes the head -- flags, the plan -- behind the six floats)
  static_assert(kHeadWords % 4 == 0 && (kVbMaxBuckets * 3) % 4 == 0, "cleared 16 bytes at a time");
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
  // (grids: what the device holds at once -- the host does not know how many buckets the plan has; kn.grid bounds them)
  const int bucket_grid = vb_knob("PCGX_VOXEL_BUCKET_WGS", vb_resident_grid(with_idx)), place_grid = 2048;
  if (getenv("PCGX_VOXEL_BUCKET_TRACE")) fprintf(stderr, "pcgx voxel bucket path: bucket kernel grid %d\n", bucket_grid);
  if (with_idx) {
    hipLaunchKernelGGL(vb_bucket_kernel<true>, dim3(bucket_grid), dim3(kVbFinalThreads), 0, st, (const float4 *)recb[0], (const float4 *)recb[1],
                       n, (const uint32_t *)idxb[0], (const uint32_t *)idxb[1], (const uint32_t *)inv_start, (const VoxelDevPlan *)d_plan,
                       recb[0], recb[1], cell_count, d_flags);
    hipLaunchKernelGGL(vb_place_kernel<true>, dim3(place_grid), dim3(256), 0, st, (const float4 *)recb[0], (const float4 *)recb[1],
                       (const uint32_t *)inv_start, (const VoxelDevPlan *)d_plan, data, stride, xyz_off, (uint8_t *)d_out,
                       (const uint32_t *)cell_count, d_total, (const int32_t *)d_flags);
  } else {
    hipLaunchKernelGGL(vb_bucket_kernel<false>, dim3(bucket_grid), dim3(kVbFinalThreads), 0, st, (const float4 *)recb[0], (const float4 *)recb[1],
                       n, (const uint32_t *)idxb[0], (const uint32_t *)idxb[1], (const uint32_t *)inv_start, (const VoxelDevPlan *)d_plan,
                       recb[0], recb[1], cell_count, d_flags);
    hipLaunchKernelGGL(vb_place_kernel<false>, dim3(place_grid), dim3(256), 0, st, (const float4 *)recb[0], (const float4 *)recb[1],
                       (const uint32_t *)inv_start, (const VoxelDevPlan *)d_plan, data, stride, xyz_off, (uint8_t *)d_out,
                       (const uint32_t *)cell_count, d_total, (const int32_t *)d_flags);
  }
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
  if (h.flags) {  // a crowded bucket: the radix path does the call
    g_vb_given_up++;
    g_vb_last_flags = h.flags;
    if (getenv("PCGX_VOXEL_BUCKET_TRACE"))
      fprintf(stderr, "pcgx voxel bucket path: flags %d (1: a bucket over %d points); low bits %d, %d buckets, digits %d + %d\n",
              h.flags, kVbCap, h.plan.plan.low_bits, h.plan.plan.nbuckets, h.plan.plan.d_bits[0], h.plan.plan.d_bits[1]);
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
