// voxel.hip -- VoxelGrid downsample filter on gfx950 + pcgx_minmax / pcgx_voxel_filter.
//
// Reference: pc/filter/voxelgrid/voxelgrid.go:35-187 (Filter / filterChunk),
// pc/minmax.go:9-26.  The reference scatters into a dense []voxel array and
// then scans the whole array; here the same result is produced sparsely:
//
//   1. min/max reduction (sort.hip)                         -> vMin, vMax
//   2. key kernel: a = x + xs*(y + ys*z) per point, with the reference's
//      quirks kept (non-chunked size = vMax, stride xs/ys on an
//      (xs+1)(ys+1)(zs+1) array: voxelgrid.go:46,137-138,151); chunked mode
//      also yields the chunk id (voxelgrid.go:76-79)
//   3. STABLE radix sort of (key, point index) (sort.hip); chunked mode sorts
//      by `a` and then by chunk id, giving (cid, a) order = the reference's
//      output order (chunks ascending, cells ascending inside a chunk)
//   4. segment kernel: one lane per occupied voxel walks its points in input
//      order (stability!) doing the sequential float32 sum of voxelgrid.go:157,
//      copies the first point's whole record and overwrites xyz with
//      sum*(1/num)+origin when num > 1 (voxelgrid.go:173-184).
//
// Out-of-range cell indices make the Go code panic; here they raise
// PCGX_E_OUT_OF_RANGE.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <utility>
#include <vector>

#include "voxel_key.h"

namespace pcgx {

// the key of point i (and its by-products); returns it
__device__ __forceinline__ uint32_t voxel_key_of(const uint8_t *__restrict__ data, int64_t i, int32_t stride, int32_t off,
                                                 const VoxelParams &vp, uint32_t *__restrict__ key_a,
                                                 uint32_t *__restrict__ a_orig, uint32_t *__restrict__ key_cid,
                                                 uint32_t *__restrict__ idx, int32_t *__restrict__ err);

__global__ __launch_bounds__(256) void voxel_key_kernel(const uint8_t *__restrict__ data, int64_t n,
                                                        int32_t stride, int32_t off, VoxelParams vp,
                                                        uint32_t *__restrict__ key_a,
                                                        uint32_t *__restrict__ a_orig,
                                                        uint32_t *__restrict__ key_cid,
                                                        uint32_t *__restrict__ idx,
                                                        int32_t *__restrict__ err) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  (void)voxel_key_of(data, i, stride, off, vp, key_a, a_orig, key_cid, idx, err);
}

// The same, a radix-sort tile per workgroup (256 x kItems consecutive points), counting the first digit of the keys
// on its way: what rs_hist_kernel would do in a launch of its own with a 40 MB read of what was just written
// (sort.hip, radix_first_hist)
template <int kItems>
__global__ __launch_bounds__(256) void voxel_key_hist_kernel(const uint8_t *__restrict__ data, int64_t n, int32_t stride,
                                                             int32_t off, VoxelParams vp, uint32_t *__restrict__ key_a,
                                                             uint32_t *__restrict__ a_orig, uint32_t *__restrict__ key_cid,
                                                             int32_t *__restrict__ err, uint32_t *__restrict__ block_hist,
                                                             int nblocks) {
  __shared__ uint32_t hist[256];
  hist[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * (256 * kItems);
#pragma unroll
  for (int r = 0; r < kItems; r++) {
    const int64_t i = base + r * 256 + threadIdx.x;
    if (i < n) {
      const uint32_t k = voxel_key_of(data, i, stride, off, vp, key_a, a_orig, key_cid, nullptr, err);
      atomicAdd(&hist[k & 255u], 1u);
    }
  }
  __syncthreads();
  block_hist[(int64_t)threadIdx.x * nblocks + blockIdx.x] = hist[threadIdx.x];
}

__device__ __forceinline__ uint32_t voxel_key_of(const uint8_t *__restrict__ data, int64_t i, int32_t stride, int32_t off,
                                                 const VoxelParams &vp, uint32_t *__restrict__ key_a,
                                                 uint32_t *__restrict__ a_orig, uint32_t *__restrict__ key_cid,
                                                 uint32_t *__restrict__ idx, int32_t *__restrict__ err) {
  const uint8_t *rec = data + i * stride + off;
  const float pt[3] = {ld_f32(rec), ld_f32(rec + 4), ld_f32(rec + 8)};
  uint32_t cid, ka;
  bool bad;
  const uint32_t key = voxel_key_xyz(pt, vp, cid, ka, bad);
  if (bad) atomicOr(err, 1);  // nIndices[cid] / f.voxels[a] would panic
  key_a[i] = key;
  if (key_cid) key_cid[i] = cid;
  if (a_orig) a_orig[i] = ka;
  if (idx) idx[i] = (uint32_t)i;  // (nullptr: the sort takes positions for values)
  return key;
}

// chunk id of sorted position j: its own array (two sorts), or the quotient of the combined key
__device__ __forceinline__ uint32_t sorted_cid(const VoxelParams &vp, const uint32_t *__restrict__ sa,
                                               const uint32_t *__restrict__ sc, int64_t j) {
  return sc ? sc[j] : (vp.combined ? sa[j] / (uint32_t)vp.n_voxels : 0u);
}

__global__ __launch_bounds__(256) void gather_u32_kernel(const uint32_t *__restrict__ src,
                                                         const uint32_t *__restrict__ index, int64_t n,
                                                         uint32_t *__restrict__ dst) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j < n) dst[j] = src[index[j]];
}

// ---- head flags + exclusive scan + per-voxel reduction ------------------------
constexpr int kSegTile = 1024;  // elements per block (256 threads x 4; 2048 and 4096 measured slower)

__device__ __forceinline__ bool is_head(const uint32_t *__restrict__ sa, const uint32_t *__restrict__ sc,
                                        int64_t j) {
  if (j == 0) return true;
  if (sa[j] != sa[j - 1]) return true;
  return sc && sc[j] != sc[j - 1];
}

__global__ __launch_bounds__(256) void seg_count_kernel(const uint32_t *__restrict__ sa,
                                                        const uint32_t *__restrict__ sc, int64_t n,
                                                        uint32_t *__restrict__ tile_count) {
  __shared__ uint32_t ws[4];
  const int64_t base = (int64_t)blockIdx.x * kSegTile;
  uint32_t c = 0;
  for (int r = 0; r < kSegTile / 256; r++) {
    const int64_t j = base + r * 256 + threadIdx.x;
    if (j < n && is_head(sa, sc, j)) c++;
  }
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) tile_count[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// In-place exclusive scan of tile_count[0..ntiles) by one block; total -> *total.
__global__ __launch_bounds__(1024) void seg_scan_kernel(uint32_t *__restrict__ tile_count, int ntiles,
                                                        int64_t *__restrict__ total) {
  __shared__ uint32_t ws[16];
  __shared__ uint32_t carry_s;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int start = 0; start < ntiles; start += 1024) {
    const int i = start + threadIdx.x;
    const uint32_t v = i < ntiles ? tile_count[i] : 0u;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      uint32_t t = __shfl_up(inc, o);
      if (lane >= o) inc += t;
    }
    if (lane == 63) ws[wave] = inc;
    __syncthreads();
    uint32_t wbase = 0;
    for (int w = 0; w < wave; w++) wbase += ws[w];
    const uint32_t carry = carry_s;
    if (i < ntiles) tile_count[i] = carry + wbase + inc - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = carry + wbase + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = (int64_t)carry_s;
}

// One lane per sorted position; head lanes own a voxel.
//
// Phase A: every lane gathers the points of its 8 sorted positions (8 independent random
// 12-byte reads in flight per lane: the gather is latency bound, so memory-level parallelism
// is what counts) and parks p = pt - origin in LDS together with a head flag.
// Phase B: head lanes run the reference's sequential float32 sum (voxelgrid.go:157) over
// their segment out of LDS; only a segment that runs past the tile's end continues with
// global gathers (at most one per tile).
__global__ __launch_bounds__(256) void seg_reduce_kernel(
    const uint8_t *__restrict__ data, int32_t stride, int32_t off, VoxelParams vp,
    const uint32_t *__restrict__ sa, const uint32_t *__restrict__ sc, const uint32_t *__restrict__ sidx,
    int64_t n, const uint32_t *__restrict__ tile_offset, uint8_t *__restrict__ out) {
  constexpr int kRounds = kSegTile / 256;
  __shared__ float sp[3][kSegTile];
  __shared__ uint8_t shead[kSegTile];
  __shared__ uint32_t ws[kRounds][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t base = (int64_t)blockIdx.x * kSegTile;
  const int64_t rem = n - base;
  const int count = rem < kSegTile ? (int)rem : kSegTile;

  uint32_t idx[kRounds];
  float v[kRounds][3];
  bool head[kRounds];
#pragma unroll
  for (int r = 0; r < kRounds; r++) {
    const int l = r * 256 + threadIdx.x;
    idx[r] = l < count ? sidx[base + l] : 0u;
  }
#pragma unroll
  for (int r = 0; r < kRounds; r++) {
    const uint8_t *rec = data + (int64_t)idx[r] * stride + off;
    v[r][0] = ld_f32(rec);
    v[r][1] = ld_f32(rec + 4);
    v[r][2] = ld_f32(rec + 8);
  }
#pragma unroll
  for (int r = 0; r < kRounds; r++) {
    const int l = r * 256 + threadIdx.x;
    head[r] = l < count && is_head(sa, sc, base + l);
    float origin[3] = {vp.vmin[0], vp.vmin[1], vp.vmin[2]};
    if (vp.chunked && l < count) chunk_origin(vp, sorted_cid(vp, sa, sc, base + l), origin);
    // p := it.Vec3().Sub(vMin)   (voxelgrid.go:149)
    sp[0][l] = v[r][0] - origin[0];
    sp[1][l] = v[r][1] - origin[1];
    sp[2][l] = v[r][2] - origin[2];
    shead[l] = head[r] ? 1 : 0;
    const uint64_t bal = __ballot(head[r]);
    if (lane == 0) ws[r][wave] = (uint32_t)__popcll(bal);
  }
  __syncthreads();

  const uint64_t lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  uint32_t running = tile_offset[blockIdx.x];
#pragma unroll
  for (int r = 0; r < kRounds; r++) {
    uint32_t wbase = 0, round_total = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
      const uint32_t c = ws[r][w];
      if (w < wave) wbase += c;
      round_total += c;
    }
    const uint64_t bal = __ballot(head[r]);
    if (head[r]) {
      int l = r * 256 + threadIdx.x;
      const int64_t j = base + l;
      const uint32_t slot = running + wbase + (uint32_t)__popcll(bal & lt_mask);
      const uint32_t cid = sorted_cid(vp, sa, sc, j);
      float origin[3] = {vp.vmin[0], vp.vmin[1], vp.vmin[2]};
      if (vp.chunked) chunk_origin(vp, cid, origin);
      float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
      uint32_t num = 0;
      do {  // v.sum = v.sum.Add(p)   (voxelgrid.go:157), points in input order (stable sort)
        s0 = s0 + sp[0][l];
        s1 = s1 + sp[1][l];
        s2 = s2 + sp[2][l];
        num++;
        l++;
      } while (l < count && !shead[l]);
      if (l == kSegTile) {  // the segment may continue in the next tile
        for (int64_t k = base + kSegTile; k < n && !is_head(sa, sc, k); k++) {
          const uint8_t *rec = data + (int64_t)sidx[k] * stride + off;
          s0 = s0 + (ld_f32(rec) - origin[0]);
          s1 = s1 + (ld_f32(rec + 4) - origin[1]);
          s2 = s2 + (ld_f32(rec + 8) - origin[2]);
          num++;
        }
      }
      // v.index: first point in input order (voxelgrid.go:152-155); its whole record is copied
      const uint8_t *src = data + (int64_t)idx[r] * stride;
      uint8_t *dst = out + (int64_t)slot * stride;
      if (((stride | off) & 3) == 0 && ((reinterpret_cast<uintptr_t>(data) | reinterpret_cast<uintptr_t>(out)) & 3) == 0) {
        if (stride == 12) {
          float *d = (float *)dst;
          d[0] = v[r][0]; d[1] = v[r][1]; d[2] = v[r][2];
        } else {
          for (int b = 0; b < stride; b += 4) *(uint32_t *)(dst + b) = *(const uint32_t *)(src + b);
        }
      } else {
        for (int b = 0; b < stride; b++) dst[b] = src[b];
      }
      if (num > 1) {  // jt.SetVec3(v.sum.Mul(1.0 / float32(n)).Add(vMin))  (voxelgrid.go:178-180)
        const float inv = 1.0f / (float)num;
        const float c0 = s0 * inv + origin[0], c1 = s1 * inv + origin[1], c2 = s2 * inv + origin[2];
        __builtin_memcpy(dst + off, &c0, 4);
        __builtin_memcpy(dst + off + 4, &c1, 4);
        __builtin_memcpy(dst + off + 8, &c2, 4);
      }
    }
    running += round_total;
  }
}

// ---- sharded filter: a rank keeps the points of ITS share of the output order -------------------
// primary[i] in [lo, hi): the (key, index) pairs of those points, input order kept (the sort's
// stability is what orders a voxel's points).  The two-sort mode's cell / chunk arrays stay as they
// are: they are looked up by point index.
constexpr int kMineTile = 2048;  // 256 threads x 8 rounds

__global__ __launch_bounds__(256) void vx_mine_count_kernel(const uint32_t *__restrict__ primary, int64_t n, uint64_t lo,
                                                            uint64_t hi, uint32_t *__restrict__ tile_count) {
  __shared__ uint32_t ws[4];
  const int64_t base = (int64_t)blockIdx.x * kMineTile;
  uint32_t c = 0;
  for (int r = 0; r < kMineTile / 256; r++) {
    const int64_t j = base + r * 256 + threadIdx.x;
    if (j < n) {
      const uint32_t k = primary[j];
      c += (k >= lo && k < hi) ? 1u : 0u;
    }
  }
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) tile_count[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

__global__ __launch_bounds__(256) void vx_mine_compact_kernel(
    const uint32_t *__restrict__ primary, int64_t n, uint64_t lo, uint64_t hi, const uint32_t *__restrict__ tile_start,
    const uint32_t *__restrict__ key_in, const uint32_t *__restrict__ idx_in, uint32_t *__restrict__ key_out,
    uint32_t *__restrict__ idx_out) {
  constexpr int kRounds = kMineTile / 256;
  __shared__ uint32_t cnt[kRounds][4];
  const int64_t base = (int64_t)blockIdx.x * kMineTile;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  uint32_t below[kRounds];
  bool mine[kRounds];
#pragma unroll
  for (int r = 0; r < kRounds; r++) {  // element base + r * 256 + t: rounds, then waves, then lanes = input order
    const int64_t j = base + r * 256 + threadIdx.x;
    mine[r] = false;
    if (j < n) {
      const uint32_t k = primary[j];
      mine[r] = k >= lo && k < hi;
    }
    const uint64_t m = __ballot(mine[r]);
    below[r] = (uint32_t)__popcll(m & lt_mask);
    if (lane == 0) cnt[r][wave] = (uint32_t)__popcll(m);
  }
  __syncthreads();
  uint32_t run = tile_start[blockIdx.x];
#pragma unroll
  for (int r = 0; r < kRounds; r++) {
    uint32_t before = run;
    for (int w = 0; w < 4; w++) {
      if (w < wave) before += cnt[r][w];
      run += cnt[r][w];
    }
    if (mine[r]) {
      const int64_t j = base + r * 256 + threadIdx.x;
      const uint32_t dst = before + below[r];
      key_out[dst] = key_in[j];
      idx_out[dst] = idx_in ? idx_in[j] : (uint32_t)j;
    }
  }
}

static int bits_for(int64_t count) { return voxel_bits_for(count); }

static pcgx_status make_params(const float mm6[6], const float leaf[3], const int32_t chunk[3], VoxelParams &vp) {
  return voxel_grid_params_or_fail(mm6, leaf, chunk, vp);
}

}  // namespace pcgx

using namespace pcgx;

// comm == nullptr: the whole filter.  Otherwise this rank's share (second half): every
// rank holds the same cloud; the min/max pass (pc/minmax.go:9-26 via voxelgrid.go:41-44) runs over the
// rank's slice of it and the six floats are exchanged; each rank then keeps the points whose place in
// the reference's output order (chunk id, cell: voxelgrid.go:49-116) falls into its contiguous share
// of that key range, and filters those.  The ranks' outputs one after the other are the reference's
// output, byte for byte.
static pcgx_status voxel_filter_core(pcgx_comm *comm, const void *d_data, int64_t n, int32_t stride, int32_t xyz_off,
                                     const float leaf[3], const int32_t chunk[3], void *d_out, int64_t *out_n,
                                     void *stream) {
  if (!out_n) return fail(PCGX_E_INVALID, "pcgx_voxel_filter_dev: out_n is NULL");
  *out_n = 0;
  if (n < 0 || !leaf || !chunk || (n > 0 && (!d_data || !d_out)))
    return fail(PCGX_E_INVALID, "pcgx_voxel_filter_dev: bad argument");
  if (n == 0) return fail(PCGX_E_NO_POINT, "no point");  // pc/minmax.go:10-12 via voxelgrid.go:41-44
  if (stride < 12 || xyz_off < 0 || xyz_off + 12 > stride)
    return fail(PCGX_E_BAD_FIELD, "pcgx_voxel_filter_dev: stride %d / xyz offset %d do not hold an xyz triple", stride, xyz_off);
  if (n > 0x7fffffffll) return fail(PCGX_E_TOO_LARGE, "pcgx_voxel_filter_dev: more than 2^31-1 points");
  PCGX_TRY(ensure_init());
  hipStream_t st = pick_stream(stream);
  Arena &ar = ctx().arena;
  PCGX_TRY(ar.begin(st));
  ProfScope prof(PCGX_PROF_VOXEL_ALL, st);

  int32_t rank = 0, world = 1;
  if (comm) PCGX_TRY(pcgx_comm_rank(comm, &rank, &world));
  float *d_mm6 = nullptr;
  PCGX_TRY(ar.alloc_n(6, &d_mm6));
  float mm6[6];
  if (world == 1) {
    // the points travel with their keys (voxel_bucket.hip) where the cloud allows it: no 12-byte gather per point, and
    // min/max, the grid and every kernel behind them in one go, without the host in between
    bool attempted = false, taken = false;
    VoxelDevPlan dp;
    PCGX_TRY(voxel_bucket_filter(d_data, n, stride, xyz_off, leaf, chunk, d_out, out_n, &attempted, &taken, &dp, st));
    if (taken) return PCGX_OK;
    if (attempted) {
      memcpy(mm6, dp.mm6, sizeof mm6);
      PCGX_TRY(ar.begin(st));  // (the attempt's temporaries are free again)
      PCGX_TRY(ar.alloc_n(6, &d_mm6));
    } else {
      PCGX_TRY(minmax_to_host(d_data, n, stride, xyz_off, d_mm6, mm6, st));
    }
  } else {
    // the slice's six floats travel as their bit patterns in slot `rank` of a vector of zeros: the
    // sum every rank receives holds all of them exactly (an all-gather out of the one collective the
    // communicator has); folded in rank order with the reference's comparisons (`<`, `>`: of equal
    // values -- +0 and -0 -- the earlier one stays, as in the sequential loop)
    const int64_t s0 = n * rank / world, s1 = n * (rank + 1) / world;
    const int per = 7;
    std::vector<double> slots((size_t)per * world, 0.0);
    if (s1 > s0) {
      PCGX_TRY(launch_minmax((const uint8_t *)d_data + s0 * stride, s1 - s0, stride, xyz_off, d_mm6, st, s0 == 0));
      PCGX_HIP_TRY(hipMemcpyAsync(mm6, d_mm6, sizeof mm6, hipMemcpyDeviceToHost, st));
      PCGX_HIP_TRY(hipStreamSynchronize(st));
      slots[(size_t)per * rank] = 1.0;
      for (int k = 0; k < 6; k++) {
        uint32_t bits;
        memcpy(&bits, &mm6[k], 4);
        slots[(size_t)per * rank + 1 + k] = (double)bits;
      }
    }
    double *d_slots = nullptr;
    PCGX_TRY(ar.alloc_n(slots.size(), &d_slots));
    PCGX_HIP_TRY(hipMemcpyAsync(d_slots, slots.data(), slots.size() * sizeof(double), hipMemcpyHostToDevice, st));
    PCGX_TRY(pcgx_comm_allreduce_f64(comm, d_slots, (int32_t)slots.size(), st));
    PCGX_HIP_TRY(hipMemcpyAsync(slots.data(), d_slots, slots.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    PCGX_HIP_TRY(hipStreamSynchronize(st));
    bool have = false;
    for (int r = 0; r < world; r++) {
      if (slots[(size_t)per * r] != 1.0) continue;
      float v[6];
      for (int k = 0; k < 6; k++) {
        const uint32_t bits = (uint32_t)slots[(size_t)per * r + 1 + k];
        memcpy(&v[k], &bits, 4);
      }
      if (!have) {
        memcpy(mm6, v, sizeof mm6);
        have = true;
        continue;
      }
      for (int k = 0; k < 3; k++) {
        if (v[k] < mm6[k]) mm6[k] = v[k];
        if (v[3 + k] > mm6[3 + k]) mm6[3 + k] = v[3 + k];
      }
    }
    if (!have) return fail(PCGX_E_RCCL, "sharded voxel filter: no rank reported a slice");
  }
  VoxelParams vp;
  PCGX_TRY(make_params(mm6, leaf, chunk, vp));

  uint32_t *keys[2], *vals[2], *a_orig = nullptr, *cid_orig = nullptr;
  void *ws = nullptr;
  int32_t *d_err = nullptr;
  int64_t *d_total = nullptr;
  uint32_t *tile_count = nullptr;
  int ntiles = (int)((n + kSegTile - 1) / kSegTile);
  const char *force_two = getenv("PCGX_VOXEL_TWO_SORTS");  // tests: keep the two-sort path covered
  bool two_level = false;
  const int key_bits = voxel_key_layout(vp, force_two && force_two[0] == '1', &two_level);
  PCGX_TRY(ar.alloc_n((size_t)n, &keys[0]));
  PCGX_TRY(ar.alloc_n((size_t)n, &keys[1]));
  PCGX_TRY(ar.alloc_n((size_t)n, &vals[0]));
  PCGX_TRY(ar.alloc_n((size_t)n, &vals[1]));
  if (two_level) {
    PCGX_TRY(ar.alloc_n((size_t)n, &a_orig));
    PCGX_TRY(ar.alloc_n((size_t)n, &cid_orig));
  }
  PCGX_TRY(ar.alloc(radix_sort_workspace_bytes(n), &ws));
  PCGX_TRY(ar.alloc_n(1, &d_err));
  PCGX_TRY(ar.alloc_n(1, &d_total));
  PCGX_TRY(ar.alloc_n((size_t)ntiles, &tile_count));
  PCGX_HIP_TRY(hipMemsetAsync(d_err, 0, sizeof(int32_t), st));

  unsigned nb = (unsigned)((n + 255) / 256);
  // one GPU: the keys are sorted as they come out of the key kernel, which then counts the first digit as well
  const bool hist_fused = world == 1 && n > 1 && key_bits > 0;
  if (hist_fused) {
    const RadixFirstHist fh = radix_first_hist(n, ws);
    if (fh.items == 8)
      hipLaunchKernelGGL(voxel_key_hist_kernel<8>, dim3(fh.nblocks), dim3(256), 0, st, (const uint8_t *)d_data, n, stride,
                         xyz_off, vp, keys[0], a_orig, cid_orig, d_err, fh.hist, fh.nblocks);
    else
      hipLaunchKernelGGL(voxel_key_hist_kernel<16>, dim3(fh.nblocks), dim3(256), 0, st, (const uint8_t *)d_data, n, stride,
                         xyz_off, vp, keys[0], a_orig, cid_orig, d_err, fh.hist, fh.nblocks);
  } else {
    hipLaunchKernelGGL(voxel_key_kernel, dim3(nb), dim3(256), 0, st, (const uint8_t *)d_data, n, stride, xyz_off,
                       vp, keys[0], a_orig, cid_orig, (uint32_t *)nullptr, d_err);
  }
  bool iota = true;  // the values of the sort are the points' indices 0 .. n-1
  if (world > 1) {
    // this rank's share of the output order: by chunk id (two sorts), else by the one key
    const uint32_t *primary = two_level ? cid_orig : keys[0];
    const uint64_t span = two_level ? (uint64_t)vp.n_chunks
                                    : (vp.combined ? (uint64_t)vp.n_chunks * (uint64_t)vp.n_voxels : (uint64_t)vp.n_voxels);
    const uint64_t lo = span * (uint64_t)rank / (uint64_t)world, hi = span * (uint64_t)(rank + 1) / (uint64_t)world;
    const int mtiles = (int)((n + kMineTile - 1) / kMineTile);
    uint32_t *mine_count = nullptr;
    PCGX_TRY(ar.alloc_n((size_t)mtiles, &mine_count));
    hipLaunchKernelGGL(vx_mine_count_kernel, dim3(mtiles), dim3(256), 0, st, primary, n, lo, hi, mine_count);
    hipLaunchKernelGGL(seg_scan_kernel, dim3(1), dim3(1024), 0, st, mine_count, mtiles, d_total);
    hipLaunchKernelGGL(vx_mine_compact_kernel, dim3(mtiles), dim3(256), 0, st, primary, n, lo, hi,
                       (const uint32_t *)mine_count,
                       (const uint32_t *)keys[0], (const uint32_t *)nullptr, keys[1], vals[1]);
    iota = false;
    int32_t h_err0 = 0;
    int64_t h_mine = 0;
    PCGX_HIP_TRY(hipMemcpyAsync(&h_err0, d_err, sizeof h_err0, hipMemcpyDeviceToHost, st));
    PCGX_HIP_TRY(hipMemcpyAsync(&h_mine, d_total, sizeof h_mine, hipMemcpyDeviceToHost, st));
    PCGX_HIP_TRY(hipStreamSynchronize(st));
    if (h_err0)  // every rank sees every point: they all fail alike
      return fail(PCGX_E_OUT_OF_RANGE, "voxel filter: a point falls outside the dense grid (the reference panics: index out of range)");
    if (h_mine == 0) return PCGX_OK;  // nothing of the cloud falls into this rank's share
    std::swap(keys[0], keys[1]);
    std::swap(vals[0], vals[1]);
    n = h_mine;
    ntiles = (int)((n + kSegTile - 1) / kSegTile);
    nb = (unsigned)((n + 255) / 256);
  }
  int res = 0;
  PCGX_TRY(radix_sort_pairs(keys, vals, n, key_bits, ws, &res, st, iota, hist_fused));
  const uint32_t *sa = keys[res], *sc = nullptr, *sidx = vals[res];
  if (two_level) {
    // second stable sort, by chunk id: (cid, a) order with input order kept inside a cell
    uint32_t *k2[2] = {keys[res ^ 1], keys[res]};
    uint32_t *v2[2] = {vals[res], vals[res ^ 1]};
    hipLaunchKernelGGL(gather_u32_kernel, dim3(nb), dim3(256), 0, st, cid_orig, v2[0], n, k2[0]);
    int res2 = 0;
    PCGX_TRY(radix_sort_pairs(k2, v2, n, bits_for(vp.n_chunks), ws, &res2, st));
    sc = k2[res2];
    sidx = v2[res2];
    uint32_t *sorted_a = k2[res2 ^ 1];
    hipLaunchKernelGGL(gather_u32_kernel, dim3(nb), dim3(256), 0, st, a_orig, sidx, n, sorted_a);
    sa = sorted_a;
  }
  hipLaunchKernelGGL(seg_count_kernel, dim3(ntiles), dim3(256), 0, st, sa, sc, n, tile_count);
  hipLaunchKernelGGL(seg_scan_kernel, dim3(1), dim3(1024), 0, st, tile_count, ntiles, d_total);
  hipLaunchKernelGGL(seg_reduce_kernel, dim3(ntiles), dim3(256), 0, st, (const uint8_t *)d_data, stride,
                     xyz_off, vp, sa, sc, sidx, n, tile_count, (uint8_t *)d_out);
  PCGX_HIP_TRY(hipGetLastError());
  int32_t h_err = 0;
  int64_t h_total = 0;
  PCGX_HIP_TRY(hipMemcpyAsync(&h_err, d_err, sizeof h_err, hipMemcpyDeviceToHost, st));
  PCGX_HIP_TRY(hipMemcpyAsync(&h_total, d_total, sizeof h_total, hipMemcpyDeviceToHost, st));
  PCGX_HIP_TRY(hipStreamSynchronize(st));
  if (h_err)
    return fail(PCGX_E_OUT_OF_RANGE, "voxel filter: a point falls outside the dense grid (the reference panics: index out of range)");
  *out_n = h_total;
  return PCGX_OK;
}

extern "C" pcgx_status pcgx_voxel_filter_dev(const void *d_data, int64_t n, int32_t stride,
                                             int32_t xyz_off, const float leaf[3],
                                             const int32_t chunk[3], void *d_out, int64_t *out_n,
                                             void *stream) {
  PCGX_API_LOCK();
  return voxel_filter_core(nullptr, d_data, n, stride, xyz_off, leaf, chunk, d_out, out_n, stream);
}

extern "C" pcgx_status pcgx_voxel_filter_sharded_dev(pcgx_comm *comm, const void *d_data, int64_t n, int32_t stride,
                                                     int32_t xyz_off, const float leaf[3], const int32_t chunk[3],
                                                     void *d_out, int64_t *out_n, void *stream) {
  PCGX_API_LOCK();
  if (!comm) return fail(PCGX_E_INVALID, "pcgx_voxel_filter_sharded_dev: comm is NULL");
  return voxel_filter_core(comm, d_data, n, stride, xyz_off, leaf, chunk, d_out, out_n, stream);
}

extern "C" pcgx_status pcgx_minmax(const void *data, int64_t n, int32_t stride, int32_t xyz_off,
                                   float vmin[3], float vmax[3]) {
  PCGX_API_CALL();
  if (n < 0 || !vmin || !vmax || (n > 0 && !data)) return fail(PCGX_E_INVALID, "pcgx_minmax: bad argument");
  if (n == 0) return fail(PCGX_E_NO_POINT, "no point");
  if (stride < 12 || xyz_off < 0 || xyz_off + 12 > stride)
    return fail(PCGX_E_BAD_FIELD, "pcgx_minmax: stride %d / xyz offset %d do not hold an xyz triple", stride, xyz_off);
  PCGX_TRY(ensure_init());
  hipStream_t st = ctx().stream;
  const void *src = data;
  const int32_t s = stride, o = xyz_off;
  uint8_t *d = nullptr;
  PCGX_TRY(ctx().host_arena.begin(st));
  PCGX_TRY(ctx().host_arena.alloc_n((size_t)n * s, &d));
  pcgx_status rc = PCGX_OK;
  float mm6[6];
  hipError_t e = hipSuccess;
  rc = staged_upload(d, src, (size_t)n * s, st);
  float *d_mm6 = nullptr;
  if (rc == PCGX_OK) rc = ctx().arena.begin(st);
  if (rc == PCGX_OK) rc = ctx().arena.alloc_n(6, &d_mm6);
  if (rc == PCGX_OK) rc = launch_minmax(d, n, s, o, d_mm6, st);
  if (rc == PCGX_OK) {
    e = hipMemcpyAsync(mm6, d_mm6, sizeof mm6, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) rc = fail(PCGX_E_HIP, "pcgx_minmax: %s", hipGetErrorString(e));
  }
  if (rc != PCGX_OK) return rc;
  memcpy(vmin, mm6, 12);
  memcpy(vmax, mm6 + 3, 12);
  return PCGX_OK;
}

// host buffers in and out around the device filter (comm: this rank's share, or nullptr)
static pcgx_status voxel_filter_host(pcgx_comm *comm, const void *data, int64_t n, int32_t stride, int32_t xyz_off,
                                     const float leaf[3], const int32_t chunk[3], void *out_data, int64_t *out_n) {
  if (!out_n) return fail(PCGX_E_INVALID, "pcgx_voxel_filter: out_n is NULL");
  *out_n = 0;
  if (n < 0 || (n > 0 && (!data || !out_data))) return fail(PCGX_E_INVALID, "pcgx_voxel_filter: bad argument");
  if (n == 0) return fail(PCGX_E_NO_POINT, "no point");
  PCGX_TRY(ensure_init());
  hipStream_t st = ctx().stream;
  uint8_t *d_in = nullptr, *d_out = nullptr;
  const size_t bytes = (size_t)n * (size_t)stride;
  PCGX_TRY(ctx().host_arena.begin(st));
  PCGX_TRY(ctx().host_arena.alloc_n(bytes, &d_in));
  PCGX_TRY(ctx().host_arena.alloc_n(bytes, &d_out));
  pcgx_status rc = PCGX_OK;
  rc = staged_upload(d_in, data, bytes, st);  // the caller's (pageable) records through the pinned ring
  int64_t m = 0;
  if (rc == PCGX_OK) rc = voxel_filter_core(comm, d_in, n, stride, xyz_off, leaf, chunk, d_out, &m, st);
  if (rc == PCGX_OK && m > 0) rc = staged_download(out_data, d_out, (size_t)m * (size_t)stride, st);  // same stream as the filter
  if (rc == PCGX_OK) *out_n = m;
  return rc;
}

extern "C" pcgx_status pcgx_voxel_filter(const void *data, int64_t n, int32_t stride, int32_t xyz_off,
                                         const float leaf[3], const int32_t chunk[3], void *out_data,
                                         int64_t *out_n) {
  PCGX_API_CALL();
  return voxel_filter_host(nullptr, data, n, stride, xyz_off, leaf, chunk, out_data, out_n);
}

extern "C" pcgx_status pcgx_voxel_filter_sharded(pcgx_comm *comm, const void *data, int64_t n, int32_t stride,
                                                 int32_t xyz_off, const float leaf[3], const int32_t chunk[3],
                                                 void *out_data, int64_t *out_n) {
  PCGX_API_LOCK();  // the exchange inside is collective: one such call at a time per process
  if (!comm) return fail(PCGX_E_INVALID, "pcgx_voxel_filter_sharded: comm is NULL");
  return voxel_filter_host(comm, data, n, stride, xyz_off, leaf, chunk, out_data, out_n);
}
