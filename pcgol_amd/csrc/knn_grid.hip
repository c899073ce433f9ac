// knn_grid.hip -- build of the uniform grid and the certified-nearest kernel (see knn_grid.h).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "knn_grid.h"

namespace pcgx {

constexpr int kGridBlock = 64;

__global__ __launch_bounds__(256) void grid_key_points_kernel(const float *__restrict__ xyz, int64_t n, GridView g,
                                                              uint32_t *__restrict__ keys) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int cx = grid_cell(xyz[3 * i], g.lo[0], g.inv_h, g.nx), cy = grid_cell(xyz[3 * i + 1], g.lo[1], g.inv_h, g.ny),
            cz = grid_cell(xyz[3 * i + 2], g.lo[2], g.inv_h, g.nz);
  keys[i] = (uint32_t)((cz * g.ny + cy) * g.nx + cx);
}

__global__ __launch_bounds__(256) void grid_gather_kernel(const float *__restrict__ xyz, const uint32_t *__restrict__ order,
                                                          const int32_t *__restrict__ labels, int64_t n,
                                                          float4 *__restrict__ pts) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const uint32_t i = order[j];
  const int32_t id = labels ? labels[i] : (int32_t)i;
  pts[j] = make_float4(xyz[3 * (int64_t)i], xyz[3 * (int64_t)i + 1], xyz[3 * (int64_t)i + 2], __int_as_float(id));
}

// start[c] = first position whose key is >= c (c = cells: n); sum of squared populations for the
// occupancy check
__global__ __launch_bounds__(256) void grid_start_kernel(const uint32_t *__restrict__ sorted_keys, int64_t n,
                                                         uint32_t cells, uint32_t *__restrict__ start,
                                                         unsigned long long *__restrict__ crowd) {
  const uint32_t c = blockIdx.x * 256u + threadIdx.x;
  if (c > cells) return;
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (sorted_keys[mid] < c) lo = mid + 1;
    else hi = mid;
  }
  start[c] = (uint32_t)lo;
  if (c == 0) start[-1] = 0u;                  // padding read (never used) by the 16-byte row loads
  if (c == cells) start[cells + 1] = (uint32_t)n;
  if (c < cells) {
    int64_t lo2 = lo, hi2 = n;
    while (lo2 < hi2) {
      const int64_t mid = (lo2 + hi2) >> 1;
      if (sorted_keys[mid] <= c) lo2 = mid + 1;
      else hi2 = mid;
    }
    const unsigned long long cnt = (unsigned long long)(lo2 - lo);
    if (cnt > 1) atomicAdd(crowd, cnt * cnt - cnt);  // ordered pairs sharing a cell
  }
}

// A point's certificate (GridView::cert): p is the nearest base point of EVERY query q whose computed DistSq(q, p) is
// below cert[p].  In real numbers: with rho the distance from p to its nearest other point and |q - p| < rho / 2, any
// other point p' has |q - p'| >= rho - |q - p| > |q - p|.  In float32, with margins that swallow every rounding on the
// way (a computed DistSq is within 3e-7 of the real one): cert = 0.24 L^2 where L^2 is a lower bound of rho^2 -- the
// smallest computed DistSq to the other points of the 3 x 3 x 3 cells around p, less 2e-6 of itself, or (0.999 h)^2
// where that is smaller (a point outside those cells is a cell's edge away).  |q - p| < 0.49 L then, every other point
// is more than 1.04 times as far, its computed DistSq more than 1.07 times the one of p: strictly larger, no tie.  A
// point with a twin (rho = 0) has cert 0: never certified.  The ICP loop keeps a pair on this alone (icp.hip).
__global__ __launch_bounds__(256) void grid_cert_kernel(GridView g, int64_t n, float *__restrict__ cert) {
  const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (f >= n) return;
  const float4 p = g.pts[f];
  const int cx = grid_cell(p.x, g.lo[0], g.inv_h, g.nx), cy = grid_cell(p.y, g.lo[1], g.inv_h, g.ny),
            cz = grid_cell(p.z, g.lo[2], g.inv_h, g.nz);
  const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.nx - 1);
  float dmin = __builtin_inff();
  for (int z = max(cz - 1, 0); z <= min(cz + 1, g.nz - 1); z++)
    for (int y = max(cy - 1, 0); y <= min(cy + 1, g.ny - 1); y++) {
      const uint32_t row = (uint32_t)((z * g.ny + y) * g.nx);
      const uint32_t s = g.start[row + (uint32_t)x0], e = g.start[row + (uint32_t)x1 + 1u];
      for (uint32_t k = s; k < e; k++) {
        if ((int64_t)k == f) continue;
        const float4 o = g.pts[k];
        const float dx = o.x - p.x, dy = o.y - p.y, dz = o.z - p.z;
        const float d = (dx * dx + dy * dy) + dz * dz;
        dmin = d < dmin ? d : dmin;  // (NaN never: the grid holds finite points only)
      }
    }
  // (a cell's edge in coordinates: the cell numbers are rounded products of up to max(nx, ny, nz), their rounding is
  // part of the margin)
  const float slack = 1.0f - 1.0e-3f - 4.0e-7f * (float)max(g.nx, max(g.ny, g.nz));
  const float edge = slack * g.h;
  float l2 = fminf(dmin - 2.0e-6f * dmin, edge * edge);
  if (!(l2 > 0.0f) || !(slack > 0.5f)) l2 = 0.0f;
  cert[__float_as_int(p.w)] = 0.24f * l2;  // by the point's id (ids are 0 .. n - 1: grid_build makes no certificates for labelled trees)
}

// One query per lane.  Certified answers are written; the rest is appended to walk_list (query
// indices) for the tree walk.
template <bool kWhy>
__global__ __launch_bounds__(kGridBlock) void grid_nearest_kernel(GridView g, const float *__restrict__ q,
                                                                  const int32_t *__restrict__ perm, int64_t nq,
                                                                  float max_range_sq, int32_t *__restrict__ out_id,
                                                                  float *__restrict__ out_dsq,
                                                                  int32_t *__restrict__ walk_list,
                                                                  uint32_t *__restrict__ walk_count,
                                                                  uint32_t *__restrict__ why_counts = nullptr) {
  // with `perm` the launch positions are in Morton order: an XCD takes a contiguous eighth of them
  // (pcgx_internal.h, xcd_tile); harmless without
  const uint32_t n_tiles = (uint32_t)((nq + kGridBlock - 1) / kGridBlock);
  const int64_t pos = (int64_t)xcd_tile(blockIdx.x, n_tiles) * kGridBlock + threadIdx.x;
  if (pos >= nq) return;
  const int64_t i = perm ? (int64_t)perm[pos] : pos;
  const float qx = q[3 * i], qy = q[3 * i + 1], qz = q[3 * i + 2];
  float4 best;
  float best_d;
  GridTrace tr;
  const GridVerdict v = grid_nearest(g, qx, qy, qz, max_range_sq, __builtin_inff(), best, best_d, kWhy ? &tr : nullptr);
  if (kWhy) {  // why_counts: [1..7] reasons, [8] / [9] point records / bound words read (64-bit, two words each)
    if (tr.why) atomicAdd(&why_counts[tr.why], 1u);
    atomicAdd(reinterpret_cast<unsigned long long *>(why_counts + 8), (unsigned long long)tr.points);
    atomicAdd(reinterpret_cast<unsigned long long *>(why_counts + 10), (unsigned long long)tr.words);
    if (tr.wave_slots) atomicAdd(reinterpret_cast<unsigned long long *>(why_counts + 12), (unsigned long long)tr.wave_slots);
    atomicAdd(&why_counts[32 + (tr.rounds9 < 0 ? 0 : min(tr.rounds9 + 1, 15))], 1u);
    for (int k = 0; k < 3; k++) {
      if (tr.slots_n[k]) atomicAdd(reinterpret_cast<unsigned long long *>(why_counts + 14 + 2 * k), (unsigned long long)tr.slots_n[k]);
      if (tr.points_n[k]) atomicAdd(reinterpret_cast<unsigned long long *>(why_counts + 20 + 2 * k), (unsigned long long)tr.points_n[k]);
    }
  }
  if (v == GRID_FOUND) {
    out_id[i] = __float_as_int(best.w);
    out_dsq[i] = best_d;
  } else if (v == GRID_NONE) {
    out_id[i] = -1;
    out_dsq[i] = max_range_sq;
  } else {
    walk_list[atomicAdd(walk_count, 1u)] = (int32_t)i;
  }
}

// ---- the queries of a large batch, partitioned once by a coarse cell ------------------------------------------------
// Large batches are searched in a spatial order (the lanes of a wave then read the same cells).  Round 1-3 made that
// order with a 16-bit Morton key, two radix passes over (key, index) pairs and a gather of every query through the
// resulting permutation: 47 of the 134 us a C2 call took.  Here ONE counting partition by an 8-bit cell (3 + 3 + 2
// bits over the base cloud's box) moves the queries themselves, {x, y, z, index} as 16-byte records: a histogram
// kernel, then a kernel that re-orders a 2048-query tile in LDS and writes every cell's run in one piece.  Where a
// tile's runs go needs no scan over the tiles: a returning atomic per (tile, cell) on the cell's cursor -- the order
// of the queries inside a cell is then whatever order the tiles arrive in, which no result depends on (the search is
// exact; ties and the like go to the walk by query index).  The search kernel reads its queries coalesced and
// scatters only the answers.
struct QueryBox {
  float lo[3], scale[3];
};
__device__ __forceinline__ uint32_t query_cell8(const QueryBox &b, float x, float y, float z) {
  // (points outside the box clamp to its faces; NaN -> 0)
  const uint32_t cx = (uint32_t)fminf(fmaxf((x - b.lo[0]) * b.scale[0], 0.0f), 7.0f);
  const uint32_t cy = (uint32_t)fminf(fmaxf((y - b.lo[1]) * b.scale[1], 0.0f), 7.0f);
  const uint32_t cz = (uint32_t)fminf(fmaxf((z - b.lo[2]) * b.scale[2], 0.0f), 3.0f);
  auto spread = [](uint32_t v) { return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4); };  // 3 bits -> every third bit
  return spread(cx) | (spread(cy) << 1) | (spread(cz) << 2);  // z has two bits: the key has eight
}
constexpr int kQpItems = 8, kQpTile = 256 * kQpItems;

__global__ __launch_bounds__(256) void qp_hist_kernel(const float *__restrict__ q, int64_t nq, QueryBox box,
                                                      uint32_t *__restrict__ totals) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kQpTile;
#pragma unroll
  for (int r = 0; r < kQpItems; r++) {
    const int64_t i = base + r * 256 + threadIdx.x;
    if (i < nq) atomicAdd(&h[query_cell8(box, q[3 * i], q[3 * i + 1], q[3 * i + 2])], 1u);
  }
  __syncthreads();
  // (eight copies of the totals and of the cursors, a tile uses copy blockIdx % 8: 489 tiles adding to ONE word per
  // cell were served one after the other, 13 us for this kernel where the pass over the queries takes 4)
  const uint32_t c = h[threadIdx.x];
  if (c) atomicAdd(&totals[(blockIdx.x & 7u) * 256 + threadIdx.x], c);
}

__global__ __launch_bounds__(256) void qp_scatter_kernel(const float *__restrict__ q, int64_t nq, QueryBox box,
                                                         const uint32_t *__restrict__ totals, uint32_t *__restrict__ cursors,
                                                         float4 *__restrict__ out) {
  __shared__ uint32_t lcnt[256], tile_pref[256], gbase[256], wsum[4], gwsum[4];
  __shared__ float4 srec[kQpTile];
  __shared__ uint8_t sdig[kQpTile];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  lcnt[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kQpTile;
  float x[kQpItems], y[kQpItems], z[kQpItems];
  uint32_t d[kQpItems], rank[kQpItems];
#pragma unroll
  for (int r = 0; r < kQpItems; r++) {
    const int64_t i = base + r * 256 + threadIdx.x;
    x[r] = y[r] = z[r] = 0.0f;
    if (i < nq) {
      x[r] = q[3 * i];
      y[r] = q[3 * i + 1];
      z[r] = q[3 * i + 2];
    }
  }
#pragma unroll
  for (int r = 0; r < kQpItems; r++) {
    const int64_t i = base + r * 256 + threadIdx.x;
    d[r] = query_cell8(box, x[r], y[r], z[r]);
    rank[r] = i < nq ? atomicAdd(&lcnt[d[r]], 1u) : 0u;
  }
  __syncthreads();
  {  // cell t: its place in the tile (exclusive scan of the tile's counts), its run's place in the output (the cell's
     // start = exclusive scan of the totals, + what earlier tiles have taken of the cell: the cursor)
    const int t = threadIdx.x;
    const uint32_t c = lcnt[t];
    // the cell's part of the output is cut into eight pieces, one per copy: piece k holds what the tiles with
    // blockIdx % 8 == k bring
    uint32_t tot = 0, before_mine = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const uint32_t v = totals[k * 256 + t];
      before_mine += k < (int)(blockIdx.x & 7u) ? v : 0u;
      tot += v;
    }
    const uint32_t inc = wave_incl_scan_u32(c), ginc = wave_incl_scan_u32(tot);
    if (lane == 63) {
      wsum[wave] = inc;
      gwsum[wave] = ginc;
    }
    __syncthreads();
    uint32_t wb = 0, gwb = 0;
    for (int w = 0; w < wave; w++) {
      wb += wsum[w];
      gwb += gwsum[w];
    }
    const uint32_t excl = wb + inc - c;
    tile_pref[t] = excl;
    const uint32_t taken = c ? atomicAdd(&cursors[(blockIdx.x & 7u) * 256 + t], c) : 0u;
    gbase[t] = (gwb + ginc - tot) + before_mine + taken - excl;  // dst = gbase[cell] + position in the tile
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < kQpItems; r++) {
    const int64_t i = base + r * 256 + threadIdx.x;
    if (i < nq) {
      const uint32_t pos = tile_pref[d[r]] + rank[r];
      srec[pos] = make_float4(x[r], y[r], z[r], __int_as_float((int)i));
      sdig[pos] = (uint8_t)d[r];
    }
  }
  __syncthreads();
  const int64_t rem = nq - base;
  const int count = rem < kQpTile ? (int)rem : kQpTile;
  for (int p = threadIdx.x; p < count; p += 256) out[(int64_t)gbase[sdig[p]] + p] = srec[p];
}

// The partition's output put into a finer order, tile by tile and in place: 2048 consecutive records (one coarse cell's
// as a rule, 3900 queries a cell at C2) counting-sorted in LDS by their place INSIDE the coarse cell, 8 x 8 x 8 sub-cells
// with x running fastest -- the base cloud's grid keeps a row of cells along x contiguous.  The search kernel's bound is
// the lines its waves miss in the vector L1 (tools/micro/vmem_mask.cpp: eight to ten cycles of a CU's texture path per
// missed line, whatever the lanes, loads in flight or waves per SIMD): sixty-four queries from anywhere in a 1.25 m
// coarse cell share next to nothing, sixty-four from two rows of sub-cells read the same few rows of the grid.
// No result depends on the order (exact search; ties go to the walk by query index).
#ifndef PCGX_QP_SUB
#define PCGX_QP_SUB 12
#endif
// (1024 threads, two records each: 489 tiles are two workgroups per CU, and four waves per CU hide nothing)
constexpr int kQpRefineThreads = 1024, kQpRefineItems = kQpTile / kQpRefineThreads;
__global__ __launch_bounds__(kQpRefineThreads) void qp_refine_kernel(float4 *__restrict__ rec, int64_t nq, QueryBox box) {
  constexpr int kSub = PCGX_QP_SUB, kSubZ = 2 * kSub, kCell = kSub * kSub * kSubZ;  // sub-cells of a coarse cell (twice as tall as wide)
  constexpr int kBins = ((2 * kCell + 1023) / 1024) * 1024, kT = kQpRefineThreads, kI = kQpRefineItems, kBinsPer = kBins / kT;
  __shared__ uint32_t bins[kBins];
  __shared__ float4 srec[kQpTile];
  __shared__ uint32_t wsum[kT / 64];
  __shared__ uint32_t s_first;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t base = (int64_t)blockIdx.x * kQpTile;
  const int64_t rem = nq - base;
  const int count = rem < kQpTile ? (int)rem : kQpTile;
#pragma unroll
  for (int k = 0; k < kBinsPer; k++) bins[k * kT + threadIdx.x] = 0;
  float4 r[kI];
#pragma unroll
  for (int k = 0; k < kI; k++) {
    const int p = k * kT + threadIdx.x;
    r[k] = p < count ? rec[base + p] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  }
  if (threadIdx.x == 0) s_first = query_cell8(box, r[0].x, r[0].y, r[0].z);  // (the tile's records ascend by coarse cell)
  __syncthreads();
  const uint32_t first = s_first;
  uint32_t key[kI], rank[kI];
#pragma unroll
  for (int k = 0; k < kI; k++) {
    const int p = k * kT + threadIdx.x;
    const float fx = fminf(fmaxf((r[k].x - box.lo[0]) * box.scale[0], 0.0f), 7.999f),
                fy = fminf(fmaxf((r[k].y - box.lo[1]) * box.scale[1], 0.0f), 7.999f),
                fz = fminf(fmaxf((r[k].z - box.lo[2]) * box.scale[2], 0.0f), 3.999f);
    const uint32_t sx = min((uint32_t)((fx - floorf(fx)) * (float)kSub), (uint32_t)kSub - 1u),
                   sy = min((uint32_t)((fy - floorf(fy)) * (float)kSub), (uint32_t)kSub - 1u),
                   sz = min((uint32_t)((fz - floorf(fz)) * (float)kSubZ), (uint32_t)kSubZ - 1u);
    const uint32_t rel = min(query_cell8(box, r[k].x, r[k].y, r[k].z) - first, 1u);  // (a tile over more than two cells: the rest share the second's bins)
    key[k] = rel * (uint32_t)kCell + (sz * (uint32_t)kSub + sy) * (uint32_t)kSub + sx;
    rank[k] = p < count ? atomicAdd(&bins[key[k]], 1u) : 0u;
  }
  __syncthreads();
  {  // bins -> their first places: thread t the bins kBinsPer t ..
    uint32_t c[kBinsPer], sum = 0;
#pragma unroll
    for (int k = 0; k < kBinsPer; k++) {
      c[k] = bins[threadIdx.x * kBinsPer + k];
      sum += c[k];
    }
    const uint32_t inc = wave_incl_scan_u32(sum);
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t run = inc - sum;
    for (int w = 0; w < wave; w++) run += wsum[w];
#pragma unroll
    for (int k = 0; k < kBinsPer; k++) {
      bins[threadIdx.x * kBinsPer + k] = run;
      run += c[k];
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kI; k++)
    if (k * kT + threadIdx.x < count) srec[bins[key[k]] + rank[k]] = r[k];
  __syncthreads();
  for (int p = threadIdx.x; p < count; p += kT) rec[base + p] = srec[p];
}

// One query record per lane (qp_scatter_kernel's output: cell after cell); the answer goes to the query's own index.
// Four waves per workgroup, and the SECOND scan of a search -- the cells of the 3 x 3 x 3 block outside the octant, needed
// by one query in four -- is not done by the lane that owns the query: the lanes that need one queue it in LDS (query,
// nine segments, what the octant gave), and the queue is scanned by as many lanes as it has entries, densely.  One query
// per lane all the way, the second scan cost a wave as many rounds as its slowest lane took with a quarter of its lanes
// switched on: 32 of a query's 56 lane-slots of point loads for 2.9 of its 14.7 records (tools/grid_probe.py) -- and
// the kernel is bound by the instructions it issues (vector ALU 66 % busy, the texture path 73 %), masked or not.
constexpr int kGridRecBlock = 256;
__global__ __launch_bounds__(kGridRecBlock) void grid_nearest_rec_kernel(GridView g, const float4 *__restrict__ qrec, int64_t nq,
                                                                         float max_range_sq, int32_t *__restrict__ out_id,
                                                                         float *__restrict__ out_dsq, int32_t *__restrict__ walk_list,
                                                                         uint32_t *__restrict__ walk_count,
                                                                         uint32_t *__restrict__ spent, int n_spent,
                                                                         uint32_t *__restrict__ next_walk_count) {
  __shared__ float s_q[3][kGridRecBlock];
  __shared__ uint32_t s_seg[18][kGridRecBlock];
  __shared__ float s_best[3][kGridRecBlock];  // d, d2, id bits: in by queue slot, out by owner
  __shared__ uint16_t s_owner[kGridRecBlock];
  __shared__ uint32_t s_n;
  // (the partition's totals and cursors are spent, the NEXT call's walk count is not in use yet: left at zero here,
  // so that no call starts with a memset launch -- Arena::zeroed_words)
  if (blockIdx.x == 0) {
    for (int k = threadIdx.x; k < n_spent; k += kGridRecBlock) spent[k] = 0u;
    if (threadIdx.x == 0) *next_walk_count = 0u;
  }
  if (threadIdx.x == 0) s_n = 0u;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const uint32_t n_tiles = (uint32_t)((nq + kGridRecBlock - 1) / kGridRecBlock);
  const int64_t pos = (int64_t)xcd_tile(blockIdx.x, n_tiles) * kGridRecBlock + threadIdx.x;  // an XCD: a contiguous eighth of the cells
  const bool active = pos < nq;
  const float4 r = active ? qrec[pos] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  const int32_t i = __float_as_int(r.w);
  GridSearch S;
  S.more = false;
  if (active) grid_nearest_begin(g, r.x, r.y, r.z, max_range_sq, __builtin_inff(), S, nullptr);
  const bool more = active && S.more;
  {
    const unsigned long long m = __ballot(more);
    uint32_t base = 0;
    if (m != 0ull) {  // uniform
      if (lane == __ffsll((long long)m) - 1) base = atomicAdd(&s_n, (uint32_t)__popcll(m));
      base = (uint32_t)__shfl((int)base, __ffsll((long long)m) - 1);
    }
    if (more) {
      const uint32_t e = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      s_q[0][e] = r.x; s_q[1][e] = r.y; s_q[2][e] = r.z;
#pragma unroll
      for (int j = 0; j < 9; j++) {
        s_seg[j][e] = S.seg_s[j];
        s_seg[9 + j][e] = S.seg_e[j];
      }
      s_best[0][e] = S.b.d; s_best[1][e] = S.b.d2; s_best[2][e] = S.b.p.w;
      s_owner[e] = (uint16_t)threadIdx.x;
    }
  }
  __syncthreads();
  const uint32_t n_more = s_n;
  float res_d = 0.0f, res_d2 = 0.0f, res_w = 0.0f;
  uint32_t res_owner = 0xffffffffu;
  if (threadIdx.x < n_more) {  // (at most one entry per thread: every thread queued at most one)
    const uint32_t e = threadIdx.x;
    uint32_t seg_s[9], seg_e[9];
#pragma unroll
    for (int j = 0; j < 9; j++) {
      seg_s[j] = s_seg[j][e];
      seg_e[j] = s_seg[9 + j][e];
    }
    GridBest b;
    b.p = make_float4(0.0f, 0.0f, 0.0f, s_best[2][e]);
    b.d = s_best[0][e];
    b.d2 = s_best[1][e];
    grid_scan_segments<9>(g, seg_s, seg_e, s_q[0][e], s_q[1][e], s_q[2][e], b, nullptr);
    res_d = b.d; res_d2 = b.d2; res_w = b.p.w;
    res_owner = s_owner[e];
  }
  __syncthreads();  // (the queue has been read: its slots take the answers, by owner)
  if (res_owner != 0xffffffffu) {
    s_best[0][res_owner] = res_d; s_best[1][res_owner] = res_d2; s_best[2][res_owner] = res_w;
  }
  __syncthreads();
  if (!active) return;
  if (more) {
    S.b.d = s_best[0][threadIdx.x]; S.b.d2 = s_best[1][threadIdx.x]; S.b.p.w = s_best[2][threadIdx.x];
  }
  float4 best;
  float best_d;
  const GridVerdict v = grid_nearest_end(g, r.x, r.y, r.z, max_range_sq, S, best, best_d, nullptr);
  if (v == GRID_FOUND) {
    out_id[i] = __float_as_int(best.w);
    out_dsq[i] = best_d;
  } else if (v == GRID_NONE) {
    out_id[i] = -1;
    out_dsq[i] = max_range_sq;
  } else {
    walk_list[atomicAdd(walk_count, 1u)] = i;
  }
}

void grid_free(pcgx_kdtree *t) {
  dev_cache_free(t->d_gpts);
  dev_cache_free(t->d_gstart);
  dev_cache_free(t->d_gcert);
  t->d_gpts = nullptr;
  t->d_gstart = nullptr;
  t->d_gcert = nullptr;
  t->grid_ok = false;
}

static int grid_mode() {  // PCGX_GRID=0: tree walk only; =2: grid even for crowded cells (tests)
  const char *e = getenv("PCGX_GRID");
  return e ? atoi(e) : 1;
}

bool grid_enabled(const pcgx_kdtree *t) { return t->grid_ok && grid_mode() != 0; }

// d_xyz: packed device xyz in accessor order; d_labels: optional ids the points report.  Finite
// coordinates only (the caller checks).  Leaves t->grid_ok false when a grid would not pay.
pcgx_status grid_build(pcgx_kdtree *t, const float *d_xyz, const int32_t *d_labels, hipStream_t st) {
  t->grid_ok = false;
  const int64_t n = t->n;
  if (n < 64 || grid_mode() == 0) return PCGX_OK;
  float ext[3];
  int dims = 0;
  double vol = 1.0;
  for (int k = 0; k < 3; k++) {
    ext[k] = t->bbox_hi[k] - t->bbox_lo[k];
    if (!(ext[k] >= 0.0f) || !(ext[k] < 1.0e30f)) return PCGX_OK;
    if (ext[k] > 0.0f) {
      dims++;
      vol *= (double)ext[k];
    }
  }
  if (dims == 0) return PCGX_OK;  // all points identical
  // ~1.5 points per cell of the occupied volume; a thin axis gets one layer of cells.  (Measured at
  // C2 / C4, points per cell -> kNN call / ICP step: 0.75 -> 0.179 ms / 31.8 us, 1.0 -> 0.155 / 32.8,
  // 1.5 -> 0.138 / 33.4, 2.0 -> 0.137 / 35.1, 3.0 -> 0.141 / 37.3: the hinted search wants small cells,
  // the one without a hint its answer inside the first octant.)
  double occ = 1.5;  // PCGX_GRID_OCC: tuning knob, points per cell of the bounding box's volume
  if (const char *e = getenv("PCGX_GRID_OCC")) {
    const double v = atof(e);
    if (v >= 0.25 && v <= 16.0) occ = v;
  }
  Arena &ar = ctx().arena;  // the caller has begun it
  uint32_t *keys[2], *vals[2];
  void *ws = nullptr;
  unsigned long long *d_crowd = nullptr;
  PCGX_TRY(ar.alloc_n((size_t)n, &keys[0]));
  PCGX_TRY(ar.alloc_n((size_t)n, &keys[1]));
  PCGX_TRY(ar.alloc_n((size_t)n, &vals[0]));
  PCGX_TRY(ar.alloc_n((size_t)n, &vals[1]));
  PCGX_TRY(ar.alloc_n(1, &d_crowd));
  {
    uint8_t *w = nullptr;
    PCGX_TRY(ar.alloc_n(radix_sort_workspace_bytes(n), &w));
    ws = w;
  }
  const unsigned nb = (unsigned)((n + 255) / 256);
  GridView g;
  int res = 0;
  double crowding = 0.0;
  // A cloud that fills its box evenly has ~occ other points in a point's cell.  Surfaces and other
  // thin shapes crowd the few cells they pass through: the cells are then made smaller (a surface's
  // crowding goes with h^2) until a point has ~3 neighbours in its cell, within 32 cells per point.
  for (int attempt = 0; attempt < 3; attempt++) {
    double h = pow(vol / ((double)n / occ), 1.0 / dims);
    memset(&g, 0, sizeof g);
    int64_t cells = 0;
    for (int tries = 0; tries < 64; tries++) {
      g.h = (float)h;
      g.inv_h = 1.0f / g.h;
      if (!(g.h > 0.0f) || !(g.inv_h < 1.0e30f)) break;
      int64_t d[3];
      bool ok = true;
      for (int k = 0; k < 3; k++) {
        d[k] = (int64_t)((double)ext[k] * (double)g.inv_h) + 1;
        ok = ok && d[k] < (1 << 20);
      }
      cells = ok ? d[0] * d[1] * d[2] : 0;
      if (ok && cells <= 32 * n + 4096 && cells <= ((int64_t)1 << 28)) {
        g.nx = (int32_t)d[0];
        g.ny = (int32_t)d[1];
        g.nz = (int32_t)d[2];
        break;
      }
      cells = 0;
      h *= 1.26;
    }
    if (cells == 0) {
      grid_free(t);
      return PCGX_OK;
    }
    for (int k = 0; k < 3; k++) g.lo[k] = t->bbox_lo[k];
    dev_cache_free(t->d_gstart);
    t->d_gstart = nullptr;
    // pad elements: one in front, one behind (GridQuad)
    hipError_t e = dev_cache_alloc((void **)&t->d_gstart, (size_t)(cells + 3) * sizeof(uint32_t));
    if (e != hipSuccess) {  // the grid is an accelerator, not a requirement: the tree is walked instead
      (void)hipGetLastError();
      grid_free(t);
      return PCGX_OK;
    }
    g.start = t->d_gstart + 1;
    PCGX_HIP_TRY(hipMemsetAsync(d_crowd, 0, sizeof(unsigned long long), st));
    hipLaunchKernelGGL(grid_key_points_kernel, dim3(nb), dim3(256), 0, st, d_xyz, n, g, keys[0]);
    int key_bits = 1;
    while (((int64_t)1 << key_bits) < cells) key_bits++;
    PCGX_TRY(radix_sort_pairs(keys, vals, n, key_bits, ws, &res, st, true));
    hipLaunchKernelGGL(grid_start_kernel, dim3((unsigned)((cells + 1 + 255) / 256)), dim3(256), 0, st,
                       (const uint32_t *)keys[res], n, (uint32_t)cells, t->d_gstart + 1, d_crowd);
    unsigned long long crowd = 0;
    PCGX_HIP_TRY(hipMemcpyAsync(&crowd, d_crowd, sizeof crowd, hipMemcpyDeviceToHost, st));
    PCGX_HIP_TRY(hipStreamSynchronize(st));
    // mean number of OTHER points in a point's cell
    crowding = (double)crowd / (double)n;
    double finer = occ * pow(3.0 / crowding, 1.5);
    const double occ_min = occ * (double)cells / (30.0 * (double)n);  // stays under 32 cells per point
    if (finer < occ_min) finer = occ_min;
    if (crowding <= 4.0 || attempt == 2 || finer > 0.7 * occ) break;
    occ = finer;
  }
  // (three records of padding: grid_scan_segments reads up to three positions behind a sequence's end, unused)
  hipError_t e = dev_cache_alloc((void **)&t->d_gpts, ((size_t)n + 3) * sizeof(float4));
  if (e != hipSuccess) {
    (void)hipGetLastError();
    grid_free(t);
    return PCGX_OK;
  }
  g.pts = t->d_gpts;
  hipLaunchKernelGGL(grid_gather_kernel, dim3(nb), dim3(256), 0, st, d_xyz, (const uint32_t *)vals[res], d_labels, n,
                     t->d_gpts);
  // the points' certificates (not required: without them every pair is searched for)
  g.cert = nullptr;
  if (d_labels == nullptr && dev_cache_alloc((void **)&t->d_gcert, (size_t)n * sizeof(float)) == hipSuccess) {
    hipLaunchKernelGGL(grid_cert_kernel, dim3(nb), dim3(256), 0, st, g, n, t->d_gcert);
    g.cert = t->d_gcert;
  } else {
    (void)hipGetLastError();
    t->d_gcert = nullptr;
  }
  PCGX_HIP_TRY(hipStreamSynchronize(st));  // vals[] live in the arena
  t->grid = g;
  // clouds that still crowd their cells (tight clusters) are better served by the tree
  t->grid_crowding = crowding;
  t->grid_ok = crowding <= 12.0 || grid_mode() == 2;
  if (!t->grid_ok) {
    dev_cache_free(t->d_gpts);
    dev_cache_free(t->d_gstart);
    dev_cache_free(t->d_gcert);
    t->d_gpts = nullptr;
    t->d_gstart = nullptr;
    t->d_gcert = nullptr;
  }
  return PCGX_OK;
}

// Exact-mode Nearest (MinDistSq == 0) of a batch: certified answers from the grid, the rest by the
// tree walk over the list the grid kernel leaves (its length stays on the device).
pcgx_status grid_launch_nearest(const pcgx_kdtree *t, const float *d_q, const int32_t *d_perm, int64_t nq,
                                float max_range_sq, int32_t *d_ids, float *d_dsq, hipStream_t st) {
  if (nq == 0) return PCGX_OK;
  Arena &ar = ctx().arena;  // begun by the caller
  int32_t *d_list = nullptr;
  uint32_t *d_count = nullptr;
  PCGX_TRY(ar.alloc_n((size_t)nq, &d_list));
  PCGX_TRY(ar.alloc_n(1, &d_count));
  PCGX_HIP_TRY(hipMemsetAsync(d_count, 0, sizeof(uint32_t), st));
  {
    ProfScope prof(PCGX_PROF_KNN_GRID, st);
    hipLaunchKernelGGL(grid_nearest_kernel<false>, dim3(xcd_grid((unsigned)((nq + kGridBlock - 1) / kGridBlock))),
                       dim3(kGridBlock), 0, st, t->grid, d_q, d_perm, nq, max_range_sq, d_ids, d_dsq, d_list, d_count,
                       (uint32_t *)nullptr);
  }
  PCGX_HIP_TRY(hipGetLastError());
  return launch_nearest_listed(t->view(), d_q, d_list, d_count, nq, max_range_sq, d_ids, d_dsq, st);
}

// The same for a large batch whose order is the library's to choose (PCGX_KNN_PRESORT): the queries partitioned by
// coarse cell, searched in that order.
pcgx_status grid_launch_nearest_partitioned(const pcgx_kdtree *t, const float *d_q, int64_t nq, float max_range_sq,
                                            int32_t *d_ids, float *d_dsq, hipStream_t st) {
  if (nq == 0) return PCGX_OK;
  Arena &ar = ctx().arena;  // begun by the caller
  int32_t *d_list = nullptr;
  // [8][256] totals, [8][256] cursors, then the walk counts of this call and of the next one (they take turns), each
  // on a line of its own: all zero when a call begins, left at zero by grid_nearest_rec_kernel
  uint32_t *d_words = nullptr;
  float4 *d_rec = nullptr;
  PCGX_TRY(ar.alloc_n((size_t)nq, &d_list));
  PCGX_TRY(ar.zeroed_words(4096 + 128, &d_words));
  PCGX_TRY(ar.alloc_n((size_t)nq, &d_rec));
  const uint32_t mine = ar.take_turn();
  uint32_t *d_walk_count = d_words + 4096 + 64 * mine, *d_walk_count_next = d_words + 4096 + 64 * (mine ^ 1u);
  QueryBox box;
  const float cells[3] = {8.0f, 8.0f, 4.0f};
  for (int k = 0; k < 3; k++) {
    const float ext = t->bbox_hi[k] - t->bbox_lo[k];
    box.lo[k] = t->bbox_lo[k] == t->bbox_lo[k] ? t->bbox_lo[k] : 0.0f;
    box.scale[k] = (ext > 0.0f && ext < 3.0e38f) ? cells[k] / ext : 0.0f;
  }
  const unsigned tiles = (unsigned)((nq + kQpTile - 1) / kQpTile);
  hipLaunchKernelGGL(qp_hist_kernel, dim3(tiles), dim3(256), 0, st, d_q, nq, box, d_words);
  hipLaunchKernelGGL(qp_scatter_kernel, dim3(tiles), dim3(256), 0, st, d_q, nq, box, (const uint32_t *)d_words, d_words + 2048, d_rec);
  static const bool refine = !(getenv("PCGX_KNN_REFINE") && getenv("PCGX_KNN_REFINE")[0] == '0');  // (measurement aid)
  if (refine) hipLaunchKernelGGL(qp_refine_kernel, dim3(tiles), dim3(kQpRefineThreads), 0, st, d_rec, nq, box);
  {
    ProfScope prof(PCGX_PROF_KNN_GRID, st);
    hipLaunchKernelGGL(grid_nearest_rec_kernel, dim3(xcd_grid((unsigned)((nq + kGridRecBlock - 1) / kGridRecBlock))), dim3(kGridRecBlock), 0, st,
                       t->grid, (const float4 *)d_rec, nq, max_range_sq, d_ids, d_dsq, d_list, d_walk_count, d_words, 4096,
                       d_walk_count_next);
  }
  // (a call that fails half-way may leave the counters in any state: they are dropped, the next call gets fresh ones)
  const hipError_t le = hipGetLastError();
  if (le != hipSuccess) {
    ar.release_words();
    return fail(PCGX_E_HIP, "Nearest batch: launch failed: %s", hipGetErrorString(le));
  }
  const pcgx_status rc = launch_nearest_listed(t->view(), d_q, d_list, d_walk_count, nq, max_range_sq, d_ids, d_dsq, st);
  if (rc != PCGX_OK) ar.release_words();
  return rc;
}

}  // namespace pcgx

extern "C" pcgx_status pcgx_debug_grid_cert(const pcgx_kdtree *t, float *cert, int64_t n) {
  PCGX_API_LOCK();
  if (!t || !cert || n != t->n) return pcgx::fail(PCGX_E_INVALID, "pcgx_debug_grid_cert: bad argument");
  if (!pcgx::grid_enabled(t) || !t->d_gcert) return pcgx::fail(PCGX_E_INVALID, "pcgx_debug_grid_cert: the tree has no certificates");
  PCGX_TRY(pcgx::ensure_init());
  PCGX_HIP_TRY(hipMemcpy(cert, t->d_gcert, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
  return PCGX_OK;
}

// Debug / tuning aid (not part of the drop-in surface): how many of the queries the grid pass leaves
// to the tree walk, and the tree's grid parameters.  out[0] = queries left to the walk, out[1] =
// cells, out[2] = crowding * 1000, out[3] = grid enabled, out[4 + k] = queries with reason k (knn_grid.h,
// GridTrace::why; k = 1..7), out[12] / out[13] = point records / cell-bound words the pass read.
extern "C" pcgx_status pcgx_debug_grid_stats(const pcgx_kdtree *t, const float *d_q, int64_t nq, float max_range,
                                             int64_t out[14]) {
  PCGX_API_LOCK();
  if (!t || !out || nq < 0 || (nq > 0 && !d_q)) return pcgx::fail(PCGX_E_INVALID, "pcgx_debug_grid_stats: bad argument");
  PCGX_TRY(pcgx::ensure_init());
  for (int k = 0; k < 14; k++) out[k] = 0;
  out[2] = (int64_t)(t->grid_crowding * 1000.0);
  out[3] = pcgx::grid_enabled(t) ? 1 : 0;
  if (!out[3]) return PCGX_OK;
  out[1] = (int64_t)t->grid.nx * t->grid.ny * t->grid.nz;
  if (nq == 0) return PCGX_OK;
  hipStream_t st = pcgx::ctx().stream;
  pcgx::Arena &ar = pcgx::ctx().arena;
  PCGX_TRY(ar.begin(st));
  int32_t *d_list = nullptr, *d_ids = nullptr;
  float *d_dsq = nullptr;
  uint32_t *d_count = nullptr;  // [0] walk count, [8 + k] reasons, [16..19] 64-bit totals
  PCGX_TRY(ar.alloc_n((size_t)nq, &d_list));
  PCGX_TRY(ar.alloc_n((size_t)nq, &d_ids));
  PCGX_TRY(ar.alloc_n((size_t)nq, &d_dsq));
  PCGX_TRY(ar.alloc_n(64, &d_count));
  PCGX_HIP_TRY(hipMemsetAsync(d_count, 0, 64 * sizeof(uint32_t), st));
  hipLaunchKernelGGL(pcgx::grid_nearest_kernel<true>, dim3(pcgx::xcd_grid((unsigned)((nq + pcgx::kGridBlock - 1) / pcgx::kGridBlock))),
                     dim3(pcgx::kGridBlock), 0, st, t->grid, d_q, (const int32_t *)nullptr, nq, max_range * max_range,
                     d_ids, d_dsq, d_list, d_count, d_count + 8);
  uint32_t c[64];
  PCGX_HIP_TRY(hipMemcpyAsync(c, d_count, sizeof c, hipMemcpyDeviceToHost, st));
  PCGX_HIP_TRY(hipStreamSynchronize(st));
  if (getenv("PCGX_GRID_TRACE_PRINT")) {
    fprintf(stderr, "9-segment scan per query: none %.4f, rounds of 4:", c[40] / (double)nq);
    for (int k = 1; k < 16; k++) fprintf(stderr, " %d:%.4f", k - 1, c[40 + k] / (double)nq);
    fprintf(stderr, "\n");
  }
  if (getenv("PCGX_GRID_TRACE_PRINT"))
    for (int k = 0; k < 3; k++)
      fprintf(stderr, "scan kind %d (%d segments): records/query %.2f lane-slots/query %.2f\n", k, k == 0 ? 4 : (k == 1 ? 9 : 5),
              (double)(((uint64_t)c[29 + 2 * k] << 32) | c[28 + 2 * k]) / (double)nq,
              (double)(((uint64_t)c[23 + 2 * k] << 32) | c[22 + 2 * k]) / (double)nq);
  out[0] = (int64_t)c[0];
  for (int k = 1; k < 8; k++) out[4 + k] = (int64_t)c[8 + k];
  out[12] = (int64_t)(((uint64_t)c[17] << 32) | c[16]);
  out[13] = (int64_t)(((uint64_t)c[19] << 32) | c[18]);
  out[4] = (int64_t)(((uint64_t)c[21] << 32) | c[20]);  // lane-slots the scan loops ran (64 per round of 4 records, idle lanes included)
  return PCGX_OK;
}
