// range.hip -- batched KDTree.Range (radius search) on gfx950.
//
// Reference: pc/storage/kdtree/kdtree.go:148-197 (Range / rangeImpl) + :415-427
// (neighborSorter).  rangeImpl is the same in-order walk as nearestImpl with a FIXED bound:
//   leaf:   append if dsq < maxRange^2                                   (:166-169)
//   unwind: skip the pivot and the far side if fp*fp > maxRange^2        (:173-177)
//           append the pivot if dsq < maxRange^2                         (:178-181)
//           recurse into the other child                                 (:182-195)
// then the neighbours are sorted by DistSq (:159).  Go's sort is unstable, so the order of
// equal DistSq is unspecified in the reference; here (and in the oracle) ties keep the
// discovery order of the walk.
//
// On a handle with a uniform grid (knn_grid.h) the neighbours are collected from the grid's cells instead of by
// the walk: which points have DistSq < maxRange^2 does not depend on the order they are visited in (the walk's
// pruning never drops one: kdtree.go:173-177 skips a side only when the plane alone is farther), the cells of
// grid_cover hold all of them, and reading a few dozen consecutive records per row replaces ~100 dependent node
// fetches per query.  The ORDER the walk would have found them in matters only among equal DistSq of one query;
// those runs are put into it afterwards (range_tie_*_kernel: the walk is an in-order traversal that takes the
// query's side of every node first, so a point's place in it follows from its node's BFS index).
//
// Two entry points because the result length is data dependent: pcgx_kdtree_range_count
// (walk, count) and pcgx_kdtree_range_fill (walk again, write at the caller's offsets, then
// one stable sort of the whole batch by (query, DistSq) with the radix sort of sort.hip).
// One query per lane; 4-byte frames [level][thread] in LDS as in knn_walk.h.
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "knn_grid.h"
#include "range_walk.h"

namespace pcgx {

constexpr int kRangeBlock = kRangeWalkBlock;
constexpr int64_t kRangePresortMin = 16384;  // batches from this size on are walked in Morton order

// kFill == false: counts[i] = number of neighbours.  kFill == true: neighbours of query i are
// written from offsets[i] in discovery order: {point id, DistSq bits, query index}.
template <bool kFill>
__global__ __launch_bounds__(kRangeBlock) void range_kernel(TreeView tv, const float *__restrict__ q,
                                                            const int32_t *__restrict__ perm, int64_t nq,
                                                            float bound, int64_t *__restrict__ counts,
                                                            const int64_t *__restrict__ offsets, int64_t total,
                                                            int32_t *__restrict__ out_id,
                                                            uint32_t *__restrict__ out_key,
                                                            uint32_t *__restrict__ out_query) {
  extern __shared__ uint32_t s_stack[];
  // launch positions are in Morton order when `perm` is given: an XCD takes a contiguous eighth of
  // them, so its L2 holds that region's part of the tree (pcgx_internal.h, xcd_tile).  (Two queries
  // per lane with their walks interleaved -- two node fetches in flight -- measured 1.4x SLOWER at
  // 200k queries: half as many waves, each with twice the instructions; the walk is bound by how
  // fast one wave issues its dependent instructions, and there are too few waves as it is.)
  const uint32_t n_tiles = (uint32_t)((nq + kRangeBlock - 1) / kRangeBlock);
  const int64_t pos = (int64_t)xcd_tile(blockIdx.x, n_tiles) * kRangeBlock + threadIdx.x;
  if (pos >= nq) return;
  // perm (optional): launch position -> query index (Morton order: the lanes of a wave walk
  // neighbouring sub-trees); everything is written at the query's own index
  const int64_t i = perm ? (int64_t)perm[pos] : pos;
  const float qx = q[3 * i], qy = q[3 * i + 1], qz = q[3 * i + 2];
  int64_t found = 0;
  // never write outside the slice the caller's offsets give this query (they may be wrong)
  const int64_t out0 = kFill ? offsets[i] : 0;
  const int64_t cap = kFill ? offsets[i + 1] - out0 : 0;
  const bool slice_ok = kFill && out0 >= 0 && cap >= 0 && out0 + cap <= total;
  range_walk(tv, s_stack + threadIdx.x, kRangeBlock, qx, qy, qz, bound, [&](int32_t id, float d) {
    if (kFill && slice_ok && found < cap) {
      out_id[out0 + found] = id;
      out_key[out0 + found] = __float_as_uint(d);  // d >= 0: the bit pattern orders like the value
      out_query[out0 + found] = (uint32_t)i;
    }
    ++found;
  });
  if (!kFill) counts[i] = found;
}

// The same on the grid: every point of the cells grid_cover names, row by row.  Hits in cell order.
template <bool kFill>
__global__ __launch_bounds__(kRangeBlock) void range_grid_kernel(GridView g, const float *__restrict__ q,
                                                                 const int32_t *__restrict__ perm, int64_t nq, float bound,
                                                                 int64_t *__restrict__ counts,
                                                                 const int64_t *__restrict__ offsets, int64_t total,
                                                                 uint4 *__restrict__ out_rec) {
  const uint32_t n_tiles = (uint32_t)((nq + kRangeBlock - 1) / kRangeBlock);
  const int64_t pos = (int64_t)xcd_tile(blockIdx.x, n_tiles) * kRangeBlock + threadIdx.x;
  if (pos >= nq) return;
  const int64_t i = perm ? (int64_t)perm[pos] : pos;
  const float qx = q[3 * i], qy = q[3 * i + 1], qz = q[3 * i + 2];
  int64_t found = 0;
  const int64_t out0 = kFill ? offsets[i] : 0;
  const int64_t cap = kFill ? offsets[i + 1] - out0 : 0;
  const bool slice_ok = kFill && out0 >= 0 && cap >= 0 && out0 + cap <= total;
  auto take = [&](const float4 &p) {
    const float dx = p.x - qx, dy = p.y - qy, dz = p.z - qz;
    const float d = (dx * dx + dy * dy) + dz * dz;  // the reference's expression (mat/vec3.go:18-20,38-40)
    if (d < bound) {  // kdtree.go:166,178
      // one 16-byte store per neighbour (every lane writes into a slice of its own: three 4-byte stores into
      // three arrays made the fill 2.3x the count); range_split_kernel spreads them afterwards, coalesced
      if (kFill && slice_ok && found < cap)
        out_rec[out0 + found] = make_uint4(__float_as_uint(p.w), __float_as_uint(d), (uint32_t)i, 0u);
      ++found;
    }
  };
  // (a NaN bound or query: the box is some cell or other and no distance compares below the bound, as in the walk)
  const GridBox box = grid_cover(g, qx, qy, qz, bound);
  // A row of thousands of records (one site of the cloud taken a hundred thousand times) is not one lane's work -- 25 ms
  // of dependent loop for 100k records, twice: the lane notes up to two such rows and the WAVE scans them together
  // below.  The order hits are found in is free here: among equal DistSq of a query it is made afterwards
  // (range_tie_*_kernel), everything else is sorted by DistSq.
  constexpr uint32_t kFatRow = 4096u;
  uint32_t fat_f0 = 0u, fat_e0 = 0u, fat_f1 = 0u, fat_e1 = 0u;
  int nfat = 0;
  for (int z = box.z0; z <= box.z1; z++) {
    for (int y = box.y0; y <= box.y1; y++) {
      const uint32_t row = ((uint32_t)z * (uint32_t)g.ny + (uint32_t)y) * (uint32_t)g.nx;
      uint32_t f = g.start[row + (uint32_t)box.x0];
      const uint32_t e = g.start[row + (uint32_t)box.x1 + 1u];
      if (e - f >= kFatRow && e > f && nfat < 2) {
        if (nfat == 0) { fat_f0 = f; fat_e0 = e; }
        else { fat_f1 = f; fat_e1 = e; }
        nfat++;
        continue;
      }
      for (; f + 4u <= e; f += 4u) {  // four records in flight
        const float4 p0 = g.pts[f], p1 = g.pts[f + 1u], p2 = g.pts[f + 2u], p3 = g.pts[f + 3u];
        take(p0); take(p1); take(p2); take(p3);
      }
      for (; f < e; f++) take(g.pts[f]);
    }
  }
  if (__ballot(nfat > 0) != 0ull) {  // (the lanes that are still here: those with a query)
    const int lane = (int)(threadIdx.x & 63u);
    const unsigned long long act = __ballot(true), below = act & ((1ull << lane) - 1ull);
    const uint32_t nact = (uint32_t)__popcll(act), myrank = (uint32_t)__popcll(below);
    for (int k = 0; k < 2; k++) {
      unsigned long long owners = __ballot(nfat > k);
      while (owners != 0ull) {  // uniform
        const int owner = __builtin_ctzll(owners);
        owners &= owners - 1ull;
        const float ox = __shfl(qx, owner), oy = __shfl(qy, owner), oz = __shfl(qz, owner);
        const uint32_t rf = __shfl(k == 0 ? fat_f0 : fat_f1, owner), re = __shfl(k == 0 ? fat_e0 : fat_e1, owner);
        const long long o_out0 = __shfl((long long)out0, owner), o_cap = __shfl((long long)cap, owner),
                        o_found = __shfl((long long)found, owner);
        const int o_ok = __shfl(slice_ok ? 1 : 0, owner);
        const uint32_t o_i = __shfl((uint32_t)i, owner);
        long long add = 0;
        for (uint32_t r0 = rf; r0 < re; r0 += nact) {  // uniform
          const uint32_t r = r0 + myrank;
          bool hit = false;
          float4 p = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
          float d = 0.0f;
          if (r < re) {
            p = g.pts[r];
            const float dx = p.x - ox, dy = p.y - oy, dz = p.z - oz;
            d = (dx * dx + dy * dy) + dz * dz;
            hit = d < bound;
          }
          const unsigned long long hb = __ballot(hit);
          if (kFill && hit && o_ok) {
            const long long at = o_found + add + (long long)__popcll(hb & ((1ull << lane) - 1ull));
            if (at < o_cap) out_rec[o_out0 + at] = make_uint4(__float_as_uint(p.w), __float_as_uint(d), o_i, 0u);
          }
          add += (long long)__popcll(hb);
        }
        if (lane == owner) found += (int64_t)add;
      }
    }
  }
  if (!kFill) counts[i] = found;
}

// {id, DistSq bits, query} records -> the three arrays the sort works on (a slot no query wrote keeps the
// poisoned id, query 0)
__global__ __launch_bounds__(256) void range_split_kernel(const uint4 *__restrict__ rec, int64_t total, int32_t *__restrict__ out_id,
                                                          uint32_t *__restrict__ out_key, uint32_t *__restrict__ out_query,
                                                          int32_t *__restrict__ bad) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= total) return;
  const uint4 v = rec[j];
  // a slot no query wrote: the caller's offsets leave room the counts do not fill.  Said HERE: behind the sort such a
  // slot stands among query 0's neighbours at DistSq 0, where the tie pass may move it onto a real one (ADVICE)
  if (v.x == 0xffffffffu) *bad = 1;
  out_id[j] = (int32_t)v.x;
  out_key[j] = v.x == 0xffffffffu ? 0u : v.y;
  out_query[j] = v.x == 0xffffffffu ? 0u : v.z;
}

// inv[id] = BFS index of the node that holds point id
__global__ __launch_bounds__(256) void range_invert_nodes_kernel(TreeView tv, uint32_t *__restrict__ inv) {
  const uint32_t b = blockIdx.x * 256u + threadIdx.x;
  if (b < 1u || b >= (1u << tv.depth)) return;
  const uint32_t size = node_size(b, 31 - __clz((int)b), (uint32_t)tv.n + 1u);
  if (size < 1u || size > (uint32_t)tv.n) return;  // no such node
  const uint32_t id = __float_as_uint(node_at(tv.nodes, b).w);
  if (id < (uint32_t)tv.n) inv[id] = b;
}

// Place of node b in the walk of query q (kdtree.go:148-197): the walk of a sub-tree takes the child on the
// query's side first (searchLeafNode's choice, :208-216), then the node itself, then the other child.  One base-3
// digit per level from the root: 0 = on the query's side, 1 = the node itself, 2 = on the other side.
__device__ __forceinline__ unsigned long long range_walk_place(const TreeView &tv, uint32_t b, float qx, float qy, float qz) {
  const int d = 31 - __clz((int)b);
  const uint32_t m1 = (uint32_t)tv.n + 1u;
  unsigned long long place = 0ull;
  for (int j = 0; j < d; j++) {
    const uint32_t anc = b >> (d - j);
    const uint32_t bit = (b >> (d - j - 1)) & 1u;
    const float pv = node_comp(tv.nodes, anc, j % 3), qv = sel3(j % 3, qx, qy, qz);
    const bool go_left = node_size(anc, j, m1) == 2u || pv > qv;
    place = place * 3ull + ((bit == 0u) == go_left ? 0ull : 2ull);
  }
  place = place * 3ull + 1ull;
  for (int j = d + 1; j < tv.depth; j++) place *= 3ull;
  return place;
}

constexpr int kTieRunLimit = 128;
// slot s of the sorted batch is tied with a neighbour (same query, same DistSq): its place in the walk
__global__ __launch_bounds__(256) void range_tie_place_kernel(TreeView tv, const uint32_t *__restrict__ inv,
                                                              const float *__restrict__ q, const uint32_t *__restrict__ query_of,
                                                              const uint32_t *__restrict__ key, const uint32_t *__restrict__ ids,
                                                              int64_t total, unsigned long long *__restrict__ place,
                                                              int32_t *__restrict__ long_run) {
  const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= total) return;
  const uint32_t qi = query_of[s], k = key[s];
  // (the slots are sorted by (query, DistSq): the same pair kTieRunLimit slots back means one run of ties longer than that
  // -- range_tie_sort_kernel's per-slot scan is quadratic in a run; the host then orders the runs by sorting)
  if (s >= kTieRunLimit && query_of[s - kTieRunLimit] == qi && key[s - kTieRunLimit] == k) *long_run = 1;
  const bool tied = (s > 0 && query_of[s - 1] == qi && key[s - 1] == k) || (s + 1 < total && query_of[s + 1] == qi && key[s + 1] == k);
  unsigned long long v = 0ull;
  if (tied) {
    const uint32_t id = ids[s];
    if (id < (uint32_t)tv.n) v = range_walk_place(tv, inv[id], q[3 * (size_t)qi], q[3 * (size_t)qi + 1], q[3 * (size_t)qi + 2]);
  }
  place[s] = v;
}

// Every slot of a run of ties finds its own place in the run: the number of slots of the run that the walk reaches
// earlier.  (Independent loads, a thread per slot: a lattice cloud with every site taken several times gives runs
// of a hundred, and a run sorted by one thread was a chain of dependent memory accesses a millisecond long.)
__global__ __launch_bounds__(256) void range_tie_sort_kernel(const uint32_t *__restrict__ query_of, const uint32_t *__restrict__ key,
                                                             int64_t total, const unsigned long long *__restrict__ place,
                                                             const uint32_t *__restrict__ ids, uint32_t *__restrict__ ids_out,
                                                             int32_t *__restrict__ bad) {
  const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= total) return;
  const unsigned long long mine = place[s];
  const uint32_t id = ids[s];
  if (id == 0xffffffffu) {  // a slot nothing was written to: the caller's offsets do not match the counts.  Said here,
    *bad = 1;               // where it is still certain -- such slots carry query 0 / key 0 and, mixed into a run of query
    ids_out[s] = id;        // 0's ties at DistSq 0, could be written over before range_widen_check_kernel looks
    return;
  }
  if (mine == 0ull) {  // not tied (a tied slot's place has its own digit 1 in it)
    ids_out[s] = id;
    return;
  }
  const uint32_t qi = query_of[s], k = key[s];
  int64_t a = s, before = 0;
  while (a > 0 && query_of[a - 1] == qi && key[a - 1] == k) {
    a--;
    before += place[a] <= mine ? 1 : 0;  // (equal places do not occur: one node, one place; kept stable anyway)
  }
  for (int64_t x = s + 1; x < total && query_of[x] == qi && key[x] == k; x++) before += place[x] < mine ? 1 : 0;
  ids_out[a + before] = id;
}

// 32 bits of the walk's place of the slot at index[j] (nullptr: j) -- the keys of the long tie runs' sort
__global__ __launch_bounds__(256) void range_place_bits_kernel(const unsigned long long *__restrict__ place, const uint32_t *__restrict__ index,
                                                               int64_t n, int shift, uint32_t *__restrict__ out) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j < n) out[j] = (uint32_t)(place[index ? index[j] : (uint32_t)j] >> shift);
}

__global__ __launch_bounds__(256) void range_iota_kernel(uint32_t *__restrict__ a, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) a[i] = (uint32_t)i;
}

__global__ __launch_bounds__(256) void range_gather_u32_kernel(const uint32_t *__restrict__ src,
                                                               const uint32_t *__restrict__ index, int64_t n,
                                                               uint32_t *__restrict__ dst) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j < n) dst[j] = src[index[j]];
}

}  // namespace pcgx

using namespace pcgx;


namespace pcgx {
__global__ __launch_bounds__(256) void range_widen_check_kernel(const uint32_t *__restrict__ in, int64_t n, int64_t *__restrict__ out,
                                                                int32_t *__restrict__ bad) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int32_t v = (int32_t)in[i];
  out[i] = (int64_t)v;
  if (v < 0) *bad = 1;
}
}  // namespace pcgx

namespace pcgx {
// PCGX_RANGE_WALK=1: the tree walk even where the handle has a grid (measurements, tests of the walk)
static bool range_on_grid(const pcgx_kdtree *t) {
  const char *e = getenv("PCGX_RANGE_WALK");  // (read per call: the tests switch between the two)
  return t->grid_ok && !(e && *e && *e != '0');
}

static pcgx_status range_inverse_map(const pcgx_kdtree *tc, const uint32_t **out, hipStream_t st) {
  pcgx_kdtree *t = const_cast<pcgx_kdtree *>(tc);  // made once per handle, on first use
  std::lock_guard<std::mutex> lock(t->mu);
  if (!t->d_inv) {
    uint32_t *p = nullptr;
    hipError_t e = dev_cache_alloc((void **)&p, (size_t)(t->n > 0 ? t->n : 1) * sizeof(uint32_t));
    if (e != hipSuccess) return fail(PCGX_E_OOM, "range: hipMalloc for the id -> node map failed: %s", hipGetErrorString(e));
    e = hipMemsetAsync(p, 0, (size_t)(t->n > 0 ? t->n : 1) * sizeof(uint32_t), st);
    const TreeView tv = t->view();
    const unsigned slots = 1u << tv.depth;
    if (e == hipSuccess) {
      hipLaunchKernelGGL(range_invert_nodes_kernel, dim3((slots + 255u) / 256u), dim3(256), 0, st, tv, p);
      e = hipGetLastError();
    }
    if (e != hipSuccess) {
      dev_cache_free(p);
      return fail(PCGX_E_HIP, "range: building the id -> node map failed: %s", hipGetErrorString(e));
    }
    t->d_inv = p;
  }
  *out = t->d_inv;
  return PCGX_OK;
}
}  // namespace pcgx

extern "C" pcgx_status pcgx_kdtree_range_count(const pcgx_kdtree *t, const float *q, int64_t nq, float max_range,
                                               int64_t *counts) {
  PCGX_API_CALL();
  if (!t || nq < 0 || (nq > 0 && (!q || !counts))) return fail(PCGX_E_INVALID, "pcgx_kdtree_range_count: bad argument");
  if (nq == 0) return PCGX_OK;
  PCGX_TRY(ensure_init());
  if (nq <= xtree_host_walk_max() && !t->points.empty()) {  // a few points: the same walk on the host (knn_explicit.hip)
    (void)xtree_host_range(t, q, nq, max_range, counts, nullptr, nullptr, nullptr);
    return PCGX_OK;
  }
  const pcgx_kdtree *outer = t;  // a handle with deletions walks the reference's patched tree (knn_explicit.hip)
  const bool patched = outer->n_deleted > 0;
  bool empty = false;
  if (!patched) PCGX_TRY(resolve_tree(t, &t, &empty));
  hipStream_t st = ctx().stream;
  float *d_q = nullptr;
  int64_t *d_c = nullptr;
  PCGX_TRY(ctx().host_arena.begin(st));
  PCGX_TRY(ctx().host_arena.alloc_n((size_t)nq * 3, &d_q));
  PCGX_TRY(ctx().host_arena.alloc_n((size_t)nq, &d_c));
  PCGX_TRY(staged_upload(d_q, q, (size_t)nq * 12, st));
  const TreeView tv = t->view();
  const size_t lds = walk_stack_bytes(tv, kRangeBlock);
  int32_t *perm = nullptr;
  if (nq >= kRangePresortMin) {
    PCGX_TRY(ctx().arena.begin(st));
    PCGX_TRY(ctx().arena.alloc_n((size_t)nq, &perm));
    PCGX_TRY(morton_order(d_q, nq, t->bbox_lo, t->bbox_hi, perm, st));
  }
  if (patched)
    PCGX_TRY(xtree_launch_range(outer, false, d_q, perm, nq, max_range * max_range, d_c, nullptr, 0, nullptr, nullptr,
                                nullptr, st));
  else if (range_on_grid(t))
    hipLaunchKernelGGL(range_grid_kernel<false>, dim3(xcd_grid((unsigned)((nq + kRangeBlock - 1) / kRangeBlock))), dim3(kRangeBlock),
                       0, st, t->grid, (const float *)d_q, (const int32_t *)perm, nq, max_range * max_range, d_c, nullptr, 0,
                       (uint4 *)nullptr);
  else
    hipLaunchKernelGGL(range_kernel<false>, dim3(xcd_grid((unsigned)((nq + kRangeBlock - 1) / kRangeBlock))), dim3(kRangeBlock),
                       lds, st, tv, (const float *)d_q, (const int32_t *)perm, nq, max_range * max_range, d_c, nullptr, 0,
                       nullptr, nullptr, nullptr);
  PCGX_HIP_TRY(hipGetLastError());
  PCGX_TRY(staged_download(counts, d_c, (size_t)nq * 8, st));
  return PCGX_OK;
}

extern "C" pcgx_status pcgx_kdtree_range_fill(const pcgx_kdtree *t, const float *q, int64_t nq, float max_range,
                                              const int64_t *offsets, int64_t *ids, float *dist_sq) {
  PCGX_API_CALL();
  if (!t || nq < 0 || (nq > 0 && (!q || !offsets))) return fail(PCGX_E_INVALID, "pcgx_kdtree_range_fill: bad argument");
  if (nq == 0) return PCGX_OK;
  const int64_t total = offsets[nq];
  if (total < 0 || offsets[0] != 0) return fail(PCGX_E_INVALID, "pcgx_kdtree_range_fill: offsets must start at 0");
  if (total == 0) return PCGX_OK;
  if (!ids || !dist_sq) return fail(PCGX_E_INVALID, "pcgx_kdtree_range_fill: NULL output");
  if (total > 0x7fffffffll) return fail(PCGX_E_TOO_LARGE, "pcgx_kdtree_range_fill: more than 2^31-1 neighbours in one batch");
  PCGX_TRY(ensure_init());
  if (nq <= xtree_host_walk_max() && !t->points.empty()) {  // a few points: the same walk on the host (knn_explicit.hip)
    for (int64_t i = 0; i < nq; i++)
      if (offsets[i + 1] < offsets[i]) return fail(PCGX_E_INVALID, "pcgx_kdtree_range_fill: offsets do not match the neighbour counts");
    if (!xtree_host_range(t, q, nq, max_range, nullptr, offsets, ids, dist_sq))
      return fail(PCGX_E_INVALID, "pcgx_kdtree_range_fill: offsets do not match the neighbour counts");
    return PCGX_OK;
  }
  const pcgx_kdtree *outer = t;
  const bool patched = outer->n_deleted > 0;
  bool empty = false;
  if (!patched) PCGX_TRY(resolve_tree(t, &t, &empty));
  hipStream_t st = ctx().stream;
  Arena &ar = ctx().arena;
  PCGX_TRY(ar.begin(st));
  float *d_q = nullptr;
  int64_t *d_off = nullptr;
  int32_t *d_id = nullptr;
  uint32_t *keys[2] = {nullptr, nullptr}, *vals[2] = {nullptr, nullptr}, *d_query = nullptr, *d_key = nullptr,
           *d_out_id = nullptr, *d_out_key = nullptr;
  void *ws = nullptr;
  PCGX_TRY(ar.alloc_n((size_t)nq * 3, &d_q));
  PCGX_TRY(ar.alloc_n((size_t)nq + 1, &d_off));
  PCGX_TRY(ar.alloc_n((size_t)total, &d_id));
  PCGX_TRY(ar.alloc_n((size_t)total, &d_key));
  PCGX_TRY(ar.alloc_n((size_t)total, &d_query));
  PCGX_TRY(ar.alloc_n((size_t)total, &keys[0]));
  PCGX_TRY(ar.alloc_n((size_t)total, &keys[1]));
  PCGX_TRY(ar.alloc_n((size_t)total, &vals[0]));
  PCGX_TRY(ar.alloc_n((size_t)total, &vals[1]));
  PCGX_TRY(ar.alloc_n((size_t)total, &d_out_id));
  PCGX_TRY(ar.alloc_n((size_t)total, &d_out_key));
  PCGX_TRY(ar.alloc(radix_sort_workspace_bytes(total), &ws));
  PCGX_TRY(staged_upload(d_q, q, (size_t)nq * 12, st));
  PCGX_TRY(staged_upload(d_off, offsets, (size_t)(nq + 1) * 8, st));
  const bool on_grid = !patched && range_on_grid(t);
  if (!on_grid) {
    // Poison the ids so that offsets inconsistent with the counts are caught below.
    PCGX_HIP_TRY(hipMemsetAsync(d_id, 0xFF, (size_t)total * 4, st));
    PCGX_HIP_TRY(hipMemsetAsync(d_key, 0, (size_t)total * 4, st));
    PCGX_HIP_TRY(hipMemsetAsync(d_query, 0, (size_t)total * 4, st));
  }
  const TreeView tv = t->view();
  const size_t lds = walk_stack_bytes(tv, kRangeBlock);
  int32_t *qperm = nullptr;
  if (nq >= kRangePresortMin) {
    PCGX_TRY(ar.alloc_n((size_t)nq, &qperm));
    PCGX_TRY(morton_order(d_q, nq, t->bbox_lo, t->bbox_hi, qperm, st));
  }
  int32_t *d_bad = nullptr;
  if (patched)
    PCGX_TRY(xtree_launch_range(outer, true, d_q, qperm, nq, max_range * max_range, nullptr, d_off, total, d_id, d_key,
                                d_query, st));
  else if (on_grid) {
    uint4 *d_rec = nullptr;
    PCGX_TRY(ar.alloc_n((size_t)total, &d_rec));
    PCGX_HIP_TRY(hipMemsetAsync(d_rec, 0xFF, (size_t)total * 16, st));  // (poisoned ids as above)
    PCGX_TRY(ar.alloc_n(1, &d_bad));
    PCGX_HIP_TRY(hipMemsetAsync(d_bad, 0, 4, st));
    hipLaunchKernelGGL(range_grid_kernel<true>, dim3(xcd_grid((unsigned)((nq + kRangeBlock - 1) / kRangeBlock))), dim3(kRangeBlock),
                       0, st, t->grid, (const float *)d_q, (const int32_t *)qperm, nq, max_range * max_range, nullptr, d_off, total,
                       d_rec);
    hipLaunchKernelGGL(range_split_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const uint4 *)d_rec, total,
                       d_id, d_key, d_query, d_bad);
  } else
    hipLaunchKernelGGL(range_kernel<true>, dim3(xcd_grid((unsigned)((nq + kRangeBlock - 1) / kRangeBlock))), dim3(kRangeBlock),
                       lds, st, tv, d_q, (const int32_t *)qperm, nq, max_range * max_range, nullptr, d_off, total, d_id,
                       d_key, d_query);
  const unsigned tb = (unsigned)((total + 255) / 256);
  hipLaunchKernelGGL(range_iota_kernel, dim3(tb), dim3(256), 0, st, vals[0], total);
  PCGX_HIP_TRY(hipMemcpyAsync(keys[0], d_key, (size_t)total * 4, hipMemcpyDeviceToDevice, st));
  // stable sort by DistSq, then stable by query: (query, DistSq) order, ties in discovery order
  int r1 = 0;
  PCGX_TRY(radix_sort_pairs(keys, vals, total, 32, ws, &r1, st));
  uint32_t *k2[2] = {keys[r1 ^ 1], keys[r1]};
  uint32_t *v2[2] = {vals[r1], vals[r1 ^ 1]};
  hipLaunchKernelGGL(range_gather_u32_kernel, dim3(tb), dim3(256), 0, st, d_query, v2[0], total, k2[0]);
  int qbits = 0;
  while (qbits < 32 && ((int64_t)1 << qbits) < nq) qbits++;
  int r2 = 0;
  PCGX_TRY(radix_sort_pairs(k2, v2, total, qbits, ws, &r2, st));
  const uint32_t *perm = v2[r2];  // final slot -> position in discovery order
  hipLaunchKernelGGL(range_gather_u32_kernel, dim3(tb), dim3(256), 0, st, (const uint32_t *)d_id, perm, total,
                     d_out_id);
  hipLaunchKernelGGL(range_gather_u32_kernel, dim3(tb), dim3(256), 0, st, d_key, perm, total, d_out_key);
  PCGX_HIP_TRY(hipGetLastError());
  if (on_grid) {  // equal DistSq of one query: into the walk's order (k2[r2]: the query of every sorted slot)
    const uint32_t *d_inv = nullptr;
    PCGX_TRY(range_inverse_map(t, &d_inv, st));
    unsigned long long *d_place = nullptr;
    PCGX_TRY(ar.alloc_n((size_t)total, &d_place));
    int32_t *d_long = nullptr;
    PCGX_TRY(ar.alloc_n(1, &d_long));
    PCGX_HIP_TRY(hipMemsetAsync(d_long, 0, 4, st));
    hipLaunchKernelGGL(range_tie_place_kernel, dim3(tb), dim3(256), 0, st, tv, d_inv, (const float *)d_q, (const uint32_t *)k2[r2],
                       (const uint32_t *)d_out_key, (const uint32_t *)d_out_id, total, d_place, d_long);
    // (into d_id: the discovery-order ids are done with)
    if (!d_bad) {
      PCGX_TRY(ar.alloc_n(1, &d_bad));
      PCGX_HIP_TRY(hipMemsetAsync(d_bad, 0, 4, st));
    }
    int32_t long_run = 0;
    PCGX_HIP_TRY(hipMemcpyAsync(&long_run, d_long, 4, hipMemcpyDeviceToHost, st));
    PCGX_HIP_TRY(hipStreamSynchronize(st));
    if (!long_run) {
      hipLaunchKernelGGL(range_tie_sort_kernel, dim3(tb), dim3(256), 0, st, (const uint32_t *)k2[r2], (const uint32_t *)d_out_key, total,
                         (const unsigned long long *)d_place, (const uint32_t *)d_out_id, (uint32_t *)d_id, d_bad);
    } else {
      // A run of more than kTieRunLimit exact ties (a cloud with one point taken a hundred thousand times): the per-slot
      // scan above is quadratic in the run.  Every slot instead by (query, DistSq, place) with stable LSD passes -- place
      // (low word, high word), DistSq, query: the slots keep their (query, DistSq) runs, inside a run they come out in
      // the walk's order, untied slots (place 0, a run each) where they were.  Four sorts: a rare input's price.
      const uint32_t *query_sorted = k2[r2];  // (lives in keys[]: copied out before the passes reuse them)
      uint32_t *d_qs = nullptr;
      PCGX_TRY(ar.alloc_n((size_t)total, &d_qs));
      PCGX_HIP_TRY(hipMemcpyAsync(d_qs, query_sorted, (size_t)total * 4, hipMemcpyDeviceToDevice, st));
      int place_bits = 0;  // place < 3^depth
      {
        unsigned long long lim = 1ull;
        for (int j = 0; j < tv.depth; j++) lim *= 3ull;
        while (place_bits < 64 && (lim >> place_bits) != 0ull) place_bits++;
      }
      uint32_t *kk[2] = {keys[0], keys[1]}, *vv[2] = {vals[0], vals[1]};
      hipLaunchKernelGGL(range_iota_kernel, dim3(tb), dim3(256), 0, st, vv[0], total);
      hipLaunchKernelGGL(range_place_bits_kernel, dim3(tb), dim3(256), 0, st, (const unsigned long long *)d_place, (const uint32_t *)nullptr,
                         total, 0, kk[0]);
      int r = 0;
      PCGX_TRY(radix_sort_pairs(kk, vv, total, place_bits < 32 ? place_bits : 32, ws, &r, st));
      auto next_pass = [&](int bits, auto fill) -> pcgx_status {  // keys of the next pass: by the order so far
        uint32_t *k_in = kk[r ^ 1], *v_in = vv[r];
        fill(k_in, (const uint32_t *)v_in);
        uint32_t *k3[2] = {k_in, kk[r]}, *v3[2] = {v_in, vv[r ^ 1]};
        int rr = 0;
        PCGX_TRY(radix_sort_pairs(k3, v3, total, bits, ws, &rr, st));
        kk[0] = k3[rr]; kk[1] = k3[rr ^ 1];
        vv[0] = v3[rr]; vv[1] = v3[rr ^ 1];
        r = 0;
        return PCGX_OK;
      };
      if (place_bits > 32)
        PCGX_TRY(next_pass(place_bits - 32, [&](uint32_t *k_in, const uint32_t *v_in) {
          hipLaunchKernelGGL(range_place_bits_kernel, dim3(tb), dim3(256), 0, st, (const unsigned long long *)d_place, v_in, total, 32, k_in);
        }));
      PCGX_TRY(next_pass(32, [&](uint32_t *k_in, const uint32_t *v_in) {
        hipLaunchKernelGGL(range_gather_u32_kernel, dim3(tb), dim3(256), 0, st, (const uint32_t *)d_out_key, v_in, total, k_in);
      }));
      PCGX_TRY(next_pass(qbits > 0 ? qbits : 1, [&](uint32_t *k_in, const uint32_t *v_in) {
        hipLaunchKernelGGL(range_gather_u32_kernel, dim3(tb), dim3(256), 0, st, (const uint32_t *)d_qs, v_in, total, k_in);
      }));
      hipLaunchKernelGGL(range_gather_u32_kernel, dim3(tb), dim3(256), 0, st, (const uint32_t *)d_out_id, (const uint32_t *)vv[0], total,
                         (uint32_t *)d_id);
    }
    PCGX_HIP_TRY(hipGetLastError());
    d_out_id = (uint32_t *)d_id;
  }
  // Go's int is 64 bits wide: widened (and checked: a poisoned id means the caller's offsets do not match the
  // counts) on the device, straight into the caller's slice
  int64_t *d_ids64 = nullptr;
  PCGX_TRY(ar.alloc_n((size_t)total, &d_ids64));
  if (!d_bad) {
    PCGX_TRY(ar.alloc_n(1, &d_bad));
    PCGX_HIP_TRY(hipMemsetAsync(d_bad, 0, 4, st));
  }
  hipLaunchKernelGGL(range_widen_check_kernel, dim3(tb), dim3(256), 0, st, (const uint32_t *)d_out_id, total, d_ids64, d_bad);
  PCGX_HIP_TRY(hipGetLastError());
  int32_t bad = 0;
  PCGX_HIP_TRY(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, st));
  PCGX_TRY(staged_download(ids, d_ids64, (size_t)total * 8, st));
  PCGX_TRY(staged_download(dist_sq, d_out_key, (size_t)total * 4, st));
  PCGX_HIP_TRY(hipStreamSynchronize(st));
  if (bad) return fail(PCGX_E_INVALID, "pcgx_kdtree_range_fill: offsets do not match the neighbour counts");
  return PCGX_OK;
}
