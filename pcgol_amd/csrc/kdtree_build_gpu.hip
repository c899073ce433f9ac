// kdtree_build_gpu.hip -- device-side construction of the reference's KD-tree order.
//
// Reference: pc/storage/kdtree/kdtree.go:348-370 (newNode): at depth d the sub-slice of a
// node is sorted ascending by coordinate d%3 (Less = strict <, :407-409), the element at
// len/2 becomes the node and the halves recurse.  After the recursion the slice, read left
// to right, is the in-order traversal of the tree (pcgx_internal.h).  Ties on the split
// coordinate keep their current relative order (this library's definition, see
// kdtree_build.cpp; Go's sort is unstable there).
//
// Level-synchronous form of the same recursion: before level d the array is cut into
// "units" -- the sub-slices of the depth-d nodes, and single elements already fixed as the
// node of a shallower level.  Level d is ONE stable sort of the whole array by the pair
//     (rank of the unit in array order, coordinate d%3 of the element)
// which sorts every depth-d sub-slice by its coordinate, leaves every fixed element where
// it is, and keeps ties in their current order.  The pair is sorted as two stable LSD radix
// sorts (coordinate first, unit second) with the hand-written radix sort of sort.hip;
// floats are mapped to order-preserving integers with -0.0 folded onto +0.0 (they compare
// equal under <).  NaN coordinates have no consistent order under <; clouds containing
// NaNs are built on the host instead.
#include <stdlib.h>

#include "pcgx_internal.h"

namespace pcgx {

// Rank (in array order) of the unit that holds position p before level `depth` is sorted: walk
// the implicit tree (node of [lo, lo+n) sits at lo + n/2).  A subtree with r more levels to split
// holds 2^r sub-slices and 2^r - 1 fixed nodes, so the rank needs only depth + 1 bits -- the
// second radix sort of a level takes ceil((depth + 1) / 8) passes instead of ceil(log2 N / 8).
// (Sub-slices that are empty still count: ranks only have to be monotonic.)  If p is the node of
// a level shallower than `depth` the unit is that single position (fixed = true).
__device__ __forceinline__ uint32_t unit_rank(uint32_t p, uint32_t n_total, int depth, bool &fixed) {
  uint32_t lo = 0, n = n_total, rank = 0;
  fixed = false;
  for (int d = 0; d < depth; d++) {
    const uint32_t half = n >> 1, mid = lo + half;
    const uint32_t left_units = (2u << (depth - d - 1)) - 1u;  // units of a child subtree
    if (p == mid && n > 0u) {
      fixed = true;
      return rank + left_units;
    }
    if (p < mid) {
      n = half;
    } else {
      rank += left_units + 1u;
      lo = mid + 1;
      n = n - half - 1;
    }
  }
  return rank;
}

__device__ __forceinline__ uint32_t ordered_bits(float v) {
  v = v == 0.0f ? 0.0f : v;  // -0.0 and +0.0 are equal under <
  const uint32_t u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// keys[p] = ordered coordinate of the element at position p (0 for fixed elements), vals[p] = p
__global__ __launch_bounds__(256) void kb_coord_key_kernel(const float *__restrict__ xyz,
                                                           const uint32_t *__restrict__ order, uint32_t n,
                                                           int depth, uint32_t *__restrict__ keys,
                                                           uint32_t *__restrict__ vals) {
  const uint32_t p = blockIdx.x * 256u + threadIdx.x;
  if (p >= n) return;
  bool fixed;
  (void)unit_rank(p, n, depth, fixed);
  keys[p] = fixed ? 0u : ordered_bits(xyz[3 * (size_t)order[p] + depth % 3]);
  vals[p] = p;
}

// keys[j] = unit rank of the ORIGINAL position vals[j] (the element sorted to slot j)
__global__ __launch_bounds__(256) void kb_unit_key_kernel(const uint32_t *__restrict__ vals, uint32_t n, int depth,
                                                          uint32_t *__restrict__ keys) {
  const uint32_t j = blockIdx.x * 256u + threadIdx.x;
  if (j >= n) return;
  bool fixed;
  keys[j] = unit_rank(vals[j], n, depth, fixed);
}

__global__ __launch_bounds__(256) void kb_permute_kernel(const uint32_t *__restrict__ order,
                                                         const uint32_t *__restrict__ vals, uint32_t n,
                                                         uint32_t *__restrict__ order_out) {
  const uint32_t j = blockIdx.x * 256u + threadIdx.x;
  if (j < n) order_out[j] = order[vals[j]];
}

__global__ __launch_bounds__(256) void kb_iota_kernel(uint32_t *__restrict__ a, uint32_t n) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < n) a[i] = i;
}

// nodes[b] = {x, y, z, bits(id)} of the point at the in-order position of BFS slot b
__global__ __launch_bounds__(256) void kb_fill_bfs_kernel(const float *__restrict__ xyz,
                                                          const uint32_t *__restrict__ order, uint32_t n_total,
                                                          uint32_t slots, const int32_t *__restrict__ labels,
                                                          float4 *__restrict__ nodes) {
  const uint32_t b = blockIdx.x * 256u + threadIdx.x;
  if (b >= slots) return;
  float4 out = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (b >= 1u) {
    const int depth = 31 - __clz((int)b);
    uint32_t lo = 0, n = n_total;
    bool exists = true;
    for (int d = depth - 1; d >= 0 && exists; d--) {  // follow b's bits from the root
      const uint32_t half = n >> 1;
      if ((b >> d) & 1u) {
        lo = lo + half + 1;
        n = n - half - 1;
      } else {
        n = half;
      }
      exists = n > 0u && n <= n_total;
    }
    if (exists && n > 0u) {
      const uint32_t id = order[lo + (n >> 1)];
      // labels: the id a node reports (trees rebuilt over the points left after DeletePoint keep
      // the ids of the original accessor)
      const uint32_t label = labels ? (uint32_t)labels[id] : id;
      out = make_float4(xyz[3 * (size_t)id], xyz[3 * (size_t)id + 1], xyz[3 * (size_t)id + 2], __uint_as_float(label));
    }
  }
  nodes[b] = out;
}

// The levels below d0, where every sub-slice fits LDS: one workgroup per sub-slice of level d0 carries out ALL the
// remaining levels there -- at every level each element counts the elements of its (ever smaller) sub-slice that
// sort before it (smaller coordinate, or equal and earlier: the stable sort the levels above do with radix passes)
// and moves to that place.  O(n x slice) compares, but in LDS and with no launch in between: eleven levels of a
// 1M-point build were 60 radix passes (1.5 ms), this is one launch.  In place: a workgroup reads its slice before it
// writes it, and the positions between the slices (the nodes of the levels above) are not touched.
constexpr int kUnitMax = 1024;
constexpr int kUnitThreads = 512;
__global__ __launch_bounds__(kUnitThreads) void kb_finish_units_kernel(const float *__restrict__ xyz, uint32_t *__restrict__ order,
                                                                       uint32_t n_total, int d0, int depth) {
  __shared__ __attribute__((aligned(16))) uint32_t s_key[2][3][kUnitMax];
  __shared__ uint32_t s_id[2][kUnitMax];
  uint32_t lo = 0, n = n_total;
  for (int d = d0 - 1; d >= 0; d--) {  // the blockIdx.x-th sub-slice of level d0, path bits from the root
    const uint32_t half = n >> 1;
    if ((blockIdx.x >> d) & 1u) {
      lo = lo + half + 1u;
      n = n > half ? n - half - 1u : 0u;
    } else {
      n = half;
    }
  }
  if (n <= 1u || n > (uint32_t)kUnitMax) return;  // (uniform; the host picks d0 so that every slice fits)
  for (uint32_t i = threadIdx.x; i < n; i += kUnitThreads) {
    const uint32_t id = order[lo + i];
    s_id[0][i] = id;
#pragma unroll
    for (int k = 0; k < 3; k++) s_key[0][k][i] = ordered_bits(xyz[3 * (size_t)id + k]);
  }
  __syncthreads();
  int cur = 0;
  for (int d = d0; d + 1 < depth; d++) {
    const int dim = d % 3;
    for (uint32_t i = threadIdx.x; i < n; i += kUnitThreads) {
      // the sub-slice of level d that holds position i (or i is the node of a level in between: it stays)
      uint32_t slo = 0, sn = n;
      bool fixed = false;
      for (int dd = d0; dd < d && !fixed && sn > 0u; dd++) {
        const uint32_t half = sn >> 1, mid = slo + half;
        if (i == mid) {
          fixed = true;
        } else if (i < mid) {
          sn = half;
        } else {
          slo = mid + 1u;
          sn = sn - half - 1u;
        }
      }
      uint32_t pos = i;
      if (!fixed && sn > 1u) {
        // (coordinate, position) pairs compared as one 64-bit number: smaller coordinate, or equal and earlier
        const uint32_t *kd = s_key[cur][dim];
        const unsigned long long mine = ((unsigned long long)kd[i] << 32) | i;
        auto before = [&](uint32_t kj, uint32_t j) { return ((((unsigned long long)kj << 32) | j) < mine) ? 1u : 0u; };
        uint32_t rank = 0, j = slo;
        const uint32_t end = slo + sn;
        for (; j < end && (j & 3u); j++) rank += before(kd[j], j);
        for (; j + 4u <= end; j += 4u) {  // four keys per LDS read
          const uint4 k4 = *reinterpret_cast<const uint4 *>(kd + j);
          rank += before(k4.x, j) + before(k4.y, j + 1u) + before(k4.z, j + 2u) + before(k4.w, j + 3u);
        }
        for (; j < end; j++) rank += before(kd[j], j);
        pos = slo + rank;
      }
      s_id[cur ^ 1][pos] = s_id[cur][i];
#pragma unroll
      for (int k = 0; k < 3; k++) s_key[cur ^ 1][k][pos] = s_key[cur][k][i];
    }
    __syncthreads();
    cur ^= 1;
  }
  for (uint32_t i = threadIdx.x; i < n; i += kUnitThreads) order[lo + i] = s_id[cur][i];
}

// d_xyz: packed xyz of the base cloud on the device; d_order (out): in-order point ids;
// d_nodes (out): BFS slots.  Uses the arena (caller has begun it).
pcgx_status build_tree_device(const float *d_xyz, int64_t n, int32_t depth, uint32_t *d_order, float4 *d_nodes,
                              const int32_t *d_labels, hipStream_t st) {
  Arena &ar = ctx().arena;
  const uint32_t un = (uint32_t)n;
  const unsigned nb = (unsigned)((n + 255) / 256);
  uint32_t *keys[2] = {nullptr, nullptr}, *vals[2] = {nullptr, nullptr}, *order_tmp = nullptr;
  void *ws = nullptr;
  PCGX_TRY(ar.alloc_n((size_t)n, &keys[0]));
  PCGX_TRY(ar.alloc_n((size_t)n, &keys[1]));
  PCGX_TRY(ar.alloc_n((size_t)n, &vals[0]));
  PCGX_TRY(ar.alloc_n((size_t)n, &vals[1]));
  PCGX_TRY(ar.alloc_n((size_t)n, &order_tmp));
  PCGX_TRY(ar.alloc(radix_sort_workspace_bytes(n), &ws));
  uint32_t *cur = d_order, *nxt = order_tmp;
  hipLaunchKernelGGL(kb_iota_kernel, dim3(nb), dim3(256), 0, st, cur, un);
  // levels 0 .. depth-2 have sub-slices longer than one element; from level d0 on every sub-slice fits LDS
  // (a child holds at most half of its parent's elements) and the rest is one launch.  PCGX_BUILD_LDS=0: radix
  // passes all the way down (tests compare the two).
  int d0 = 0;
  while (d0 + 1 < depth && (n >> d0) > kUnitMax) d0++;
  if (const char *e = getenv("PCGX_BUILD_LDS"))
    if (*e == '0') d0 = depth;
  for (int d = 0; d + 1 < depth && d < d0; d++) {
    hipLaunchKernelGGL(kb_coord_key_kernel, dim3(nb), dim3(256), 0, st, d_xyz, cur, un, d, keys[0], vals[0]);
    int r1 = 0;
    PCGX_TRY(radix_sort_pairs(keys, vals, n, 32, ws, &r1, st));
    // second (stable) sort by unit start; its input values are vals[r1]
    uint32_t *k2[2] = {keys[r1 ^ 1], keys[r1]};
    uint32_t *v2[2] = {vals[r1], vals[r1 ^ 1]};
    hipLaunchKernelGGL(kb_unit_key_kernel, dim3(nb), dim3(256), 0, st, v2[0], un, d, k2[0]);
    int r2 = 0;
    // unit ranks at level d: d + 1 bits (level 0 is one unit: nothing to regroup)
    if (d > 0) PCGX_TRY(radix_sort_pairs(k2, v2, n, d + 1, ws, &r2, st));
    hipLaunchKernelGGL(kb_permute_kernel, dim3(nb), dim3(256), 0, st, cur, v2[r2], un, nxt);
    uint32_t *t = cur;
    cur = nxt;
    nxt = t;
  }
  if (d0 + 1 < depth)
    hipLaunchKernelGGL(kb_finish_units_kernel, dim3(1u << d0), dim3(kUnitThreads), 0, st, d_xyz, cur, un, d0, (int)depth);
  if (cur != d_order)
    PCGX_HIP_TRY(hipMemcpyAsync(d_order, cur, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
  const uint32_t slots = 1u << depth;
  hipLaunchKernelGGL(kb_fill_bfs_kernel, dim3((slots + 255) / 256), dim3(256), 0, st, d_xyz, d_order, un, slots,
                     d_labels, d_nodes);
  PCGX_HIP_TRY(hipGetLastError());
  return PCGX_OK;
}

}  // namespace pcgx
