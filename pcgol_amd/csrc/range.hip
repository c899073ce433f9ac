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
// Two entry points because the result length is data dependent: pcgx_kdtree_range_count
// (walk, count) and pcgx_kdtree_range_fill (walk again, write at the caller's offsets, then
// one stable sort of the whole batch by (query, DistSq) with the radix sort of sort.hip).
// One query per lane; 4-byte frames [level][thread] in LDS as in knn_walk.h.
#include <string.h>

#include <vector>

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

extern "C" pcgx_status pcgx_kdtree_range_count(const pcgx_kdtree *t, const float *q, int64_t nq, float max_range,
                                               int64_t *counts) {
  PCGX_API_CALL();
  if (!t || nq < 0 || (nq > 0 && (!q || !counts))) return fail(PCGX_E_INVALID, "pcgx_kdtree_range_count: bad argument");
  if (nq == 0) return PCGX_OK;
  PCGX_TRY(ensure_init());
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
  // Poison the ids so that offsets inconsistent with the counts are caught below.
  PCGX_HIP_TRY(hipMemsetAsync(d_id, 0xFF, (size_t)total * 4, st));
  PCGX_HIP_TRY(hipMemsetAsync(d_key, 0, (size_t)total * 4, st));
  PCGX_HIP_TRY(hipMemsetAsync(d_query, 0, (size_t)total * 4, st));
  const TreeView tv = t->view();
  const size_t lds = walk_stack_bytes(tv, kRangeBlock);
  int32_t *qperm = nullptr;
  if (nq >= kRangePresortMin) {
    PCGX_TRY(ar.alloc_n((size_t)nq, &qperm));
    PCGX_TRY(morton_order(d_q, nq, t->bbox_lo, t->bbox_hi, qperm, st));
  }
  if (patched)
    PCGX_TRY(xtree_launch_range(outer, true, d_q, qperm, nq, max_range * max_range, nullptr, d_off, total, d_id, d_key,
                                d_query, st));
  else
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
  // Go's int is 64 bits wide: widened (and checked: a poisoned id means the caller's offsets do not match the
  // counts) on the device, straight into the caller's slice
  int64_t *d_ids64 = nullptr;
  int32_t *d_bad = nullptr;
  PCGX_TRY(ar.alloc_n((size_t)total, &d_ids64));
  PCGX_TRY(ar.alloc_n(1, &d_bad));
  PCGX_HIP_TRY(hipMemsetAsync(d_bad, 0, 4, st));
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
