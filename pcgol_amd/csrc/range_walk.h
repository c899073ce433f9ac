// range_walk.h -- device-side restatement of KDTree.Range's walk (rangeImpl), one query per lane.
//
// Reference: pc/storage/kdtree/kdtree.go:148-197.  rangeImpl is the same in-order walk as
// nearestImpl with a FIXED bound:
//   leaf:   hit if dsq < maxRange^2                                      (:166-169)
//   unwind: skip the pivot and the far side if fp*fp > maxRange^2        (:173-177)
//           hit the pivot if dsq < maxRange^2                            (:178-181)
//           recurse into the other child                                 (:182-195)
// Hits are reported in the reference's discovery order.  Used by range.hip (KDTree.Range) and
// segment.hip (region growing: the same neighbourhoods feed a union-find).
#pragma once
#include "knn_walk.h"

namespace pcgx {

constexpr int kRangeWalkBlock = 64;  // threads per block of a kernel using range_walk

// stk: this lane's frame column in LDS ([level][stk_stride], walk_stack_bytes(tv, block)).
// on_hit(id, dist_sq) is called for every point with dist_sq < bound, in discovery order.
template <class Hit>
__device__ __forceinline__ void range_walk(const TreeView &tv, uint32_t *__restrict__ stk, const int stk_stride,
                                           const float qx, const float qy, const float qz, const float bound,
                                           Hit &&on_hit) {
  const uint32_t np1 = (uint32_t)tv.n + 1u;
  uint32_t b = 1;
  int32_t n = tv.n, sp = 0;
  bool desc = true;
  // every iteration fetches one node; a walk touches a node at most twice
  for (int64_t guard = 2 * (int64_t)tv.n + 2; guard > 0; --guard) {
    uint32_t at = b, fw = 0;
    if (!desc) {
      if (sp == 0) break;
      fw = stk[(--sp) * stk_stride];
      at = fw & 0x07FFFFFFu;
    }
    const float4 nd = node_at(tv.nodes, at);
    const float dx = nd.x - qx, dy = nd.y - qy, dz = nd.z - qz;
    const float d = (dx * dx + dy * dy) + dz * dz;
    const int32_t depth = 31 - __clz((int)at);
    const int dim = depth % 3;
    const float pv = sel3(dim, nd.x, nd.y, nd.z), qv = sel3(dim, qx, qy, qz);
    const float fp = qv - pv;
    const bool plane_ok = !(fp * fp > bound);
    bool hit;
    if (desc) {
      if (n == 1) {  // leaf
        hit = d < bound;
        desc = false;
      } else {  // searchLeafNode step (kdtree.go:202-221); a frame that cannot pass is not pushed
        hit = false;
        const int32_t half = n >> 1;
        const bool go_left = n == 2 || pv > qv;
        if (plane_ok) {
          const uint32_t size_bit = (uint32_t)n - ((np1 >> depth) - 1u);
          stk[(sp++) * stk_stride] = b | (go_left ? (1u << 27) : 0u) | (size_bit << 31);
        }
        b = 2u * b + (go_left ? 0u : 1u);
        n = go_left ? half : n - half - 1;
      }
    } else {  // a popped frame always passes its plane test (the bound is fixed)
      hit = d < bound;
      const int32_t fn = (int32_t)((np1 >> depth) - 1u + (fw >> 31));
      if (fn != 2) {  // the other child (kdtree.go:182-195)
        const bool went_left = ((fw >> 27) & 1u) != 0u;
        const int32_t half = fn >> 1;
        b = 2u * at + (went_left ? 1u : 0u);
        n = went_left ? fn - half - 1 : half;
        desc = true;
      }
    }
    if (hit) on_hit(__float_as_int(nd.w), d);
  }
}

}  // namespace pcgx
