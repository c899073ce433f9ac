// knn_xwalk.h -- walk of the explicit (patched) tree of a handle that has seen DeletePoint; see
// knn_explicit.hip.  Reference: pc/storage/kdtree/kdtree.go:83-146 (nearestImpl), :148-197
// (rangeImpl), :199-222 (searchLeafNode).
#pragma once
#include "knn_walk.h"

namespace pcgx {

struct XTreeView {
  const float4 *pts;   // [n] {x, y, z, bits(id)} of node k
  const int4 *links;   // [n] {child0, child1, dim, -}; -1 = nil
  int32_t root;        // -1: empty tree
  int32_t depth;       // frames a walk may need
};

constexpr int kXBlock = 256;

// frame word: node index (27 bits) | side taken << 27
// on_leaf / on_pivot(node {x, y, z, bits(id)}, DistSq): called for every leaf / pivot the reference
// evaluates, in its order; return false to stop (MinDistSq cut).  bound(): current pruning bound (best.d for Nearest, maxRange^2 for Range).
template <class Bound, class Leaf, class Pivot>
__device__ __forceinline__ void xwalk(const XTreeView &xv, uint32_t *__restrict__ stk, const int stk_stride,
                                      const float qx, const float qy, const float qz, int64_t guard, Bound &&bound,
                                      Leaf &&on_leaf, Pivot &&on_pivot) {
  if (xv.root < 0) return;
  int32_t cur = xv.root, sp = 0;
  bool desc = true;
  for (; guard > 0; --guard) {
    if (desc) {
      // searchLeafNode step (kdtree.go:202-221)
      const int4 lk = xv.links[cur];
      const float4 nd = xv.pts[cur];
      if (lk.x < 0 && lk.y < 0) {  // no children: the leaf of this descent
        const float dx = nd.x - qx, dy = nd.y - qy, dz = nd.z - qz;
        if (!on_leaf(nd, (dx * dx + dy * dy) + dz * dz)) return;
        desc = false;
        continue;
      }
      int side;
      if (lk.x < 0) side = 1;        // only child1
      else if (lk.y < 0) side = 0;   // only child0
      else side = sel3(lk.z, nd.x, nd.y, nd.z) > sel3(lk.z, qx, qy, qz) ? 0 : 1;  // pivotVal > val -> child0
      stk[(sp++) * stk_stride] = (uint32_t)cur | ((uint32_t)side << 27);
      cur = side ? lk.y : lk.x;
    } else {
      if (sp == 0) return;
      const uint32_t fw = stk[(--sp) * stk_stride];
      const int32_t n = (int32_t)(fw & 0x07FFFFFFu);
      const int side = (int)(fw >> 27);
      const int4 lk = xv.links[n];
      const float4 nd = xv.pts[n];
      const float fp = sel3(lk.z, qx, qy, qz) - sel3(lk.z, nd.x, nd.y, nd.z);  // p[dim] - pivot[dim]
      if (fp * fp > bound()) continue;  // kdtree.go:111-115 / :173-177
      const float dx = nd.x - qx, dy = nd.y - qy, dz = nd.z - qz;
      if (!on_pivot(nd, (dx * dx + dy * dy) + dz * dz)) return;
      const int32_t other = side ? lk.x : lk.y;  // the child that is not on the stack (:124-132)
      if (other >= 0) {
        cur = other;
        desc = true;
      }
    }
  }
}

// Device copy of t's patched tree, uploaded on st if deletions happened since the last call.
pcgx_status xtree_view(const pcgx_kdtree *t, XTreeView *xv, hipStream_t st);

}  // namespace pcgx
