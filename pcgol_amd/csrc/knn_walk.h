// knn_walk.h -- device-side restatement of KDTree.Nearest for one query per lane.
//
// Reference: pc/storage/kdtree/kdtree.go:83-146 (Nearest / nearestImpl) and
// :199-222 (searchLeafNode).  The reference recursion
//     descend to a leaf; evaluate it; walk the stack upwards: plane test,
//     pivot test, recurse into the other child with the current best as bound
// is the in-order walk  visit(n) = visit(near(n)); test n; visit(far(n))
// with ONE running best, where
//   * a leaf replaces the best when NOT (d > best)      (kdtree.go:100-103,138-139)
//   * a pivot replaces the best only when d < best       (kdtree.go:117)
//   * the far side and the pivot are skipped when fp*fp > best  (kdtree.go:113)
//   * everything stops once best < MinDistSq             (kdtree.go:104,120,140)
// Visit order, comparisons and float32 expression order are those of the
// reference, so ids agree even on exact distance ties and for MinDistSq > 0.
//
// Tree encoding: see pcgx_internal.h (implicit in-order layout).  Subtree
// sizes at depth d are smin(d) or smin(d)+1 with smin(d) = ((N+1) >> d) - 1.
//
// Traversal stack: one 8-byte frame per pending ancestor, in LDS, laid out
// [level][thread] so that a wave's accesses are conflict-free:
//   .x = node index (26 bits) | depth (5 bits) << 26 | size bit << 31
//   .y = bits of fp = q[dim] - pivot[dim]   (sign = which side was taken,
//        fp*fp = the plane test; both bit-identical to recomputing them)
#pragma once
#include "pcgx_internal.h"

namespace pcgx {

__device__ __forceinline__ float sel3(int dim, float a, float b, float c) {
  return dim == 0 ? a : (dim == 1 ? b : c);
}

struct WalkResult {
  int32_t id;
  float dist_sq;
  float bx, by, bz;  // coordinates of the matched base point (valid if id >= 0)
};

template <bool kMinDist>
__device__ __forceinline__ WalkResult nearest_walk(const TreeView tv, uint2 *stk, int stk_stride,
                                                   float qx, float qy, float qz,
                                                   float max_range_sq, float min_dist_sq) {
  WalkResult best;
  best.id = -1;
  best.dist_sq = max_range_sq;
  best.bx = best.by = best.bz = 0.0f;
  const uint32_t np1 = (uint32_t)tv.n + 1u;
  int32_t lo = 0, n = tv.n, depth = 0, sp = 0;
  int32_t fdepth = 0, fn = 0;
  float ffp = 0.0f;
  bool desc = true;
  // Every iteration fetches exactly one node; a walk touches each node at
  // most twice, so 2n+2 bounds the loop whatever the data.
  for (int64_t guard = 2 * (int64_t)tv.n + 2; guard > 0; --guard) {
    int32_t mid;
    if (desc) {
      mid = lo + (n >> 1);
    } else {
      bool found = false;
      while (sp > 0) {
        --sp;
        uint2 f = stk[sp * stk_stride];
        ffp = __uint_as_float(f.y);
        if (ffp * ffp > best.dist_sq) continue;  // kdtree.go:111-115
        mid = (int32_t)(f.x & 0x03FFFFFFu);
        fdepth = (int32_t)((f.x >> 26) & 31u);
        fn = (int32_t)((np1 >> fdepth) - 1u + (f.x >> 31));
        found = true;
        break;
      }
      if (!found) break;
    }
    const float4 nd = tv.nodes[mid];
    const float dx = nd.x - qx, dy = nd.y - qy, dz = nd.z - qz;  // pivot.Sub(p)
    const float d = (dx * dx + dy * dy) + dz * dz;                // NormSq, mat/vec3.go:18-20
    if (desc) {
      if (n == 1) {  // leaf: kdtree.go:95-106
        if (!(d > best.dist_sq)) {
          best.dist_sq = d;
          best.id = __float_as_int(nd.w);
          best.bx = nd.x; best.by = nd.y; best.bz = nd.z;
        }
        if (kMinDist && best.dist_sq < min_dist_sq) break;
        desc = false;
      } else {  // searchLeafNode step: kdtree.go:202-221
        const int dim = depth % 3;
        const float pv = sel3(dim, nd.x, nd.y, nd.z);
        const float qv = sel3(dim, qx, qy, qz);
        const float fp = qv - pv;
        const uint32_t size_bit = (uint32_t)n - ((np1 >> depth) - 1u);
        stk[sp * stk_stride] =
            make_uint2((uint32_t)mid | ((uint32_t)depth << 26) | (size_bit << 31), __float_as_uint(fp));
        ++sp;
        const int32_t half = n >> 1;
        if (n == 2 || pv > qv) {  // only child, or pivotVal > val -> child0
          n = half;
        } else {
          lo = mid + 1;
          n = n - half - 1;
        }
        ++depth;
      }
    } else {  // unwinding through a frame that passed the plane test: kdtree.go:116-143
      if (d < best.dist_sq) {
        best.dist_sq = d;
        best.id = __float_as_int(nd.w);
        best.bx = nd.x; best.by = nd.y; best.bz = nd.z;
        if (kMinDist && best.dist_sq < min_dist_sq) break;
      }
      if (fn == 2) continue;  // single child: nextNode == nil (kdtree.go:130-132)
      const int32_t half = fn >> 1;
      if (ffp < 0.0f) {  // went to child0, other side is child1
        lo = mid + 1;
        n = fn - half - 1;
      } else {
        lo = mid - half;
        n = half;
      }
      depth = fdepth + 1;
      desc = true;
    }
  }
  return best;
}

}  // namespace pcgx
