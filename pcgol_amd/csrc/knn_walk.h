// knn_walk.h -- device-side restatement of KDTree.Nearest, one query per lane,
// organised as a wave-persistent state machine.
//
// Reference: pc/storage/kdtree/kdtree.go:83-146 (Nearest / nearestImpl) and
// :199-222 (searchLeafNode).  The reference recursion
//     descend to a leaf; evaluate it; walk the stack upwards: plane test,
//     pivot test, recurse into the other child with the current best as bound
// is the in-order walk  visit(n) = visit(near(n)); test n; visit(far(n))
// with ONE running best, where
//   * a leaf replaces the best when NOT (d > best)      (kdtree.go:100-103,138-139)
//   * a pivot replaces the best only when d < best       (kdtree.go:117)
//   * the pivot and the far side are skipped when fp*fp > best  (kdtree.go:113)
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
// A frame whose plane test already fails when it would be pushed can never
// pass later (the best only shrinks), so it is not pushed at all.
//
// Execution shape (DESIGN.md "walk kernel"): a wave owns a contiguous range of
// (Morton-ordered) queries.  Every loop iteration performs ONE step for every
// lane -- at most one 16-byte node fetch, shared by both modes (descending /
// unwinding) -- and lanes whose query has finished pull the next query of the
// range at once, so lanes stay busy until the range is exhausted.  Which lane
// gets which query is a deterministic function of the input (no atomics).
#pragma once
#include "pcgx_internal.h"

namespace pcgx {

__device__ __forceinline__ float sel3(int dim, float a, float b, float c) {
  return dim == 0 ? a : (dim == 1 ? b : c);
}

// fetch(idx, qx, qy, qz): loads query idx.   emit(idx, qx, qy, qz, best_pos, best_d):
// consumes the result; best_pos is the in-order node index of the match or -1.
template <bool kMinDist, class Fetch, class Emit>
__device__ __forceinline__ void walk_range(const TreeView tv, uint2 *__restrict__ stk,
                                           const int stk_stride, const int64_t q_begin,
                                           const int64_t q_end, const float max_range_sq,
                                           const float min_dist_sq, Fetch &&fetch, Emit &&emit) {
  const int lane = (int)(threadIdx.x & 63u);
  const uint64_t lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  const uint32_t np1 = (uint32_t)tv.n + 1u;
  int64_t next = q_begin;  // wave-uniform
  bool active = false;
  int64_t my_q = 0;
  float qx = 0.0f, qy = 0.0f, qz = 0.0f, best_d = 0.0f;
  int32_t best_pos = -1;
  int32_t lo = 0, n = 0, depth = 0, sp = 0;
  bool desc = false;

  for (;;) {
    // ---- refill idle lanes from the wave's range --------------------------------
    const uint64_t idle = __ballot(!active);
    if (idle != 0ull) {
      if (next < q_end) {
        if (!active) {
          const int64_t idx = next + (int64_t)__popcll(idle & lt_mask);
          if (idx < q_end) {
            fetch(idx, qx, qy, qz);
            my_q = idx;
            best_d = max_range_sq;
            best_pos = -1;
            lo = 0;
            n = tv.n;
            depth = 0;
            sp = 0;
            desc = true;
            active = true;
          }
        }
        next += (int64_t)__popcll(idle);
      }
      if (__ballot(active) == 0ull) break;  // range exhausted and every lane done
    }

    // ---- choose the node this lane looks at in this step -------------------------
    int32_t mid = 0;
    bool look = false;
    float ffp = 0.0f;
    int32_t fdepth = 0, fn = 0;
    bool finish = false;
    if (active) {
      if (desc) {
        mid = lo + (n >> 1);
        look = true;
      } else if (sp == 0) {
        finish = true;
      } else {
        // unwind: examine the two topmost frames at once (kdtree.go:107-115)
        const uint2 f0 = stk[(sp - 1) * stk_stride];
        const uint2 f1 = stk[(sp >= 2 ? sp - 2 : 0) * stk_stride];
        const float fp0 = __uint_as_float(f0.y), fp1 = __uint_as_float(f1.y);
        const bool pass0 = !(fp0 * fp0 > best_d);
        const bool pass1 = sp >= 2 && !(fp1 * fp1 > best_d);
        uint32_t fx = 0;
        if (pass0) {
          fx = f0.x; ffp = fp0; sp -= 1; look = true;
        } else if (pass1) {
          fx = f1.x; ffp = fp1; sp -= 2; look = true;
        } else {
          sp = sp >= 2 ? sp - 2 : 0;
          finish = sp == 0;
        }
        mid = (int32_t)(fx & 0x03FFFFFFu);
        fdepth = (int32_t)((fx >> 26) & 31u);
        fn = (int32_t)((np1 >> fdepth) - 1u + (fx >> 31));
      }
    }

    if (look) {
      const float4 nd = tv.nodes[mid];
      const float dx = nd.x - qx, dy = nd.y - qy, dz = nd.z - qz;  // pivot.Sub(p)
      const float d = (dx * dx + dy * dy) + dz * dz;                // NormSq, mat/vec3.go:18-20
      if (desc) {
        if (n == 1) {  // leaf: kdtree.go:95-106
          if (!(d > best_d)) {
            best_d = d;
            best_pos = mid;
          }
          desc = false;
          if (kMinDist && best_d < min_dist_sq) finish = true;
        } else {  // searchLeafNode step: kdtree.go:202-221
          const int dim = depth % 3;
          const float pv = sel3(dim, nd.x, nd.y, nd.z);
          const float qv = sel3(dim, qx, qy, qz);
          const float fp = qv - pv;
          if (!(fp * fp > best_d)) {
            const uint32_t size_bit = (uint32_t)n - ((np1 >> depth) - 1u);
            stk[sp * stk_stride] = make_uint2((uint32_t)mid | ((uint32_t)depth << 26) | (size_bit << 31),
                                              __float_as_uint(fp));
            ++sp;
          }
          const int32_t half = n >> 1;
          if (n == 2 || pv > qv) {  // only child, or pivotVal > val -> child0
            n = half;
          } else {
            lo = mid + 1;
            n = n - half - 1;
          }
          ++depth;
        }
      } else {  // a frame that passed the plane test: kdtree.go:116-143
        if (d < best_d) {
          best_d = d;
          best_pos = mid;
          if (kMinDist && best_d < min_dist_sq) finish = true;
        }
        if (fn != 2) {  // fn == 2: single child, nextNode == nil (kdtree.go:130-132)
          const int32_t half = fn >> 1;
          if (ffp < 0.0f) {  // went to child0, the other side is child1
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
    }

    if (finish) {
      emit(my_q, qx, qy, qz, best_pos, best_d);
      active = false;
    }
  }
}

}  // namespace pcgx
