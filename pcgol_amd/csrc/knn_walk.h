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
// Tree encoding: see pcgx_internal.h (implicit tree in BFS order: node b has
// children 2b / 2b+1, depth = floor(log2 b)).  Subtree sizes at depth d are
// smin(d) or smin(d)+1 with smin(d) = ((N+1) >> d) - 1.
//
// Traversal stack: one 8-byte frame per pending ancestor, in LDS, laid out
// [level][thread] so that a wave's accesses are conflict-free:
//   .x = BFS node index (27 bits) | size bit << 31
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
//
// The step is written with selects rather than nested branches: on gfx950 a
// divergent if/else costs scalar exec-mask bookkeeping on the CU's single
// scalar unit, which (not memory) was what bound the first version of this
// kernel (profiles/r01a_pmc.json: 121 SALU and 85 VALU instructions per node
// fetch, 29 % of the lanes active per VALU instruction).
constexpr int kRefillThreshold = 12;  // refill when this many lanes are idle (or nothing is active)

template <bool kMinDist, class Fetch, class Emit>
__device__ __forceinline__ void walk_range(const TreeView tv, uint2 *__restrict__ stk,
                                           const int stk_stride, const int64_t q_begin,
                                           const int64_t q_end, const float max_range_sq,
                                           const float min_dist_sq, Fetch &&fetch, Emit &&emit) {
  const int lane = (int)(threadIdx.x & 63u);
  const uint64_t lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  const uint32_t np1 = (uint32_t)tv.n + 1u;
  const int32_t count = (int32_t)(q_end - q_begin);  // a wave's range is far below 2^31
  int32_t next = 0;                                   // wave-uniform, relative to q_begin
  bool active = false;
  int32_t my_q = 0;
  float qx = 0.0f, qy = 0.0f, qz = 0.0f, best_d = 0.0f;
  int32_t best_pos = -1;
  uint32_t b = 1;  // BFS index of the node the lane is at
  int32_t n = 0, sp = 0;
  bool desc = false;
  bool pending = false;  // finished, result not yet emitted

  for (;;) {
    // ---- emit finished lanes and refill them from the wave's range ---------------
    // Both are divergent sections, so they run only once enough lanes are waiting
    // (or nothing is left to step): a lane that finished just idles until then.
    const uint64_t idle = __ballot(!active);
    const int n_idle = __popcll(idle);
    if (n_idle >= kRefillThreshold || n_idle == 64) {
      if (pending) {
        emit(q_begin + my_q, qx, qy, qz, best_pos, best_d);
        pending = false;
      }
      if (next < count) {
        if (!active) {
          const int32_t idx = next + (int32_t)__popcll(idle & lt_mask);
          if (idx < count) {
            fetch(q_begin + idx, qx, qy, qz);
            my_q = idx;
            best_d = max_range_sq;
            best_pos = -1;
            b = 1u;
            n = tv.n;
            sp = 0;
            desc = true;
            active = true;
          }
        }
        next += n_idle;
      } else if (n_idle == 64) {
        break;  // range exhausted, every lane done and emitted
      }
    }

    // ---- unwinding lanes: look at the two topmost frames (kdtree.go:107-115) ------
    const bool popping = active && !desc;
    const int32_t i0 = sp >= 1 ? sp - 1 : 0, i1 = sp >= 2 ? sp - 2 : 0;
    const uint2 f0 = stk[i0 * stk_stride];
    const uint2 f1 = stk[i1 * stk_stride];
    const float fp0 = __uint_as_float(f0.y), fp1 = __uint_as_float(f1.y);
    const bool pass0 = sp >= 1 && !(fp0 * fp0 > best_d);
    const bool pass1 = sp >= 2 && !(fp1 * fp1 > best_d);
    const bool pfound = popping && (pass0 || pass1);
    const uint32_t fx = pass0 ? f0.x : f1.x;
    const float ffp = pass0 ? fp0 : fp1;
    const int32_t sp_pop = pass0 ? sp - 1 : i1;  // pass1 -> sp-2; neither -> max(sp-2, 0)
    sp = popping ? sp_pop : sp;
    bool finish = popping && !pfound && sp == 0;
    const uint32_t fb = fx & 0x07FFFFFFu;
    const int32_t fdepth = 31 - __clz((int)(fb | 1u));
    const int32_t fn = (int32_t)((np1 >> fdepth) - 1u + (fx >> 31));

    // ---- the one node fetch of this step -----------------------------------------
    const uint32_t at = desc ? b : fb;
    const bool look = active && (desc || pfound);
    if (look) {
      const float4 nd = tv.nodes[at];
      const float dx = nd.x - qx, dy = nd.y - qy, dz = nd.z - qz;  // pivot.Sub(p)
      const float d = (dx * dx + dy * dy) + dz * dz;                // NormSq, mat/vec3.go:18-20
      const bool leaf = desc && n == 1;
      const bool inner = desc && n != 1;
      // leaf: replace unless d > best (kdtree.go:95-103); pivot: replace if d < best (:116-119)
      const bool take = leaf ? !(d > best_d) : (!desc && d < best_d);
      best_d = take ? d : best_d;
      best_pos = take ? (int32_t)at : best_pos;
      if (kMinDist) finish = finish || ((leaf || take) && best_d < min_dist_sq);  // :104,120,140

      // descending through an inner node: searchLeafNode step (kdtree.go:202-221)
      const int32_t depth = 31 - __clz((int)b);
      const int dim = depth % 3;
      const float pv = sel3(dim, nd.x, nd.y, nd.z);
      const float qv = sel3(dim, qx, qy, qz);
      const float fp = qv - pv;
      if (inner && !(fp * fp > best_d)) {
        const uint32_t size_bit = (uint32_t)n - ((np1 >> depth) - 1u);
        stk[sp * stk_stride] = make_uint2(b | (size_bit << 31), __float_as_uint(fp));
        ++sp;
      }
      const int32_t half = n >> 1;
      const bool go_left = n == 2 || pv > qv;  // only child, or pivotVal > val -> child0
      const uint32_t d_b = 2u * b + (go_left ? 0u : 1u);
      const int32_t d_n = go_left ? half : n - half - 1;

      // unwinding through a frame that passed the plane test: the other side (kdtree.go:124-137)
      const int32_t phalf = fn >> 1;
      const bool went_left = ffp < 0.0f;
      const uint32_t p_b = 2u * fb + (went_left ? 1u : 0u);
      const int32_t p_n = went_left ? fn - phalf - 1 : phalf;
      const bool p_far = fn != 2;  // fn == 2: single child, nextNode == nil (:130-132)

      b = desc ? d_b : p_b;
      n = desc ? d_n : p_n;
      desc = desc ? inner : p_far;
    }

    if (finish) {
      active = false;
      pending = true;
    }
  }
}

}  // namespace pcgx
