// knn_walk.h -- device-side restatement of KDTree.Nearest, one query per lane,
// organised as a wave-persistent state machine with a speculative, parallel
// first descent.
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
// Tree encoding: pcgx_internal.h (implicit tree in BFS order: node b has
// children 2b / 2b+1, depth = floor(log2 b)).  Subtree sizes at depth d are
// smin(d) or smin(d)+1 with smin(d) = ((N+1) >> d) - 1.
//
// ---- 1. speculative first descent (prepare_query) -----------------------------
// The first root-to-leaf descent is a chain of ~log2(N) DEPENDENT node fetches
// in the reference.  Here a grid directory (TreeView::dir, built with the tree)
// predicts the leaf a query will reach; in BFS order the ancestors of that leaf
// are plain shifts of its index, so the split values of the WHOLE predicted path
// (and the leaf itself) are fetched in parallel -- one memory round trip; the
// split values of the top 11 levels come from an LDS copy per workgroup.  The
// prediction is then VERIFIED level by level with the reference's own
// comparison (pivotVal > val -> child0, kdtree.go:216): up to the first level m
// where the real descent leaves the predicted path everything is exactly what
// the reference computes; from m on the lane continues with ordinary dependent
// steps.  A wrong prediction therefore costs time, never correctness.
//
// The verification is branch-free: one bit per level for "the real descent goes
// to child1" (pivotVal > val is false) is compared with the path bits of the
// predicted leaf; the first differing level is m.  Subtree sizes are never
// tracked: the size of BFS node b is a closed form of b (node_size below), needed
// only to recognise leaves (size 1) and single-child nodes (size 2).
//
// Frames of this first descent are not materialised: the path is one word
// (path_b, BFS index of the deepest verified node) and `pend` has bit j set when
// the ancestor at level j still has to be examined when the walk unwinds to it.
// When the walk pops such a level it re-fetches the ancestor, recomputes fp
// (bit-identical) and applies the plane test, as the reference does.
//
// Pruning bound.  A level is recorded in `pend` (a frame pushed, a popped frame
// followed) only if its plane test can still pass: fp*fp <= bound, where
//   * bound = the running best, as in the reference, and additionally
//   * with MinDistSq == 0 (kExact): bound = min(best, ub), ub = the distance to
//     some point OF THE TREE: the predicted leaf's point, and optionally a caller's
//     hint (the ICP loop passes the distance to the point matched in the previous
//     iteration).  ub >= d*, the final nearest distance; every
//     subtree holding a point at distance d* has fp*fp <= d* <= bound for the
//     planes of all its ancestors, so it is visited when the reference visits
//     it and in the same relative order, and candidates farther than d* never
//     decide the outcome (a leaf at d* replaces, a pivot at d* replaces only a
//     strictly larger best -- the same in both walks).  The returned {id, dist}
//     hence equals the reference's, exact ties included.
//   * with MinDistSq > 0 (approximate, visit-order dependent search) only the
//     running best is used; the first-descent levels are filtered with the best
//     right after the verified leaf, which is when the reference tests them.
//
// ---- 2. wave-persistent stepping (walk_queries) -------------------------------
// A workgroup owns a contiguous range of 64-query chunks; its waves pull chunks
// from an LDS counter until the range is exhausted (a wave that gets easy queries
// simply takes more chunks, so the waves of a workgroup finish together).  Ranges
// are handed to workgroups so that the workgroups of one XCD (blockIdx % 8, the
// observed round-robin placement) cover one contiguous eighth of the batch: with
// spatially ordered queries a CU then touches a small part of the tree (L1) and
// an XCD one eighth of it (fits its private 4 MB L2).  Handing chunks to
// arbitrary CUs through a global counter was measured 1.3-2x slower.  Placement
// affects speed only: results are written per query.
// A chunk is prepared by all 64 lanes together (full lane
// efficiency) into the wave's LDS queue; every loop
// iteration then performs ONE step for every lane -- at most one 16-byte node
// fetch, shared by all modes -- and lanes whose query has finished take the
// next prepared query from the queue.  Explicit frames exist only below the
// verified path: one 32-bit word per pending ancestor, in LDS laid out
// [level][thread] (conflict-free): the BFS index c of the CHILD the descent
// took (the ancestor is c >> 1, the other side c ^ 1) -- the same word a level
// of the first descent yields as a shift of path_b.
// Popping a frame re-fetches its node and recomputes fp = q[dim] - pivot[dim]
// (bit-identical to the value at push time) for the plane test, exactly like a
// level of the first descent; a frame is pushed only if that test can still pass.
// Results are written per query, so they do not depend on which wave or lane
// served a query.
#pragma once
#include "pcgx_internal.h"

namespace pcgx {

__device__ __forceinline__ float sel3(int dim, float a, float b, float c) {
  return dim == 0 ? a : (dim == 1 ? b : c);
}

constexpr int kMaxLevels = 26;       // inner levels on a root-to-leaf path (N <= 2^26)

// Size of the subtree of BFS node b at depth `depth` in a tree of m1 - 1 points.  With
// m = size + 1 the reference's split (kdtree.go:357-369: len/2 left of the median, the rest
// right) reads m_child0 = ceil(m / 2), m_child1 = floor(m / 2); after `depth` halvings the
// 2^depth parts differ by at most one and the larger ones are those whose bit-reversed path is
// below the remainder.  Nodes that do not exist give 0 or 0xFFFFFFFF.
__device__ __forceinline__ uint32_t node_size(uint32_t b, int depth, uint32_t m1) {
  const uint32_t rev = __brev(b) >> ((32 - depth) & 31);  // path bits, the root's choice lowest
  const uint32_t rem = m1 & ((1u << depth) - 1u);
  return (m1 >> depth) - 1u + (rev < rem ? 1u : 0u);
}

// State a prepared query starts the stepping loop with (= one LDS queue entry).
struct Prepared {
  float qx, qy, qz;
  float best_d;       // running best (kdtree.go neighbor1.DistSq)
  float bound_d;      // pruning bound: min(best, ub) in exact mode, = best otherwise
  float4 best;        // record of the current best {x, y, z, bits(id)}, id < 0 = none
  uint32_t path_b;    // deepest verified node of the first descent, plus the flags below
  uint32_t pend;      // first-descent levels still to examine
};
constexpr uint32_t kPathFinished = 0x80000000u;  // query finished inside prepare (MinDistSq cut)
constexpr uint32_t kPathDescend = 0x40000000u;   // prediction failed at path_b: descend from there
constexpr uint32_t kPathMask = 0x07FFFFFFu;

// One float of node b; 32-bit byte offset (tree < 2^27 slots x 16 B) so the load uses
// scalar-base + 32-bit-offset addressing instead of a 64-bit address per lane.
__device__ __forceinline__ float node_comp(const float4 *nodes, uint32_t b, int dim) {
  const uint32_t off = (b << 4) + 4u * (uint32_t)dim;
  return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(nodes) + off);
}
__device__ __forceinline__ float4 node_at(const float4 *nodes, uint32_t b) {
  const uint32_t off = b << 4;
  return *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(nodes) + off);
}

constexpr int kTopLevels = kWalkTopLevels;     // split values of levels 0..10 (BFS 1..2047) live in LDS
constexpr int kTopEntries = 1 << kTopLevels;   // 2048 floats = 8 KB per workgroup
constexpr int kDeepGroup = 5;                  // deeper levels are fetched in groups of this many
static_assert((kMaxLevels - kTopLevels) % kDeepGroup == 0, "deep levels must split into whole groups");

// Fills the workgroup's LDS copy of the top split values (call before walk_queries, then sync).
__device__ __forceinline__ void load_top_levels(const TreeView &tv, float *__restrict__ top) {
  const uint32_t slots = 1u << tv.depth;  // BFS slots that exist
  for (uint32_t b = threadIdx.x; b < (uint32_t)kTopEntries; b += blockDim.x) {
    float v = 0.0f;
    if (b >= 1u && b < slots) v = node_comp(tv.nodes, b, (31 - __clz((int)b)) % 3);
    top[b] = v;
  }
}

template <bool kExact>
__device__ __forceinline__ Prepared prepare_query(const TreeView &tv, const float *__restrict__ top, float qx,
                                                  float qy, float qz, float max_range_sq, float min_dist_sq,
                                                  float ub_hint) {
  Prepared p;
  p.qx = qx; p.qy = qy; p.qz = qz;
  p.best_d = max_range_sq;
  p.best = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(-1));
  const uint32_t m1 = (uint32_t)tv.n + 1u;
  // --- predicted leaf from the grid directory
  const int g = tv.dir_bits;
  const float cmax = (float)((1 << g) - 1);
  const float fx = fminf(fmaxf((qx - tv.dir_lo[0]) * tv.dir_scale[0], 0.0f), cmax);
  const float fy = fminf(fmaxf((qy - tv.dir_lo[1]) * tv.dir_scale[1], 0.0f), cmax);
  const float fz = fminf(fmaxf((qz - tv.dir_lo[2]) * tv.dir_scale[2], 0.0f), cmax);
  const uint32_t cell = (uint32_t)fx | ((uint32_t)fy << g) | ((uint32_t)fz << (2 * g));
  const uint32_t bl = tv.dir[cell];
  const int L = 31 - __clz((int)bl);  // depth of the predicted leaf = inner levels above it
  // The predicted path extended to kMaxLevels: its ancestor at level j is ext >> (kMaxLevels - j)
  // for every j; for j >= L that is a slot without a node (those levels are masked off below).
  // Bit (kMaxLevels - 1 - j) of ext says whether the predicted path takes child1 at level j; the
  // per-level masks below use the same (root-highest) bit order.
  const uint32_t ext = bl << (kMaxLevels - L);

  // --- ONE round of independent fetches: the predicted leaf and the split values of the
  //     predicted path (top levels from LDS; deeper ones from memory in groups of kDeepGroup
  //     levels, a group is skipped when the tree ends above it; slots exist for j < 2^depth)
  const float4 leaf = node_at(tv.nodes, bl);
  float pv[kMaxLevels];
#pragma unroll
  for (int j = 0; j < kTopLevels; j++) pv[j] = top[ext >> (kMaxLevels - j)];
#pragma unroll
  for (int g0 = kTopLevels; g0 < kMaxLevels; g0 += kDeepGroup) {
    if (g0 < tv.depth) {  // wave-uniform
#pragma unroll
      for (int j = g0; j < g0 + kDeepGroup; j++) {
        // the last group may reach past the slots of a tree whose depth is not a group boundary
        const uint32_t idx = ext >> (kMaxLevels - j);
        pv[j] = node_comp(tv.nodes, idx & ((1u << tv.depth) - 1u), j % 3);
      }
    } else {
#pragma unroll
      for (int j = g0; j < g0 + kDeepGroup; j++) pv[j] = 0.0f;
    }
  }
  const float ldx = leaf.x - qx, ldy = leaf.y - qy, ldz = leaf.z - qz;
  const float d_leaf = (ldx * ldx + ldy * ldy) + ldz * ldz;
  // Bound for recording a level in `pend`: the best while the reference descends (= maxRange^2),
  // tightened by ub in exact mode (header, "Pruning bound").  fminf drops a NaN operand: a NaN
  // hint or leaf distance simply does not tighten the bound.
  const float ub = fminf(d_leaf, ub_hint);
  const float bound0 = kExact ? fminf(max_range_sq, ub) : max_range_sq;

  // --- per level: does the real descent go to child1 (kdtree.go:216: pivotVal > val -> child0),
  //     and can the level's plane test still pass.  One bit per level, shifted in root first.
  uint32_t right = 0, near = 0;
#pragma unroll
  for (int j = 0; j < kMaxLevels; j++) {
    const float qv = (j % 3 == 0) ? qx : ((j % 3 == 1) ? qy : qz);
    right = right + right + ((pv[j] > qv) ? 0u : 1u);
    const float fp = qv - pv[j];
    near = near + near + ((fp * fp > bound0) ? 0u : 1u);
  }
  const uint32_t valid = ((1u << L) - 1u) << (kMaxLevels - L);  // levels above the predicted leaf
  // only child (kdtree.go:208-211): a node of size 2 sends every query to child0
  if (L > 0 && node_size(bl >> 1, L - 1, m1) == 2u) right &= ~(1u << (kMaxLevels - L));
  const uint32_t wrong = (right ^ ext) & valid;
  const bool verified = wrong == 0u;
  const int m = verified ? L : __clz((int)wrong) - (32 - kMaxLevels);  // first level where the prediction fails

  // --- the leaf (kdtree.go:95-106), only if the real descent arrives there
  bool finished = false;
  if (verified) {
    if (!(d_leaf > p.best_d)) {
      p.best_d = d_leaf;
      p.best = leaf;
    }
    if (!kExact && p.best_d < min_dist_sq) finished = true;
  }
  p.bound_d = kExact ? fminf(p.best_d, ub) : p.best_d;
  p.path_b = (bl >> (L - m)) | (finished ? kPathFinished : 0u) | (verified ? 0u : kPathDescend);
  // only levels above the mismatch; stored with bit j = level j
  p.pend = (near >> (kMaxLevels - m)) == 0u ? 0u : __brev(near >> (kMaxLevels - m)) >> ((32 - m) & 31);
  return p;
}

constexpr int kQueueWords = 10;  // LDS queue entry words (SoA [word][slot], 64 slots per wave)
static_assert(kQueueWords * 64 * 4 == kWalkQueueBytesPerWave, "pcgx_internal.h kWalkQueueBytesPerWave");

// Chunk range [begin, end) of workgroup `bid` out of `nblocks` (a multiple of 8, or < 8):
// workgroups with equal bid % 8 get adjacent ranges.
__device__ __forceinline__ void block_chunk_range(int64_t nq, uint32_t bid, uint32_t nblocks,
                                                  uint32_t &begin, uint32_t &end) {
  const uint64_t n_chunks = (uint64_t)((nq + 63) / 64);
  uint32_t slot = bid;
  if ((nblocks & 7u) == 0u) slot = (bid & 7u) * (nblocks >> 3) + (bid >> 3);
  begin = (uint32_t)(n_chunks * slot / nblocks);
  end = (uint32_t)(n_chunks * (slot + 1) / nblocks);
}

// fetch(idx, qx, qy, qz, ub): loads query idx; ub = squared distance (the walk's own float32
// expression) from the query to ANY point stored in the tree, or +inf: a pruning hint used in
// exact mode only (header, "Pruning bound").   emit(idx, best, best_d): consumes the result;
// best = {x, y, z, bits(id)} of the matched base point, id < 0 = no match.
// `queue`: this wave's LDS queue (kQueueWords * 64 words); `top`: the workgroup's LDS copy of the
// top split values (load_top_levels); `next_chunk`: the workgroup's LDS chunk counter,
// initialised to the first chunk of its range [.., chunk_end).
// Optional instrumentation (kStats): per-wave counts, one row of 24 words per wave in stats[]
// (summed by the host, [15] and [18] maximised):
// [0] loop iterations, [1] active lanes summed over iterations, [2] node fetches (lanes),
// [3] emit/refill sections run, [4] chunks prepared, [5] queries that verified down to the leaf,
// [6] first-descent levels recorded in pend, [7] queries, [8] fetches while descending,
// [9] explicit-frame pops, [10] of those passing the plane test, [11] first-descent pops,
// [12] of those passing, [13] leaves evaluated in the loop, [14] iterations after the wave's last
// query was handed out, [15] most iterations of any wave, [16] / [17] 100 MHz ticks summed over
// waves before / after that hand-out, [18] longest wave (ticks), [19] waves.
template <bool kMinDist, bool kStats = false, class Fetch, class Emit>
__device__ __forceinline__ void walk_queries(const TreeView tv, uint32_t *__restrict__ stk,
                                             const int stk_stride, uint32_t *__restrict__ queue,
                                             const float *__restrict__ top, const int64_t nq,
                                             uint32_t *__restrict__ next_chunk, const uint32_t chunk_end,
                                             const float max_range_sq, const float min_dist_sq,
                                             Fetch &&fetch, Emit &&emit,
                                             unsigned long long *__restrict__ stats = nullptr) {
  constexpr bool kExact = !kMinDist;
  unsigned long long st_iter = 0, st_active = 0, st_look = 0, st_refill = 0, st_prep = 0, st_verified = 0,
                     st_pend = 0, st_queries = 0, st_desc = 0, st_epop = 0, st_epass = 0, st_ipop = 0, st_ipass = 0,
                     st_leaf = 0, st_tail = 0;
  const int lane = (int)(threadIdx.x & 63u);
  const uint64_t lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  const uint32_t m1 = (uint32_t)tv.n + 1u;
  bool exhausted = false;           // wave-uniform: the workgroup's range is used up
  int32_t q_head = 0, q_count = 0;  // wave-uniform: LDS queue state
  int64_t q_base = 0;               // wave-uniform: first query of the queued chunk

  bool active = false, pending = false, desc = false;
  int64_t my_q = 0;
  float qx = 0.0f, qy = 0.0f, qz = 0.0f, best_d = 0.0f, bound_d = 0.0f;
  float4 best = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(-1));
  uint32_t b = 1, path_b = 1, pend = 0;
  int32_t sp = 0;

  const unsigned long long t_start = kStats ? wall_clock64() : 0ull;
  unsigned long long t_dry = 0ull;  // when the wave's last query was handed out
  for (;;) {
    // ---- emit finished lanes, prepare more queries, refill ------------------------
    // Divergent sections, so they run only once enough lanes are waiting (or nothing
    // is left to step): a lane that finished idles until then.
    const uint64_t idle = __ballot(!active);
    const int n_idle = __popcll(idle);
    if (kStats) {
      st_iter += 1;
      st_active += (unsigned long long)(64 - n_idle);
      if (exhausted && q_head == q_count) {
        st_tail += 1;
        if (t_dry == 0ull) t_dry = wall_clock64();
      }
    }
    if (n_idle >= tv.refill_threshold || n_idle == 64) {
      if (kStats) st_refill += 1;
      if (pending) {
        emit(my_q, best, best_d);
        pending = false;
      }
      if (q_head == q_count && !exhausted) {
        // queue empty: take the next chunk; all 64 lanes prepare one query each, whatever
        // they are walking
        uint32_t c = 0;
        if (lane == 0) c = atomicAdd(next_chunk, 1u);  // LDS atomic
        c = __builtin_amdgcn_readfirstlane(c);
        q_head = 0;
        q_count = 0;
        if (c < chunk_end) {
          q_base = (int64_t)c * 64;
          const int64_t left = nq - q_base;
          q_count = left >= 64 ? 64 : (int32_t)left;
        }
        exhausted = c + 1u >= chunk_end;
        if (kStats) st_prep += 1;
        const int64_t idx = q_base + lane;
        if (lane < q_count) {
          float x, y, z, ub_hint;
          fetch(idx, x, y, z, ub_hint);
          const Prepared p = prepare_query<kExact>(tv, top, x, y, z, max_range_sq, min_dist_sq, ub_hint);
          if (kStats) {
            st_queries += 1;
            st_verified += (p.path_b & kPathDescend) == 0u;
            st_pend += (unsigned long long)__popc(p.pend);
          }
          queue[0 * 64 + lane] = __float_as_uint(p.qx);
          queue[1 * 64 + lane] = __float_as_uint(p.qy);
          queue[2 * 64 + lane] = __float_as_uint(p.qz);
          // best_d is implied: the distance to `best` if there is one (recomputed bit-identically
          // when the entry is taken), else maxRange^2
          queue[3 * 64 + lane] = __float_as_uint(p.bound_d);
          queue[4 * 64 + lane] = p.path_b;
          queue[5 * 64 + lane] = p.pend;
          queue[6 * 64 + lane] = __float_as_uint(p.best.x);
          queue[7 * 64 + lane] = __float_as_uint(p.best.y);
          queue[8 * 64 + lane] = __float_as_uint(p.best.z);
          queue[9 * 64 + lane] = __float_as_uint(p.best.w);
        }
      }
      if (!active) {
        const int32_t slot = q_head + (int32_t)__popcll(idle & lt_mask);
        if (slot < q_count) {
          qx = __uint_as_float(queue[0 * 64 + slot]);
          qy = __uint_as_float(queue[1 * 64 + slot]);
          qz = __uint_as_float(queue[2 * 64 + slot]);
          bound_d = __uint_as_float(queue[3 * 64 + slot]);
          const uint32_t pw = queue[4 * 64 + slot];
          pend = queue[5 * 64 + slot];
          best.x = __uint_as_float(queue[6 * 64 + slot]);
          best.y = __uint_as_float(queue[7 * 64 + slot]);
          best.z = __uint_as_float(queue[8 * 64 + slot]);
          best.w = __uint_as_float(queue[9 * 64 + slot]);
          const float bx = best.x - qx, by = best.y - qy, bz = best.z - qz;
          best_d = __float_as_int(best.w) >= 0 ? (bx * bx + by * by) + bz * bz : max_range_sq;
          my_q = q_base + slot;
          sp = 0;
          path_b = pw & kPathMask;
          if (pw & kPathFinished) {  // MinDistSq cut at the first leaf: nothing left to walk
            pending = true;
          } else {
            b = path_b;
            desc = (pw & kPathDescend) != 0u;
            active = true;
          }
        }
      }
      q_head = q_head + n_idle < q_count ? q_head + n_idle : q_count;
      if (q_head == q_count && exhausted && __ballot(active) == 0ull) {
        if (pending) emit(my_q, best, best_d);
        break;  // batch exhausted, every lane done and emitted
      }
    }

    // ---- unwinding lanes: the topmost explicit frame, else the deepest pending level of the
    //      first descent (kdtree.go:107-110); either way the frame is the child c the descent
    //      took below the ancestor c >> 1.  Its plane test follows the node fetch.
    const bool popping = active && !desc;
    const bool has_exp = sp > 0;
    const uint32_t fw = stk[(has_exp ? sp - 1 : 0) * stk_stride];
    const bool implicit = popping && !has_exp && pend != 0u;
    bool finish = popping && !has_exp && pend == 0u;
    sp = (popping && has_exp) ? sp - 1 : sp;
    const int m = 31 - __clz((int)path_b);
    const int j = 31 - __clz((int)(pend | 1u));
    pend = implicit ? (pend & ~(1u << j)) : pend;
    const uint32_t c = has_exp ? fw : (path_b >> ((m - j - 1) & 31));

    // ---- the one node fetch of this step -----------------------------------------
    const uint32_t at = desc ? b : (c >> 1);
    const bool look = active && (desc || has_exp || implicit);
    if (kStats) st_look += look;
    if (look) {
      const float4 nd = node_at(tv.nodes, at);
      const float dx = nd.x - qx, dy = nd.y - qy, dz = nd.z - qz;  // pivot.Sub(p)
      const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
      const float d = (xx + yy) + zz;                                // NormSq, mat/vec3.go:18-20
      const int32_t depth = 31 - __clz((int)at);
      const uint32_t n = node_size(at, depth, m1);
      const bool leaf = desc && n == 1u;
      const bool inner = desc && n != 1u;
      const int dim = depth % 3;
      // fp = q[dim] - pivot[dim] = -(pivot[dim] - q[dim]) exactly, so fp * fp is the square already
      // formed for the distance, and pivotVal > val <=> pivot[dim] - q[dim] > 0 (float32
      // denormals are on for this code object: the difference is zero only for equal operands)
      const float fp2 = sel3(dim, xx, yy, zz);
      const float sd = sel3(dim, dx, dy, dz);
      const bool plane_ok = !(fp2 > bound_d);  // kdtree.go:111-115
      if (kStats) {
        st_desc += desc;
        st_leaf += leaf;
        st_epop += !desc && has_exp;
        st_epass += !desc && has_exp && plane_ok;
        st_ipop += !desc && !has_exp;
        st_ipass += !desc && !has_exp && plane_ok;
      }
      // leaf: replace unless d > best (kdtree.go:95-103); pivot: replace if d < best (:116-119)
      const bool take = leaf ? !(d > best_d) : (!desc && plane_ok && d < best_d);
      best_d = take ? d : best_d;
      bound_d = fminf(bound_d, best_d);
      best.x = take ? nd.x : best.x;
      best.y = take ? nd.y : best.y;
      best.z = take ? nd.z : best.z;
      best.w = take ? nd.w : best.w;
      if (kMinDist) finish = finish || ((leaf || take) && best_d < min_dist_sq);  // :104,120,140

      // descending through an inner node: searchLeafNode step (kdtree.go:202-221)
      const bool go_left = n == 2u || sd > 0.0f;  // only child, or pivotVal > val -> child0
      const uint32_t child = 2u * at + (go_left ? 0u : 1u);
      if (inner && plane_ok) {
        stk[sp * stk_stride] = child;
        ++sp;
      }
      // unwinding through a node that passed the plane test: the other side (kdtree.go:124-137);
      // n == 2: single child, nextNode == nil (:130-132)
      b = desc ? child : (c ^ 1u);
      desc = desc ? inner : (plane_ok && n != 2u);
    }

    if (finish) {
      active = false;
      pending = true;
    }
  }
  unsigned long long t_end = 0ull;
  if (kStats) {  // read the clock before the contended atomics below; the asm keeps it there
    t_end = wall_clock64();
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(t_end)::"memory");
  }
  if (kStats && stats) {
    // per-lane counters: sum over the wave; wave-uniform ones are taken from lane 0
    for (int o = 32; o > 0; o >>= 1) {
      st_look += __shfl_down(st_look, o);
      st_verified += __shfl_down(st_verified, o);
      st_pend += __shfl_down(st_pend, o);
      st_queries += __shfl_down(st_queries, o);
      st_desc += __shfl_down(st_desc, o);
      st_epop += __shfl_down(st_epop, o);
      st_epass += __shfl_down(st_epass, o);
      st_ipop += __shfl_down(st_ipop, o);
      st_ipass += __shfl_down(st_ipass, o);
      st_leaf += __shfl_down(st_leaf, o);
    }
    if (lane == 0) {  // one row of 24 counters per wave: the host adds them up (no contended atomics)
      unsigned long long *row =
          stats + ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 24;
      if (t_dry == 0ull) t_dry = t_end;
      row[0] = st_iter; row[1] = st_active; row[2] = st_look; row[3] = st_refill; row[4] = st_prep;
      row[5] = st_verified; row[6] = st_pend; row[7] = st_queries; row[8] = st_desc; row[9] = st_epop;
      row[10] = st_epass; row[11] = st_ipop; row[12] = st_ipass; row[13] = st_leaf; row[14] = st_tail;
      row[15] = st_iter; row[16] = t_dry - t_start; row[17] = t_end - t_dry; row[18] = t_end - t_start;
      row[19] = 1ull;
    }
  }
}

}  // namespace pcgx
