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
// split values of the top kWalkTopLevels (6) levels come from an LDS copy per workgroup.  The
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
//     hence equals the reference's, exact ties included (HISTORY.md 3.1).
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

// Result of the per-lane part of a chunk's preparation (prepare_query).
struct Prepared {
  float bound0;       // bound for recording first-descent levels (maxRange^2, ub folded in when exact)
  float ub;           // min(distance to the predicted leaf's point, caller's hint): >= d*
  float4 leaf;        // the predicted leaf's record
  float d_leaf;       // its distance
  uint32_t path_b;    // deepest node of the predicted path the real descent is known to reach
  int32_t depth;      // its depth
  uint32_t pend;      // levels above it whose plane test can still pass (bit j = level j)
  bool verified;      // path_b is the predicted leaf: the prediction was right
};
// Queue entry word 4: BFS index of the deepest verified node | flags
constexpr uint32_t kPathDescend = 0x40000000u;    // the first descent is not finished: descend from path_b
constexpr uint32_t kPathLeafTaken = 0x20000000u;  // path_b is the first leaf and it is the running best
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

constexpr int kTopLevels = kWalkTopLevels;     // split values of levels 0..kTopLevels-1 live in LDS
constexpr int kTopEntries = 1 << kTopLevels;   // 64 floats = 256 B per workgroup at the default 6 levels
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

// Per-lane part of the preparation: predicted leaf (the caller's `pred` if it has one, else the
// grid directory), one round of fetches for the predicted path, verification.
template <bool kExact>
__device__ __forceinline__ Prepared prepare_query(const TreeView &tv, const float *__restrict__ top, float qx,
                                                  float qy, float qz, float max_range_sq, float ub_hint,
                                                  uint32_t pred) {
  Prepared p;
  const uint32_t m1 = (uint32_t)tv.n + 1u;
  uint32_t bl = pred;
  if (pred == 0u || pred >= (1u << tv.depth)) {  // no (usable) prediction from the caller: grid directory
    const int g = tv.dir_bits;
    const float cmax = (float)((1 << g) - 1);
    const float fx = fminf(fmaxf((qx - tv.dir_lo[0]) * tv.dir_scale[0], 0.0f), cmax);
    const float fy = fminf(fmaxf((qy - tv.dir_lo[1]) * tv.dir_scale[1], 0.0f), cmax);
    const float fz = fminf(fmaxf((qz - tv.dir_lo[2]) * tv.dir_scale[2], 0.0f), cmax);
    const uint32_t cell = (uint32_t)fx | ((uint32_t)fy << g) | ((uint32_t)fz << (2 * g));
    bl = tv.dir[cell];
  }
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
  const float ub = kExact ? fminf(d_leaf, ub_hint) : __builtin_inff();
  const float bound0 = fminf(max_range_sq, ub);

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

  p.bound0 = bound0;
  p.ub = ub;
  p.leaf = leaf;
  p.d_leaf = d_leaf;
  p.path_b = bl >> (L - m);
  p.depth = m;
  // only levels above the mismatch; stored with bit j = level j
  p.pend = __brev(near >> (kMaxLevels - m)) >> ((32 - m) & 31);
  p.verified = verified;
  return p;
}

constexpr int kStatWords = 32;    // instrumentation: counters per wave
constexpr int kQueueWords = 7;    // LDS queue entry words (SoA [word][slot])
constexpr int kQueueSlots = 128;  // ring of prepared queries per wave: < 64 left over + <= 64 new
static_assert(kQueueWords * kQueueSlots * 4 == kWalkQueueBytesPerWave, "pcgx_internal.h kWalkQueueBytesPerWave");

// Chunk range [begin, end) of workgroup `bid` out of `nblocks` (a multiple of 8, or < 8):
// workgroups with equal bid % 8 get adjacent ranges.
__device__ __forceinline__ uint32_t block_slot(uint32_t bid, uint32_t nblocks) {
  return (nblocks & 7u) == 0u ? (bid & 7u) * (nblocks >> 3) + (bid >> 3) : bid;
}
__device__ __forceinline__ void block_chunk_range(int64_t nq, uint32_t bid, uint32_t nblocks,
                                                  uint32_t &begin, uint32_t &end) {
  const uint64_t n_chunks = (uint64_t)((nq + 63) / 64);
  const uint32_t slot = block_slot(bid, nblocks);
  begin = (uint32_t)(n_chunks * slot / nblocks);
  end = (uint32_t)(n_chunks * (slot + 1) / nblocks);
}
// the slot whose chunk range holds query i, and that range's first chunk
__device__ __forceinline__ uint32_t slot_of_query(int64_t nq, int64_t i, uint32_t nblocks, uint32_t &begin) {
  const uint64_t n_chunks = (uint64_t)((nq + 63) / 64);
  const uint64_t c = (uint64_t)i / 64;
  const uint32_t slot = (uint32_t)(((c + 1) * nblocks - 1) / n_chunks);
  begin = (uint32_t)(n_chunks * slot / nblocks);
  return slot;
}

// fetch(idx, qx, qy, qz, ub, pred): loads query idx.  ub = squared distance (the walk's own float32
// expression) from the query to ANY point stored in the tree, or +inf: a pruning hint used in
// exact mode only (header, "Pruning bound").  pred = BFS index of the LEAF the caller expects the
// query's first descent to end in, or 0 (then the grid directory predicts it).
// emit(idx, best, best_d): consumes the result; best = {x, y, z, bits(id)} of the matched base
// point, id < 0 = no match.   note_leaf(idx, leaf): the leaf the first descent really ended in, or
// 0 if the preparation did not get that far (a caller may feed it back as `pred` next time).
// `queue`: this wave's LDS queue (kQueueWords * kQueueSlots words); `top`: the workgroup's LDS copy
// of the top split values (load_top_levels); `next_chunk`: the workgroup's LDS chunk counter,
// initialised to the first chunk of its range [range_first / 64, chunk_end).
// Optional instrumentation (kStats): per-wave counts, one row of kStatWords words per wave in stats[]
// (summed by the host, [15] and [18] maximised):
// [0] loop iterations, [1] active lanes summed over iterations, [2] node fetches (lanes),
// [3] emit/refill sections run, [4] chunks prepared, [5] queries whose prediction verified,
// [6] first-descent levels recorded in pend, [7] queries, [8] fetches while descending,
// [9] explicit-frame pops, [10] of those passing the plane test, [11] first-descent pops,
// [12] of those passing, [13] leaves evaluated in the loop, [14] iterations after the wave's last
// query was handed out, [15] most iterations of any wave, [16] / [17] 100 MHz ticks before / after
// that hand-out, [18] longest wave (ticks), [19] waves, [20] (host) kernel ns, [21] queries finished
// inside the preparation, [22] lane-steps of the preparation's descent loop, [23] its iterations,
// [24..29] ticks in: query fetch, path fetch + verification, descent loop, first leaf + emit +
// enqueue, taking queued queries, stepping.
template <bool kMinDist, bool kStats = false, class Fetch, class Emit, class NoteLeaf>
__device__ __forceinline__ void walk_queries(const TreeView tv, uint32_t *__restrict__ stk,
                                             const int stk_stride, uint32_t *__restrict__ queue,
                                             const float *__restrict__ top, const int64_t nq,
                                             uint32_t *__restrict__ next_chunk, const uint32_t chunk_end,
                                             const int64_t range_first, const float max_range_sq,
                                             const float min_dist_sq, Fetch &&fetch, Emit &&emit,
                                             NoteLeaf &&note_leaf, unsigned long long *__restrict__ stats = nullptr) {
  constexpr bool kExact = !kMinDist;
  unsigned long long st_iter = 0, st_active = 0, st_look = 0, st_refill = 0, st_prep = 0, st_verified = 0,
                     st_pend = 0, st_queries = 0, st_desc = 0, st_epop = 0, st_epass = 0, st_ipop = 0, st_ipass = 0,
                     st_leaf = 0, st_tail = 0, st_early = 0, st_tsteps = 0, st_titer = 0;
  unsigned long long tk_fetch = 0, tk_path = 0, tk_tight = 0, tk_finish = 0, tk_take = 0, tk_step = 0, tk_mark = 0;
  auto lap = [&](unsigned long long &acc) {  // kStats: time since the previous lap goes to acc
    if (kStats) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      const unsigned long long now = wall_clock64();
      acc += now - tk_mark;
      tk_mark = now;
    }
  };
  const int lane = (int)(threadIdx.x & 63u);
  const uint64_t lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  const uint32_t m1 = (uint32_t)tv.n + 1u;
  bool exhausted = false;            // wave-uniform: the workgroup's range is used up
  uint32_t q_head = 0, q_tail = 0;   // wave-uniform: LDS ring [q_head, q_tail), slot = index % kQueueSlots

  bool active = false, pending = false, desc = false;
  int64_t my_q = 0;
  float qx = 0.0f, qy = 0.0f, qz = 0.0f, best_d = 0.0f, bound_d = 0.0f;
  float4 best = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(-1));
  uint32_t b = 1, path_b = 1, pend = 0;
  int32_t sp = 0;

  const unsigned long long t_start = kStats ? wall_clock64() : 0ull;
  tk_mark = t_start;
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
      if (exhausted && q_head == q_tail) {
        st_tail += 1;
        if (t_dry == 0ull) t_dry = wall_clock64();
      }
    }
    lap(tk_step);
    if (exhausted && q_head == q_tail) {
      // nothing left to hand out: finished lanes keep their result until the wave is done
      if (n_idle == 64) {
        if (pending) emit(my_q, best, best_d);
        break;
      }
    } else if (n_idle >= tv.refill_threshold) {
      if (kStats) st_refill += 1;
      if (pending) {
        emit(my_q, best, best_d);
        pending = false;
      }
      // ---- prepare chunks while the queue cannot serve every waiting lane (at most
      //      chunks_per_refill in a row).  All 64 lanes prepare one query each, whatever they are
      //      walking (full lane efficiency).
      for (int k = 0; k < tv.chunks_per_refill && (int)(q_tail - q_head) < n_idle && !exhausted; k++) {
        uint32_t c = 0;
        if (lane == 0) c = atomicAdd(next_chunk, 1u);  // LDS atomic
        c = __builtin_amdgcn_readfirstlane(c);
        int32_t cnt = 0;
        const int64_t c_base = (int64_t)c * 64;
        if (c < chunk_end) {
          const int64_t left = nq - c_base;
          cnt = left >= 64 ? 64 : (int32_t)left;
        }
        exhausted = c + 1u >= chunk_end;
        if (kStats) st_prep += 1;
        const int64_t idx = c_base + lane;
        const bool valid = lane < cnt;
        float x = 0.0f, y = 0.0f, z = 0.0f, ub_hint = __builtin_inff();
        uint32_t pred = 0;
        Prepared p = {};
        bool unresolved = false;
        if (valid) fetch(idx, x, y, z, ub_hint, pred);
        lap(tk_fetch);
        if (valid) {
          p = prepare_query<kExact>(tv, top, x, y, z, max_range_sq, ub_hint, pred);
          unresolved = !p.verified;
          if (kStats) {
            st_queries += 1;
            st_verified += p.verified;
          }
        }
        // ---- wrong prediction: the real descent continues below the mismatch with the
        //      reference's steps (kdtree.go:202-221), all such lanes of the chunk in lockstep and
        //      for a bounded number of levels; whoever is still not at a leaf finishes the
        //      descent in the stepping loop.
        lap(tk_path);
        uint32_t tb = p.path_b, tpend = p.pend;
        int32_t td = p.depth;
        bool at_leaf = valid && p.verified;
        for (int it = 0; it < tv.tight_levels && __ballot(unresolved) != 0ull; it++) {
          if (kStats) {
            st_titer += 1;
            st_tsteps += unresolved;
          }
          if (unresolved) {
            const uint32_t n = node_size(tb, td, m1);
            if (n == 1u) {
              unresolved = false;
              at_leaf = true;
            } else {
              const int dim = td % 3;
              const float pvv = node_comp(tv.nodes, tb, dim);
              const float qv = sel3(dim, x, y, z);
              const float fp = qv - pvv;
              if (!(fp * fp > p.bound0)) tpend |= 1u << td;
              tb = 2u * tb + ((n == 2u || pvv > qv) ? 0u : 1u);
              ++td;
            }
          }
        }
        if (valid && unresolved && node_size(tb, td, m1) == 1u) {  // reached a leaf on the last allowed step
          unresolved = false;
          at_leaf = true;
        }
        lap(tk_tight);
        // ---- the first leaf (kdtree.go:95-106)
        float4 lf = p.leaf;
        float d_lf = p.d_leaf;
        if (at_leaf && !p.verified) {
          lf = node_at(tv.nodes, tb);
          const float ldx = lf.x - x, ldy = lf.y - y, ldz = lf.z - z;
          d_lf = (ldx * ldx + ldy * ldy) + ldz * ldz;
        }
        const bool leaf_taken = at_leaf && !(d_lf > max_range_sq);
        const float bd = leaf_taken ? d_lf : max_range_sq;  // the running best after the first leaf
        if (valid) note_leaf(idx, at_leaf ? tb : 0u);
        if (kStats && valid) st_pend += (unsigned long long)__popc(tpend);
        // finished already: nothing left to unwind (or the MinDistSq cut, kdtree.go:104)
        const bool done = at_leaf && (tpend == 0u || (kMinDist && bd < min_dist_sq));
        if (valid && done) {
          if (kStats) st_early += 1;
          float4 r = lf;
          if (!leaf_taken) r = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(-1));
          emit(idx, r, bd);
        }
        const bool enq = valid && !done;
        const uint64_t enq_mask = __ballot(enq);
        if (enq) {
          const uint32_t slot = (q_tail + (uint32_t)__popcll(enq_mask & lt_mask)) % kQueueSlots;
          queue[0 * kQueueSlots + slot] = __float_as_uint(x);
          queue[1 * kQueueSlots + slot] = __float_as_uint(y);
          queue[2 * kQueueSlots + slot] = __float_as_uint(z);
          queue[3 * kQueueSlots + slot] = __float_as_uint(kExact ? fminf(bd, p.ub) : bd);  // bound_d
          queue[4 * kQueueSlots + slot] = tb | (at_leaf ? 0u : kPathDescend) | (leaf_taken ? kPathLeafTaken : 0u);
          queue[5 * kQueueSlots + slot] = tpend;
          queue[6 * kQueueSlots + slot] = (uint32_t)(idx - range_first);
        }
        q_tail += (uint32_t)__popcll(enq_mask);
        lap(tk_finish);
      }
      // ---- waiting lanes take queued queries
      const int32_t avail = (int32_t)(q_tail - q_head);
      if (!active) {
        const int32_t r = (int32_t)__popcll(idle & lt_mask);
        if (r < avail) {
          const uint32_t slot = (q_head + (uint32_t)r) % kQueueSlots;
          qx = __uint_as_float(queue[0 * kQueueSlots + slot]);
          qy = __uint_as_float(queue[1 * kQueueSlots + slot]);
          qz = __uint_as_float(queue[2 * kQueueSlots + slot]);
          bound_d = __uint_as_float(queue[3 * kQueueSlots + slot]);
          const uint32_t pw = queue[4 * kQueueSlots + slot];
          pend = queue[5 * kQueueSlots + slot];
          my_q = range_first + (int64_t)queue[6 * kQueueSlots + slot];
          path_b = pw & kPathMask;
          best = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(-1));
          best_d = max_range_sq;
          if (pw & kPathLeafTaken) {  // the running best is the first leaf: its record and distance again
            best = node_at(tv.nodes, path_b);
            const float bx = best.x - qx, by = best.y - qy, bz = best.z - qz;
            best_d = (bx * bx + by * by) + bz * bz;
          }
          sp = 0;
          b = path_b;
          desc = (pw & kPathDescend) != 0u;
          active = true;
        }
      }
      q_head += (uint32_t)(n_idle < avail ? n_idle : avail);
      lap(tk_take);
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
      st_early += __shfl_down(st_early, o);
      st_tsteps += __shfl_down(st_tsteps, o);
    }
    if (lane == 0) {  // one row of 24 counters per wave: the host adds them up (no contended atomics)
      unsigned long long *row =
          stats + ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * kStatWords;
      if (t_dry == 0ull) t_dry = t_end;
      row[0] = st_iter; row[1] = st_active; row[2] = st_look; row[3] = st_refill; row[4] = st_prep;
      row[5] = st_verified; row[6] = st_pend; row[7] = st_queries; row[8] = st_desc; row[9] = st_epop;
      row[10] = st_epass; row[11] = st_ipop; row[12] = st_ipass; row[13] = st_leaf; row[14] = st_tail;
      row[15] = st_iter; row[16] = t_dry - t_start; row[17] = t_end - t_dry; row[18] = t_end - t_start;
      row[19] = 1ull; row[21] = st_early; row[22] = st_tsteps; row[23] = st_titer;
      row[24] = tk_fetch; row[25] = tk_path; row[26] = tk_tight; row[27] = tk_finish; row[28] = tk_take;
      row[29] = tk_step;
    }
  }
}

}  // namespace pcgx
