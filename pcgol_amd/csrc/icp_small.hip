// icp_small.hip -- PointToPointICPGradient.Fit for SMALL clouds in ONE launch.
//
// Reference: pc/registration/icp/icp.go:23-67 (Fit), correspondence.go:22-37 (Pairs), evaluator.go:91-189 (Evaluate),
// updater.go:44-71 (Update); pc/storage/kdtree/kdtree.go:83-146,199-222 (Nearest).  The reference's own benchmark of
// the path, BenchmarkPointToPointICPGradient (icp_test.go:100-142), runs 1024 ... 16384 points with MinDistSq = res^2:
// the approximate search, whose answer depends on the walk's visit order.  The general path (icp.hip + strict.hip)
// takes four to five dependent launches per iteration, ~0.1 ms however few the points, and its walk kernel is built
// for a million queries: 96 us per iteration at 1024 targets, 388 at 16384 (profiles/r05e_rows.json).
//
// Here a Fit is one persistent launch of G = Q * P workgroups (Q = ceil(nt / 64) groups of 64 targets, P workgroups
// a group; G <= 256: resident together on the chip's 256 CUs) that loops over the iterations:
//   search  NOT a walk.  A lane walking its query alone is a chain of dependent fetches -- 140 ... 700 visits of ~200
//          cycles on the benchmark's ground plane, where every third level of the tree ties and nothing is pruned:
//          58 us an iteration at 1024 points (this file's first form) -- while all 1024 x 1024 distances are a million
//          independent evaluations, under a microsecond of the chip's arithmetic.  The walk's answer can be picked
//          out of ALL distances because what it skips cannot change it: kdtree.go:111-115 drops a pivot and its far
//          side only where plane distance^2 > best, every point there is STRICTLY farther than the best (float32
//          squares and their sums are monotone), and a strictly farther point replaces nothing (:100, :117).  So the
//          answer is the full in-order traversal's ( near sub-tree ; node ; far sub-tree , knn_walk.h):
//            * the MinDistSq cut (:104-106,120-122,140-142): the FIRST point in visit order with DistSq < MinDistSq;
//            * none such: the smallest DistSq d*; the first point at d* in visit order takes the best, a LEAF at d*
//              behind it takes it again (:100-103 replaces unless strictly farther, a pivot needs strictly nearer,
//              :117): the LAST leaf at d*, else the first point at d*; d* > maxRange^2: {-1, maxRange^2}; d* ==
//              maxRange^2: leaves only  (tools/model/order_search.py: the rule against the oracle, ties and all).
//          A point's place in the query's visit order is a number: two bits per level from the root down -- 0: in the
//          near child's sub-tree, 2: in the far child's, 1: this node -- "near" by the reference's own comparison at
//          every ancestor (:216; a node of two points has child0 only).  LANE = QUERY, and a wave goes through a
//          sub-tree of four levels (a "chunk", 15 nodes) node by node: the node is the same for all lanes -- the
//          chunk's records arrive by one load and are read lane by lane into scalar registers, the code is straight-
//          line (SmallChunkInfo) -- and a lane's work per node is eight float operations for the distance
//          (mat/vec3.go:18-20,38-40, unfused), the key's digit for the children and two running minima (SmallAcc).
//          What a group's waves can rule out with the reference's own plane test -- every point below a chunk lies
//          beyond the planes of the ancestors on whose far side the chunk hangs -- is not gone through, and where the
//          tree is large not looked at either (small_chunk_go; the queue and the lists in the kernel); the targets are
//          grouped by place for that, whatever order the caller has them in (small_order_kernel).  The minima of a
//          group's waves meet in LDS, those of its P workgroups in its first one's hands, which decides, fetches the
//          partner's record and writes the pair's nine float32 terms (evaluator.go:130-144; strict_terms.h, pair_terms)
//          into rows of the CALLER's target order.  A query that is not finite (an overflowing pose) is walked the
//          reference's way by its lane (small_walk): NaN compares false everywhere and the rule above is about
//          numbers.  (A tree with a NaN point, or maxRange^2 < MinDistSq, is not a small session.)
//   sums   evaluator.go:122-145 adds the terms up in float32, one after the other from 0.0f: a row's first 2048 terms
//          are ONE wave's chain (v_readlane + v_add_f32, 5 ns a term), the tiles of 2048 behind them are summarised by
//          waves of their own meanwhile (namespace mini: strict_sum.h's arithmetic, a wave a tile) and the row's wave
//          walks through their records.
//   update workgroup 0's first wave takes the nine sums, runs the evaluate tail and the pose update (evaluator.go:156-186,
//          updater.go:44-71: icp_update_step, the code the other paths run) and hands the new pose to everybody.
// NO BARRIERS between workgroups.  Everything that passes between them inside the launch -- the workgroups' minima, the
// terms, the tiles' records, the sums, the pose -- is a 64-bit word {payload, tag}, tag = {launch number, iteration + 1},
// stored and polled with agent-scope atomics: a word says by itself whether it is this iteration's, so nobody waits for
// anything but the words they need, and a hand-over is one store and one load's flight (0.5 us one way between any two
// workgroups of the chip, tools/micro/xcd_pingpong.cpp) -- where a grid barrier is a write-back of the L2, a returning
// atomic, a poll and an invalidate: 8 us a barrier by this kernel's own stamps, two an iteration, and 4 us more to read
// the pose past the caches (this file's second form: 55 us an iteration at 1024 points, 6 of them arithmetic).
// Every wait is bounded by wall-clock time and looks at an abort word: a wave that gives up raises it, everybody leaves
// the kernel and the Fit ends with PCGX_E_HIP -- the grid drains whatever happens.  DESIGN.md section 3.1 has the
// measurements (0.22 / 0.58 / 1.24 ms per 10-iteration Fit at 1024 / 4096 / 16384 points; the general path: 0.96 / 1.84 / 3.9).
#include "knn_walk.h"
#include "strict_terms.h"
#include "strict_sum.h"
#include "wg_stamps.h"

namespace pcgx {

#ifndef PCGX_STAMP_ITER
#define PCGX_STAMP_ITER 2  // (the launch's iteration whose phases are stamped)
#endif
PCGX_STAMPS_DECLARE(small_fit, 256, 8)
#if defined(PCGX_STAMPS)
__device__ unsigned long long g_small_dbg[8];  // chunks looked at / ruled out whole / gone through (PCGX_SMALL_COUNTS); [6], [7]: free
__device__ unsigned long long g_small_fail[16];  // the stamped iteration: row r's tiles whose record did not cover the state (low byte: key < 0)
__device__ unsigned long long g_small_iter_t[64];  // workgroup 0's first wave: wall clock at the kernel's start, every iteration's top, the end
#if defined(PCGX_SMALL_COUNTS)  // (an atomic per chunk and wave: not beside time measurements)
#define PCGX_SMALL_COUNT(K) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_small_dbg[K], 1ull); } while (0)
#else
#define PCGX_SMALL_COUNT(K) ((void)0)
#endif
extern "C" __attribute__((visibility("default"))) int pcgx_debug_small_fails(unsigned long long *out) {
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_small_fail), 16 * 8) == hipSuccess ? 0 : 1;
}
extern "C" __attribute__((visibility("default"))) int pcgx_debug_small_iter_times(unsigned long long *out) {
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_small_iter_t), 64 * 8) == hipSuccess ? 0 : 1;
}
extern "C" __attribute__((visibility("default"))) int pcgx_debug_small_counts(unsigned long long *out, int reset) {
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_small_dbg), 64) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[8] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_small_dbg), z, 64) != hipSuccess) return 1;
  }
  return 0;
}
#else
#define PCGX_SMALL_COUNT(K) ((void)0)
#endif  // (measurements: tools/stamps.py small_fit; the launch's second iteration)

#ifndef PCGX_SMALL_BLOCK
#define PCGX_SMALL_BLOCK 512
#endif
constexpr int kSmallBlock = PCGX_SMALL_BLOCK;
constexpr long long kSmallWaitTicks = 200000000;  // 2 s (s_memrealtime: 100 MHz)
constexpr int kPartWords = 6;  // a lane's running minima, as words
constexpr int kSmallTagIterBits = 12;  // a tag: {launch number (20 bits), iteration + 1}

struct SmallSync {  // device words, zero between launches
  unsigned int abort;   // a wave gave up
};
// a tagged word: payload in the low half, tag in the high half
__device__ __forceinline__ unsigned long long tagged(uint32_t payload, uint32_t tag) { return ((unsigned long long)tag << 32) | payload; }
__device__ __forceinline__ unsigned long long word_in(const unsigned long long *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void word_out(unsigned long long *p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
struct SmallWait {
  long long t_first = 0;
  int spins = 0;
};
// between two looks at words that are not there yet: true = give up (somebody raised the abort word, or 2 s are over)
__device__ __forceinline__ bool small_give_up(SmallSync *sy, SmallWait &w) {
  if ((++w.spins & 63) != 0) {
    __builtin_amdgcn_s_sleep(1);
    return false;
  }
  if (__hip_atomic_load(&sy->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return true;
  const long long now = (long long)wall_clock64();
  if (w.t_first == 0) w.t_first = now;
  if (now - w.t_first > kSmallWaitTicks) {
    __hip_atomic_store(&sy->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
  }
  return false;
}
// the launch's scratch block (small_fit_sync_bytes): SmallSync | pose words | sums' words | pair counts | minima
constexpr size_t kSmallPoseAt = 64, kSmallPoseWords = 24;    // trans[16], iter, done
constexpr size_t kSmallSumsAt = 256, kSmallSumsWords = 16;   // S_VALUE ... (the session's slots), pairs at S_PAIRS
constexpr size_t kSmallCountsAt = 512;                        // [256] a group's matched targets
constexpr size_t kSmallExitedAt = 3584;                        // workgroups that have left the launch (the last one zeroes it and the abort word)
constexpr size_t kSmallPartAt = 4096;                         // [256 workgroups][kPartWords][64]
constexpr size_t kSmallPartnersAt = kSmallPartAt + (size_t)256 * kPartWords * 64 * 8;  // [16384] a target's partner last time (a node; 0: none)
constexpr int kSmallMaxTiles = 8;  // 16384 terms in tiles of ss::kTile
constexpr size_t kSmallTileSumsAt = kSmallPartnersAt + (size_t)16384 * 4;                      // [kStrictRows][kSmallMaxTiles][2] a tile's terms' float64 sum
constexpr size_t kSmallTileRecsAt = kSmallTileSumsAt + (size_t)kStrictRows * kSmallMaxTiles * 2 * 8;  // [kStrictRows][kSmallMaxTiles][16] a tile's record (ss::TileRec)
constexpr size_t kSmallScratchBytes = kSmallTileRecsAt + (size_t)kStrictRows * kSmallMaxTiles * 16 * 8;

// kdtree.go:94-146 on the implicit tree (pcgx_internal.h: node b's children are 2b and 2b + 1, its depth floor(log2 b),
// its size a closed form of b: node_size), as the in-order walk  visit(near) ; test node ; visit(far)  with one running
// best (knn_walk.h says why that is the reference's recursion) -- and no stack: a node's parent is a shift of its index,
// and which side of the parent it hangs on follows from the parent's split value, so "where did I come from" is
// recomputed instead of stored.  The kernel's way for a query that is not finite only (see the head of this file):
// every value out of device memory, one lane on its own.
template <bool kMinDist>
__device__ __forceinline__ void small_walk(const float4 *__restrict__ nodes, uint32_t m1, float qx, float qy, float qz, float max_dist_sq,
                                        float min_dist_sq, float4 &best, float &best_d) {
  best = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(-1));
  best_d = max_dist_sq;  // nothing in range: {-1, maxRange^2} (kdtree.go:100-103)
  uint32_t b = 1u;       // the node the walk stands at
  int d = 0;             // its depth
  bool desc = true;      // descending to a leaf / unwinding from b, whose sub-tree is done
  for (uint32_t guard = 2u * m1 + 8u; guard != 0u; --guard) {
    // ---- to the next point to evaluate: a leaf (desc), or a pivot the plane test lets through (unwinding)
    uint32_t near_of_p = 0u, szp = 0u;
    if (desc) {  // searchLeafNode (kdtree.go:199-222)
      for (;;) {
        const uint32_t sz = node_size(b, d, m1);
        if (sz <= 1u) break;
        const float qv = sel3(d % 3, qx, qy, qz);
        b = sz == 2u ? 2u * b : (node_comp(nodes, b, d % 3) > qv ? 2u * b : 2u * b + 1u);  // one child: that one; else pivot > p -> child0 (:216)
        d++;
      }
    } else {
      bool at_pivot = false;
      while (b != 1u) {
        const uint32_t p = b >> 1;
        const int dp = d - 1;
        szp = node_size(p, dp, m1);
        const float qv = sel3(dp % 3, qx, qy, qz), sv = node_comp(nodes, p, dp % 3);
        near_of_p = szp == 2u ? 2u * p : (sv > qv ? 2u * p : 2u * p + 1u);  // the side the descent took at p
        const bool from_near = b == near_of_p;
        b = p;
        d = dp;
        if (!from_near) continue;  // p's far side is done: so is p
        const float fp = qv - sv;  // p[dim] - pivot[dim]
        if (fp * fp > best_d) continue;  // kdtree.go:111-115: neither the pivot nor the far side
        at_pivot = true;
        break;
      }
      if (!at_pivot) return;  // the root's sub-tree is done
    }
    // ---- its record, its distance (mat/vec3.go:18-20,38-40)
    const float4 nd = node_at(nodes, b);
    const float dx = nd.x - qx, dy = nd.y - qy, dz = nd.z - qz;
    const float dd = (dx * dx + dy * dy) + dz * dz;
    if (desc) {
      if (!(dd > best_d)) {  // a leaf replaces unless strictly farther (kdtree.go:100-103,138-139)
        best = nd;
        best_d = dd;
      }
      if (kMinDist && best_d < min_dist_sq) return;  // :104-106,140-142
      desc = false;
    } else {
      if (dd < best_d) {  // a pivot: strictly nearer only (kdtree.go:116-123)
        best = nd;
        best_d = dd;
        if (kMinDist && best_d < min_dist_sq) return;
      }
      if (szp != 2u) {  // the far side, with the running best as its bound (:124-137); no other child: on upwards
        b = near_of_p ^ 1u;
        d = d + 1;
        desc = true;
      }
    }
  }
}

// ---- the search over ALL distances (the head of this file) --------------------------------------------------------
constexpr int kSmallBand = 4;  // levels of a chunk: 15 nodes
constexpr int kSmallWaves = kSmallBlock / 64;
// A lane's running minima.  "DistSq" here is 0 for every point under the MinDistSq cut and the point's DistSq otherwise:
// the smallest (DistSq, key) is then the FIRST point under the cut where there is one, and the first point at the
// smallest distance where there is none -- one minimum for both rules.  A point is named by its chunk and its place in
// it (16 * chunk + T; 0: none).
struct SmallAcc {
  unsigned long long fd;  // (DistSq bits, key): see above
  uint32_t fd_id;
  unsigned long long ld;  // (DistSq bits, ~key) over LEAVES: the last leaf at the smallest leaf distance
  uint32_t ld_id;
  __device__ __forceinline__ void clear() {
    fd = ~0ull;
    fd_id = 0u;
    ld = ~0ull;
    ld_id = 0u;
  }
  __device__ __forceinline__ void meet(unsigned long long ofd, uint32_t ofd_id, unsigned long long old_, uint32_t old_id) {
    if (ofd < fd) {
      fd = ofd;
      fd_id = ofd_id;
    }
    if (old_ < ld) {
      ld = old_;
      ld_id = old_id;
    }
  }
};

// A chunk's records arrive by ONE load instruction: lane T - 1 asks for the chunk's node T (T = 1 ... 15: the chunk's own
// BFS numbering, node T's children 2T and 2T + 1), lane 16 + j for the chunk root's ancestor at level j; the wave then
// reads them lane by lane (v_readlane: the values are the same for all lanes, they live in scalar registers) -- and the
// next chunk's load is under way while this one's distances are evaluated.  (One dependent load per node, 15 of them
// in a row and the ancestors' in front: 10 us of latency a chunk for 0.6 us of arithmetic.)
__device__ __forceinline__ void small_chunk_root(int c, int top_levels, uint32_t &r, int &lr) {
  r = 1u;
  lr = 0;
  if (c > 0) {
    int rest = c - 1;
    lr = top_levels;
    while (rest >= (1 << lr)) {
      rest -= 1 << lr;
      lr += kSmallBand;
    }
    r = (1u << lr) + (uint32_t)rest;
  }
}
struct SmallChunk {  // a chunk in a wave's registers: per lane its node's (or ancestor's) record; per wave the rest
  float4 rec;
  uint32_t r;
  int lr;
};
__device__ __forceinline__ void small_chunk_fetch(const float4 *__restrict__ nodes, int c, int top_levels, int lane, SmallChunk &C) {
  uint32_t r;
  int lr;
  small_chunk_root(c, top_levels, r, lr);
  C.r = __builtin_amdgcn_readfirstlane(r);
  C.lr = __builtin_amdgcn_readfirstlane(lr);
  uint32_t idx = 1u;
  if (lane < 15) {
    const uint32_t t = (uint32_t)lane + 1u;
    const int k = 31 - __clz((int)t);
    idx = (C.r << k) | (t - (1u << k));
  } else if (lane >= 16 && lane - 16 < C.lr) {
    idx = C.r >> (C.lr - (lane - 16));
  }
  C.rec = node_at(nodes, idx);  // (every slot below 2^depth is memory of the tree's; a slot without a node: never looked at)
}
__device__ __forceinline__ float lane_value(float v, int src_lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src_lane));
}

// What the wave knows of a chunk before it goes through its nodes.  The tree is balanced (kdtree.go:357-369, node_size):
// every node above the last two levels has both children; a node of the last level but one has 1, 2 or 3 points (a
// leaf, child0 only, both); the last level's nodes are leaves, and only there are BFS slots without a node.  So:
//   * a slot without a node gets a NaN coordinate by its lane: its DistSq is NaN, whose bits are above every number's
//     and which is under no cut -- it never wins anything, and the code has no "is there such a node" in it;
//   * a node of two points gets +inf for a split value: "pivot > p" (:216) is then always true, child0 is the near one
//     -- the one child it has (:206-213);
//   * leaves are looked for in the LAST band's two last levels only (kLast), the last but one by a bit per node.
struct SmallChunkInfo {
  uint32_t leaf_m;            // bit T - 1: node T is a leaf (kLast)
  float sv;                   // per lane: the lane's node's split value
  float qa[kSmallBand];       // the query's coordinate along the axis of each of the chunk's levels
  uint32_t mark[kSmallBand];  // 1 << (the level's place in the key)
  uint32_t id0;               // 16 * chunk
};

// the chunk's node T (the same for every lane of the wave) and, K levels deep, its sub-tree -- straight-line code.
// key_above: the lane's digits of the levels above the node.  kNodes: the nodes' distances and the minima (else only
// what follows).  kBelow (a band with bands below it): `beyond` -- the plane test's bound (small_chunk_go) -- carried
// down to the chunk's last level, and for each of the chunks that hang there a bit in `live` unless no lane has any use
// for its sub-tree (small_chunk_go's test, with the best as it is now).
template <int K, int KC, int T, bool kMinDist, bool kLast, bool kNodes, bool kBelow>
__device__ __forceinline__ void small_chunk(const float4 &rec, const SmallChunkInfo &I, uint32_t key_above, float beyond, float qx, float qy,
                                            float qz, float min_dist_sq, bool active, uint32_t best_bits, SmallAcc &A, uint32_t &live) {
  constexpr int k = T >= 8 ? 3 : (T >= 4 ? 2 : (T >= 2 ? 1 : 0));
  if (kNodes) {
    const float nx = lane_value(rec.x, T - 1), ny = lane_value(rec.y, T - 1), nz = lane_value(rec.z, T - 1);
    const float dx = nx - qx, dy = ny - qy, dz = nz - qz;  // mat/vec3.go:18-20 (pivot.Sub(p)), :38-40 (NormSq)
    float dd = (dx * dx + dy * dy) + dz * dz;
    if (kMinDist) dd = dd < min_dist_sq ? 0.0f : dd;  // (SmallAcc)
    const uint32_t key = key_above | I.mark[k];
    const uint32_t ddb = __float_as_uint(dd);  // (a sum of squares: never negative, the bits order like the values; NaN: above all)
    const uint32_t id = I.id0 + (uint32_t)T;
    const unsigned long long v = ((unsigned long long)ddb << 32) | key;
    if (v < A.fd) {
      A.fd = v;
      A.fd_id = id;
    }
    if (kLast && K <= 2) {  // the tree's last two levels: leaves (searchLeafNode ends at nodes without children only, kdtree.go:204-205)
      const bool leaf = K == 1 || ((I.leaf_m >> (T - 1)) & 1u) != 0u;  // (uniform)
      const unsigned long long w = leaf ? ((unsigned long long)ddb << 32) | (~key) : ~0ull;
      if (w < A.ld) {
        A.ld = w;
        A.ld_id = id;
      }
    }
  }
  if constexpr (K > 1 || kBelow) {
    // searchLeafNode: pivotVal > val -> child0 (:216-220); child0 near: child1's sub-tree is the far one, and the other way round
    const float sv = lane_value(I.sv, T - 1);
    const bool near0 = sv > I.qa[k];
    const uint32_t far = I.mark[k] << 1;
    const uint32_t key0 = kNodes ? key_above | (near0 ? 0u : far) : 0u;
    float beyond0 = beyond, beyond1 = beyond;
    if (kBelow) {
      const float fp = I.qa[k] - sv;  // p[dim] - pivot[dim] (:111)
      const float fp2 = fp * fp;
      const float deeper = fp2 > beyond ? fp2 : beyond;
      beyond0 = near0 ? beyond : deeper;
      beyond1 = near0 ? deeper : beyond;
    }
    if constexpr (K > 1) {
      small_chunk<K - 1, KC, 2 * T, kMinDist, kLast, kNodes, kBelow>(rec, I, key0, beyond0, qx, qy, qz, min_dist_sq, active, best_bits, A, live);
      small_chunk<K - 1, KC, 2 * T + 1, kMinDist, kLast, kNodes, kBelow>(rec, I, key0 ^ far, beyond1, qx, qy, qz, min_dist_sq, active, best_bits, A, live);
    } else {
      constexpr int t0 = 2 * (T - (1 << (KC - 1)));  // the chunk below node T's child0, among the chunks below this one
      const bool cut_found = kMinDist && best_bits == 0u;
      const float best = __uint_as_float(best_bits);
      const bool no_use0 = !active || ((!kMinDist || !(beyond0 < min_dist_sq)) && (cut_found || beyond0 > best));
      const bool no_use1 = !active || ((!kMinDist || !(beyond1 < min_dist_sq)) && (cut_found || beyond1 > best));
      if (__ballot(no_use0) != ~0ull) live |= 1u << t0;
      if (__ballot(no_use1) != ~0ull) live |= 2u << t0;
    }
  }
}

// one chunk: the wave's 64 queries against its (up to) fifteen nodes.  `active`: the lane has a query.
// Returns a bit for each of the chunks that hang below this one (kBelow; up to sixteen) unless the chunk's SUB-TREE is of
// no use to any lane; mine: this workgroup goes through the chunk's nodes (else it only wants to know where to look below it).
template <int K, bool kMinDist, bool kLast, bool kBelow>
__device__ __forceinline__ uint32_t small_chunk_go(const SmallChunk &C, int c, int lane, uint32_t m1, int D, float qx, float qy, float qz,
                                                   float min_dist_sq, bool active, bool mine, uint32_t seed_bits, uint32_t *s_best, SmallAcc &A) {
  const uint32_t r = C.r;
  const int lr = C.lr;
  SmallChunkInfo I;
  float4 rec = C.rec;
  // ---- by the lanes: sizes, split values
  {
    uint32_t b = 0u;
    int depth = 0;
    if (lane < 15) {
      const uint32_t t = (uint32_t)lane + 1u;
      const int k = 31 - __clz((int)t);
      b = (r << k) | (t - (1u << k));
      depth = lr + k;
    } else if (lane >= 16 && lane - 16 < lr) {
      depth = lane - 16;
      b = r >> (lr - depth);
    }
    const int ax = depth % 3;
    const float own = ax == 0 ? rec.x : (ax == 1 ? rec.y : rec.z);
    I.sv = own;
    I.leaf_m = 0u;
    if (kLast) {
      const uint32_t sz = b != 0u ? node_size(b, depth, m1) : 0u;
      const bool exists = sz != 0u && sz != 0xFFFFFFFFu;
      I.leaf_m = (uint32_t)__ballot(lane < 15 && sz == 1u);
      if (sz == 2u) I.sv = __builtin_inff();
      if (lane < 15 && !exists) rec.x = __builtin_nanf("");
    }
  }
  // ---- the lane's digits of the levels above the chunk's root: the reference's comparison at every ancestor (:216);
  // and what its plane test (:111-115) would say of the whole sub-tree: every point of it lies beyond the planes of the
  // ancestors on whose FAR side the root hangs, its DistSq is at least the largest of those plane distances squared
  uint32_t above = 0u;
  float beyond = 0.0f;
#pragma unroll
  for (int j = 0; j < 15; j++) {
    if (j >= lr) break;  // uniform
    const float sv = lane_value(I.sv, 16 + j);
    const float qv = j % 3 == 0 ? qx : (j % 3 == 1 ? qy : qz);
    const bool to1 = ((r >> (lr - 1 - j)) & 1u) != 0u;  // the root's side of the ancestor (uniform)
    const bool near0 = sv > qv;                           // the query's
    const bool far = near0 == to1;
    above |= far ? (2u << (2 * (D - 1 - j))) : 0u;
    const float fp = qv - sv;  // p[dim] - pivot[dim] (:111)
    const float fp2 = fp * fp;
    beyond = far && fp2 > beyond ? fp2 : beyond;
  }
  // Nothing in the sub-tree can matter to a lane whose best is already nearer than `beyond` -- STRICTLY (:113 skips on >;
  // a point at the same DistSq behind the best changes the answer only as a leaf at the smallest distance, which is not
  // beyond it) -- and that has nothing under the MinDistSq cut to find there.  All lanes: the chunk is not gone through.
  // (the best: this wave's, or another wave's of the workgroup -- they have the same 64 targets and meet in the end)
  {
    const uint32_t own0 = (uint32_t)(A.fd >> 32), theirs0 = s_best[lane];
    const uint32_t best01 = own0 < theirs0 ? own0 : theirs0;
    const uint32_t best_bits = best01 < seed_bits ? best01 : seed_bits;
    const bool cut_found = kMinDist && best_bits == 0u;
    const bool no_use = !active || ((!kMinDist || !(beyond < min_dist_sq)) && (cut_found || beyond > __uint_as_float(best_bits)));
    PCGX_SMALL_COUNT(0);
    if (__ballot(no_use) == ~0ull) {
      PCGX_SMALL_COUNT(1);
      return 0u;
    }
  }
  if (!mine && !kBelow) return 0u;
  const int lr3 = lr % 3;
#pragma unroll
  for (int k = 0; k < kSmallBand; k++) {
    const int ax = (lr3 + k) % 3;
    I.qa[k] = ax == 0 ? qx : (ax == 1 ? qy : qz);
    I.mark[k] = 1u << ((2 * (D - 1 - lr - k)) & 31);  // (levels below the tree's last: no nodes, never used)
  }
  I.id0 = (uint32_t)c << 4;
  uint32_t live = 0u;
  const uint32_t own = (uint32_t)(A.fd >> 32), theirs = s_best[lane];
  const uint32_t best2 = own < theirs ? own : theirs;
  const uint32_t best_bits = best2 < seed_bits ? best2 : seed_bits;
  if (mine) {
    PCGX_SMALL_COUNT(2);
    small_chunk<K, K, 1, kMinDist, kLast, true, kBelow>(rec, I, above, beyond, qx, qy, qz, min_dist_sq, active, best_bits, A, live);
    (void)atomicMin(&s_best[lane], (uint32_t)(A.fd >> 32));
  } else if (kBelow) {
    small_chunk<K, K, 1, kMinDist, kLast, false, kBelow>(rec, I, above, beyond, qx, qy, qz, min_dist_sq, active, best_bits, A, live);
  }
  return live;
}

// ---- the sums of LONG rows: tiles of 2048 terms summarised in parallel (strict_sum.h) ------------------------------------
// One wave a sum is 5 ns a term: 85 us of a 140 us iteration at 16384 targets.  Beyond a row's first tile the tiles are
// summarised by waves of their own, all at once -- strict_sum.h: inside a binade a run of float32 additions moves the
// state's mantissa by an amount that depends only on its parity class, as long as no step leaves the binade; a wave
// finds that translation and the interval of states it holds for by running the hardware's adds from a GUESS of the
// tile's first state (the float64 sum of everything in front of it) -- and the row's wave walks through the records:
// a record that covers the exact state is applied (proven equal to the additions one by one), one that does not is
// replaced by the additions one by one.  ss_host_model restated for a wave a tile, the general path's kernels
// (strict.hip) without their batching: here a row has eight tiles at most.
namespace mini {
using namespace ss;
static_assert(sizeof(TileRec) == 64, "sixteen words a record");

template <class T, class F>
__device__ __forceinline__ T ordered_total(T v, int lane, F compose) {  // compose(lane 0's, lane 1's, ... lane 63's), in every lane
  constexpr int kInts = (int)(sizeof(T) / 4);
  static_assert(sizeof(T) % 4 == 0, "a record of 32-bit words");
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    T o;
    int *po = reinterpret_cast<int *>(&o);
    const int *pv = reinterpret_cast<const int *>(&v);
#pragma unroll
    for (int k = 0; k < kInts; k++) po[k] = __shfl_up(pv[k], off);
    if (lane >= off) v = compose(o, v);
  }
  T out;
  int *pout = reinterpret_cast<int *>(&out);
  const int *pv = reinterpret_cast<const int *>(&v);
#pragma unroll
  for (int k = 0; k < kInts; k++) pout[k] = __shfl(pv[k], 63);
  return out;
}
__device__ __forceinline__ double wave_excl_scan(double v, int lane, double &total) {
  double x = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const double o = __shfl_up(x, off);
    if (lane >= off) x += o;
  }
  total = __shfl(x, 63);
  const double before = __shfl_up(x, 1);
  return lane == 0 ? 0.0 : before;
}
__device__ __forceinline__ uint32_t wave_umin(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = umin(v, (uint32_t)__shfl_xor((int)v, o));
  return v;
}
__device__ __forceinline__ uint32_t wave_umax(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = umax(v, (uint32_t)__shfl_xor((int)v, o));
  return v;
}

// A tile in a wave's registers: t[kLeaf] the lane's leaf (terms 32 * lane ... of the tile), and what the lane's guess chain
// gave.  base: the guess of the state in front of the tile (ss_host_model, tile_guesses: the float64 prefix of the leaves'
// sums, refined once by the prefix of the rounding errors the chains make when started from those first guesses).
struct Leaf {
  uint32_t guess;
  ChainRange cr;
};
__device__ __forceinline__ Leaf tile_guesses(const float (&t)[kLeaf], double base, double lsum, int lane) {
  Leaf L;
  double tot;
  const double pre = wave_excl_scan(lsum, lane, tot);
  L.guess = f2u((float)(base + pre));
  L.cr = guess_chain(t, L.guess);
  const double err = ((double)u2f(L.cr.end) - (double)u2f(L.guess)) - lsum;
  double etot;
  const double e = wave_excl_scan(err, lane, etot);
  const uint32_t g2 = f2u((float)(base + pre + e));
  if (__ballot(g2 != L.guess) != 0ull) {
    L.guess = g2;
    L.cr = guess_chain(t, L.guess);
  }
  return L;
}
// the tile's window (-1: none: its states change sign or run through three binades) and whether it stays in ONE binade (E)
__device__ __forceinline__ int32_t tile_window(const Leaf &L, int lane, bool &one_binade, uint32_t &E, uint32_t &sign) {
  const uint32_t mn = wave_umin(L.cr.mn), mx = wave_umax(L.cr.mx);
  const uint32_t sg_or = __ballot(L.cr.sg_or != 0u) != 0ull ? 1u : 0u, sg_and = __ballot(L.cr.sg_and == 0u) != 0ull ? 0u : 1u;
  const uint32_t guess0 = (uint32_t)__shfl((int)L.guess, 0);
  const int32_t key = sg_or == sg_and ? choose_window(mn, mx, sg_or, guess0 & 0x7fffffffu) : -1;
  one_binade = key >= 0 && (mn >> 23) == (mx >> 23);
  E = mn >> 23;
  sign = sg_or;
  return key;
}
// the record of one tile (ss_host_model: "strict_sum_kernel: one record per tile", a leaf a lane)
__device__ __forceinline__ TileRec tile_record(const float (&t)[kLeaf], const Leaf &L, int lane) {
  bool one_binade;
  uint32_t E, sign;
  TileRec T;
  T.key = tile_window(L, lane, one_binade, E, sign);
  T.in = (uint32_t)__shfl((int)L.guess, 0);
  T.out = (uint32_t)__shfl((int)L.cr.end, 63);
  const uint32_t next_guess = (uint32_t)__shfl_down((int)L.guess, 1);
  T.cons = __ballot(lane < 63 && next_guess != L.cr.end) == 0ull ? 1 : 0;
  T.s = summary_identity();
  if (one_binade) {  // parity summaries
    const Par mine = leaf_parity_summary(t, L.guess, L.cr, E, sign);
    const Par acc = ordered_total(mine, lane, [](const Par &X, const Par &Y) { return par_compose(X, Y); });
    T.s = par_expand(acc, E, T.key);
  } else if (T.key >= 0) {
    const bool leaf_one = (L.cr.mn >> 23) == (L.cr.mx >> 23);
    const Summary mine = leaf_one ? leaf_summary_binade(t, L.guess, T.key) : leaf_summary_general(t, L.guess, T.key);
    T.s = ordered_total(mine, lane, [](const Summary &X, const Summary &Y) { return compose(X, Y); });
  }
  return T;
}
}  // namespace mini

// 64 terms added to s one after the other (term k in lane k of `term`; evaluator.go:122-145): v_readlane + v_add_f32 with
// the term in a scalar register, 12 cycles = 5 ns a term whatever the order of the two (tools/micro/dpp_chain.cpp: the
// v_readlane eight terms ahead of its add, as here: 12.1; the compiler's order: 11.8 there, 19 in this kernel between
// the batches' bookkeeping; the running sum hopping from lane to lane, v_add_f32_dpp wave_shr:1: 15.7).  As one
// statement so that a chain's speed does not depend on what the compiler weaves into it.
__device__ __forceinline__ float chain64(float s, float term) {
  int t0, t1, t2, t3, t4, t5, t6, t7;
  asm volatile(
      "v_readlane_b32 %1, %9, 0\n\t"
      "v_readlane_b32 %2, %9, 1\n\t"
      "v_readlane_b32 %3, %9, 2\n\t"
      "v_readlane_b32 %4, %9, 3\n\t"
      "v_readlane_b32 %5, %9, 4\n\t"
      "v_readlane_b32 %6, %9, 5\n\t"
      "v_readlane_b32 %7, %9, 6\n\t"
      "v_readlane_b32 %8, %9, 7\n\t"
      "v_add_f32 %0, %1, %0\n\t"
      "v_readlane_b32 %1, %9, 8\n\t"
      "v_add_f32 %0, %2, %0\n\t"
      "v_readlane_b32 %2, %9, 9\n\t"
      "v_add_f32 %0, %3, %0\n\t"
      "v_readlane_b32 %3, %9, 10\n\t"
      "v_add_f32 %0, %4, %0\n\t"
      "v_readlane_b32 %4, %9, 11\n\t"
      "v_add_f32 %0, %5, %0\n\t"
      "v_readlane_b32 %5, %9, 12\n\t"
      "v_add_f32 %0, %6, %0\n\t"
      "v_readlane_b32 %6, %9, 13\n\t"
      "v_add_f32 %0, %7, %0\n\t"
      "v_readlane_b32 %7, %9, 14\n\t"
      "v_add_f32 %0, %8, %0\n\t"
      "v_readlane_b32 %8, %9, 15\n\t"
      "v_add_f32 %0, %1, %0\n\t"
      "v_readlane_b32 %1, %9, 16\n\t"
      "v_add_f32 %0, %2, %0\n\t"
      "v_readlane_b32 %2, %9, 17\n\t"
      "v_add_f32 %0, %3, %0\n\t"
      "v_readlane_b32 %3, %9, 18\n\t"
      "v_add_f32 %0, %4, %0\n\t"
      "v_readlane_b32 %4, %9, 19\n\t"
      "v_add_f32 %0, %5, %0\n\t"
      "v_readlane_b32 %5, %9, 20\n\t"
      "v_add_f32 %0, %6, %0\n\t"
      "v_readlane_b32 %6, %9, 21\n\t"
      "v_add_f32 %0, %7, %0\n\t"
      "v_readlane_b32 %7, %9, 22\n\t"
      "v_add_f32 %0, %8, %0\n\t"
      "v_readlane_b32 %8, %9, 23\n\t"
      "v_add_f32 %0, %1, %0\n\t"
      "v_readlane_b32 %1, %9, 24\n\t"
      "v_add_f32 %0, %2, %0\n\t"
      "v_readlane_b32 %2, %9, 25\n\t"
      "v_add_f32 %0, %3, %0\n\t"
      "v_readlane_b32 %3, %9, 26\n\t"
      "v_add_f32 %0, %4, %0\n\t"
      "v_readlane_b32 %4, %9, 27\n\t"
      "v_add_f32 %0, %5, %0\n\t"
      "v_readlane_b32 %5, %9, 28\n\t"
      "v_add_f32 %0, %6, %0\n\t"
      "v_readlane_b32 %6, %9, 29\n\t"
      "v_add_f32 %0, %7, %0\n\t"
      "v_readlane_b32 %7, %9, 30\n\t"
      "v_add_f32 %0, %8, %0\n\t"
      "v_readlane_b32 %8, %9, 31\n\t"
      "v_add_f32 %0, %1, %0\n\t"
      "v_readlane_b32 %1, %9, 32\n\t"
      "v_add_f32 %0, %2, %0\n\t"
      "v_readlane_b32 %2, %9, 33\n\t"
      "v_add_f32 %0, %3, %0\n\t"
      "v_readlane_b32 %3, %9, 34\n\t"
      "v_add_f32 %0, %4, %0\n\t"
      "v_readlane_b32 %4, %9, 35\n\t"
      "v_add_f32 %0, %5, %0\n\t"
      "v_readlane_b32 %5, %9, 36\n\t"
      "v_add_f32 %0, %6, %0\n\t"
      "v_readlane_b32 %6, %9, 37\n\t"
      "v_add_f32 %0, %7, %0\n\t"
      "v_readlane_b32 %7, %9, 38\n\t"
      "v_add_f32 %0, %8, %0\n\t"
      "v_readlane_b32 %8, %9, 39\n\t"
      "v_add_f32 %0, %1, %0\n\t"
      "v_readlane_b32 %1, %9, 40\n\t"
      "v_add_f32 %0, %2, %0\n\t"
      "v_readlane_b32 %2, %9, 41\n\t"
      "v_add_f32 %0, %3, %0\n\t"
      "v_readlane_b32 %3, %9, 42\n\t"
      "v_add_f32 %0, %4, %0\n\t"
      "v_readlane_b32 %4, %9, 43\n\t"
      "v_add_f32 %0, %5, %0\n\t"
      "v_readlane_b32 %5, %9, 44\n\t"
      "v_add_f32 %0, %6, %0\n\t"
      "v_readlane_b32 %6, %9, 45\n\t"
      "v_add_f32 %0, %7, %0\n\t"
      "v_readlane_b32 %7, %9, 46\n\t"
      "v_add_f32 %0, %8, %0\n\t"
      "v_readlane_b32 %8, %9, 47\n\t"
      "v_add_f32 %0, %1, %0\n\t"
      "v_readlane_b32 %1, %9, 48\n\t"
      "v_add_f32 %0, %2, %0\n\t"
      "v_readlane_b32 %2, %9, 49\n\t"
      "v_add_f32 %0, %3, %0\n\t"
      "v_readlane_b32 %3, %9, 50\n\t"
      "v_add_f32 %0, %4, %0\n\t"
      "v_readlane_b32 %4, %9, 51\n\t"
      "v_add_f32 %0, %5, %0\n\t"
      "v_readlane_b32 %5, %9, 52\n\t"
      "v_add_f32 %0, %6, %0\n\t"
      "v_readlane_b32 %6, %9, 53\n\t"
      "v_add_f32 %0, %7, %0\n\t"
      "v_readlane_b32 %7, %9, 54\n\t"
      "v_add_f32 %0, %8, %0\n\t"
      "v_readlane_b32 %8, %9, 55\n\t"
      "v_add_f32 %0, %1, %0\n\t"
      "v_readlane_b32 %1, %9, 56\n\t"
      "v_add_f32 %0, %2, %0\n\t"
      "v_readlane_b32 %2, %9, 57\n\t"
      "v_add_f32 %0, %3, %0\n\t"
      "v_readlane_b32 %3, %9, 58\n\t"
      "v_add_f32 %0, %4, %0\n\t"
      "v_readlane_b32 %4, %9, 59\n\t"
      "v_add_f32 %0, %5, %0\n\t"
      "v_readlane_b32 %5, %9, 60\n\t"
      "v_add_f32 %0, %6, %0\n\t"
      "v_readlane_b32 %6, %9, 61\n\t"
      "v_add_f32 %0, %7, %0\n\t"
      "v_readlane_b32 %7, %9, 62\n\t"
      "v_add_f32 %0, %8, %0\n\t"
      "v_readlane_b32 %8, %9, 63\n\t"
      "v_add_f32 %0, %1, %0\n\t"
      "v_add_f32 %0, %2, %0\n\t"
      "v_add_f32 %0, %3, %0\n\t"
      "v_add_f32 %0, %4, %0\n\t"
      "v_add_f32 %0, %5, %0\n\t"
      "v_add_f32 %0, %6, %0\n\t"
      "v_add_f32 %0, %7, %0\n\t"
      "v_add_f32 %0, %8, %0\n\t"
      : "+v"(s), "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3), "=&s"(t4), "=&s"(t5), "=&s"(t6), "=&s"(t7)
      : "v"(term));
  return s;
}

// One launch = `iters` iterations of Fit's loop (icp.go:48-65) from the state in *state; see the head of this file.
// tx / ty / tz: the ORIGINAL target in the caller's order; terms: [kStrictRows][ntp] tagged words, ntp = nt rounded up to
// 64; valid: [ntp / 64] matched-target bits (for whoever looks after the launch); sums10: the session's sums (device
// memory); scratch: small_fit_sync_bytes(), its first small_fit_zero_bytes() zero at the first launch; launch_no: a
// number no other launch on this scratch block has had for 2^20 launches.  gridDim.x = (ntp / 64) * P.
template <bool kMinDist>
__global__ __launch_bounds__(kSmallBlock) void icp_small_fit_kernel(TreeView tv, const float *__restrict__ tx,
                                                                    const float *__restrict__ ty, const float *__restrict__ tz,
                                                                    int64_t nt, int64_t ntp, IcpState *__restrict__ state,
                                                                    IcpKernelParams kp, unsigned long long *__restrict__ terms,
                                                                    unsigned long long *__restrict__ valid,
                                                                    double *__restrict__ sums10, char *__restrict__ scratch,
                                                                    uint32_t launch_no, int P, int mode, int iters, const int32_t *__restrict__ perm,
                                                                    volatile uint32_t *__restrict__ mailbox,
                                                                    uint32_t mailbox_seq) {
  const bool hier = (mode & 1) != 0, seeded = (mode & 2) != 0, queued = (mode & 4) != 0, queued_flat = (mode & 8) != 0;
  const int seeds_per_wg = mode >> 8;  // (the queue: the first band with this many chunks a workgroup is on it from the start)
  __shared__ uint32_t s_part[kPartWords][kSmallWaves][64];
  __shared__ int s_exited;         // the workgroup's waves that have left the loop
  __shared__ uint32_t s_best[64];  // the group's targets' best DistSq bits so far, over all waves of the workgroup (0: under the cut)
  __shared__ int s_next[8];        // the workgroup's next chunk (flat: its n-th is chunk n * P + p), band by band
  __shared__ int s_cnt[8];         // band by band: the chunks on a band's list
  __shared__ uint16_t s_list[2][4096];  // ... a chunk by its root's number among the band's
  SmallSync *sy = reinterpret_cast<SmallSync *>(scratch);
  unsigned long long *pose_w = reinterpret_cast<unsigned long long *>(scratch + kSmallPoseAt);
  unsigned long long *sums_w = reinterpret_cast<unsigned long long *>(scratch + kSmallSumsAt);
  unsigned long long *counts_w = reinterpret_cast<unsigned long long *>(scratch + kSmallCountsAt);
  unsigned long long *part_w = reinterpret_cast<unsigned long long *>(scratch + kSmallPartAt);
  uint32_t *partner_of = reinterpret_cast<uint32_t *>(scratch + kSmallPartnersAt);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned int G = gridDim.x;
  const uint32_t m1 = (uint32_t)tv.n + 1u;
  const int g = (int)blockIdx.x / P, p = (int)blockIdx.x % P;  // the group of 64 targets, this workgroup's share of its chunks
  const int Q = (int)(ntp >> 6);
  const int64_t i = (int64_t)g * 64 + lane;                     // the lane's target (every wave of the group's workgroups: the same 64) ...
  // ... by its place in the session's order (small_order_kernel); the terms go to the CALLER's place (padding: its own)
  float x0 = 0.0f, y0 = 0.0f, z0 = 0.0f;
  if (i < nt) {
    x0 = tx[i];
    y0 = ty[i];
    z0 = tz[i];
  }
  // the chunks: the tree's levels in bands of kSmallBand from the BOTTOM (the top band takes what is left), a chunk a
  // band's sub-tree: chunk 0 the root's, then the bands' roots level by level
  const int D = tv.depth;
  const int top_levels = D % kSmallBand == 0 ? kSmallBand : D % kSmallBand;
  int nchunks = 1;
  for (int lr = top_levels; lr < D; lr += kSmallBand) nchunks += 1 << lr;
  const int nrows = kp.weight_fn == PCGX_WEIGHT_ONE ? kStrictRows - 1 : kStrictRows;
  // the sums' workers: every wave but a workgroup's first (which decides and, in workgroup 0, updates), workgroup by
  // workgroup first -- nine workgroups or more: a row a workgroup
  const int worker = wave == 0 ? -1 : (wave - 1) * (int)G + (int)blockIdx.x, nworkers = (kSmallWaves - 1) * (int)G;
  // (the queue of chunks, below: what is on it from the start)
  int seed_lr = 0, seed_total = 0;  // the deepest band on the queue from the start: its roots' level; the chunks down to it
  {
    int lr = 0, cnt = 1;
    for (;;) {
      seed_total += cnt;
      seed_lr = lr;
      const int kb = lr == 0 ? top_levels : kSmallBand;
      if (cnt >= seeds_per_wg * P || lr + kb >= D || queued_flat) break;
      lr += kb;
      cnt = 1 << lr;
    }
    if (queued_flat) {  // (every chunk from the start: nothing is put on)
      seed_total = nchunks;
      seed_lr = D;
    }
  }
  const int n_seeds = seed_total > p ? (seed_total - p + P - 1) / P : 0;
  auto reset_queue = [&]() {  // (by the workgroup's first wave)
    uint4 *q = reinterpret_cast<uint4 *>(&s_list[0][0]);
    if (queued) {
      for (int k = lane; k < 8192 * 2 / 16; k += 64) q[k] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    }
    if (lane < 8) {
      s_next[lane] = 0;
      s_cnt[lane] = queued ? (lane == 0 ? n_seeds : 0) : (lane == 0 ? 1 : 0);
    }
    if (lane == 0 && !queued) s_list[0][0] = 0;
  };
  bool alive = true;
  SmallWait wait;
  if (wave == 0) {
    s_best[lane] = 0xFFFFFFFFu;
    reset_queue();
    if (lane == 0) s_exited = 0;
  }
  __syncthreads();
#if defined(PCGX_STAMPS)
  if (threadIdx.x == 0 && blockIdx.x == 0) g_small_iter_t[0] = wall_clock64();
#endif
  for (int it = 0; it < iters; it++) {
#if defined(PCGX_STAMPS)
    if (threadIdx.x == 0 && blockIdx.x == 0 && it < 60) g_small_iter_t[1 + it] = wall_clock64();
#endif
    const uint32_t tag = (launch_no << kSmallTagIterBits) | (uint32_t)(it + 1);
    // ---- the loop state: the launch's first iteration from *state (the launch before, the host), the others from the
    // updater's words
    PCGX_STAMP_IF(it == PCGX_STAMP_ITER, small_fit, 8, blockIdx.x, 0);
    float m[16];
    int upd_iter, done;
    if (it == 0) {
#pragma unroll
      for (int k = 0; k < 16; k++) m[k] = state->trans[k];
      upd_iter = state->iter;
      done = state->done;
    } else {
      unsigned long long w = 0ull;
      for (;;) {
        w = lane < 18 ? word_in(&pose_w[lane]) : tagged(0u, tag);
        if (__ballot((uint32_t)(w >> 32) == tag) == ~0ull) break;
        if (small_give_up(sy, wait)) {
          alive = false;
          break;
        }
      }
      if (!alive) break;
#pragma unroll
      for (int k = 0; k < 16; k++) m[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane((int)(uint32_t)w, k));
      upd_iter = __builtin_amdgcn_readlane((int)(uint32_t)w, 16);
      done = __builtin_amdgcn_readlane((int)(uint32_t)w, 17);
    }
    if (done) break;  // (the same words for everybody)
    // ---- correspondence (correspondence.go:25-36): the minima of this wave's chunks
    float x = x0, y = y0, z = z0;
    if (upd_iter > 0) mat4_transform(m, x0, y0, z0, x, y, z);  // icp.go:27-30,62-64
    const bool finite = (__float_as_uint(x) & 0x7F800000u) != 0x7F800000u && (__float_as_uint(y) & 0x7F800000u) != 0x7F800000u &&
                        (__float_as_uint(z) & 0x7F800000u) != 0x7F800000u;
    const bool has_query = i < nt && finite;  // (a query that is not finite is walked, below: its minima are not looked at)
    // What the target found last time is still a point of the tree: its distance from where the target is now bounds the
    // nearest one's -- the plane test's bound from the first chunk on instead of from whenever a wave comes by the
    // target's neighbourhood.  (ANY node is a bound: a partner that is an iteration out of date, or 0 for none, costs
    // time, never the answer.  The words are the session's: zero when it is made.)
    uint32_t seed_bits = 0xFFFFFFFFu;
    if (has_query && seeded) {
      const uint32_t pb = __hip_atomic_load(&partner_of[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (pb >= 1u && pb < (1u << D)) {
        const float4 pn = node_at(tv.nodes, pb);
        const float dx = pn.x - x, dy = pn.y - y, dz = pn.z - z;
        const float dd = (dx * dx + dy * dy) + dz * dz;
        seed_bits = kMinDist && dd < kp.min_dist_sq ? 0u : __float_as_uint(dd);
      }
    }
    SmallAcc A;
    A.clear();
    PCGX_STAMP_IF(it == PCGX_STAMP_ITER, small_fit, 8, blockIdx.x, 1);
    {
      // Two sets of registers take turns, the next chunk's records on their way while this one's distances are
      // evaluated.  (The empty asm statements make the compiler wait for a chunk's records where they are first needed
      // and nowhere else: its own waits count loads in order, and a load still pending at a loop's head is waited for at
      // the first use behind the next one's issue -- every fetch synchronous.)
      auto go_flat = [&](const SmallChunk &C, int c) {
        if (c == 0) {  // the root's chunk: the levels the bands leave over (the tree's last band too where it has four levels at most)
          if (top_levels == D) {
            if (top_levels == 1) (void)small_chunk_go<1, kMinDist, true, false>(C, 0, lane, m1, D, x, y, z, kp.min_dist_sq, has_query, true, seed_bits, s_best, A);
            else if (top_levels == 2) (void)small_chunk_go<2, kMinDist, true, false>(C, 0, lane, m1, D, x, y, z, kp.min_dist_sq, has_query, true, seed_bits, s_best, A);
            else if (top_levels == 3) (void)small_chunk_go<3, kMinDist, true, false>(C, 0, lane, m1, D, x, y, z, kp.min_dist_sq, has_query, true, seed_bits, s_best, A);
            else (void)small_chunk_go<kSmallBand, kMinDist, true, false>(C, 0, lane, m1, D, x, y, z, kp.min_dist_sq, has_query, true, seed_bits, s_best, A);
          } else {
            if (top_levels == 1) (void)small_chunk_go<1, kMinDist, false, false>(C, 0, lane, m1, D, x, y, z, kp.min_dist_sq, has_query, true, seed_bits, s_best, A);
            else if (top_levels == 2) (void)small_chunk_go<2, kMinDist, false, false>(C, 0, lane, m1, D, x, y, z, kp.min_dist_sq, has_query, true, seed_bits, s_best, A);
            else if (top_levels == 3) (void)small_chunk_go<3, kMinDist, false, false>(C, 0, lane, m1, D, x, y, z, kp.min_dist_sq, has_query, true, seed_bits, s_best, A);
            else (void)small_chunk_go<kSmallBand, kMinDist, false, false>(C, 0, lane, m1, D, x, y, z, kp.min_dist_sq, has_query, true, seed_bits, s_best, A);
          }
        } else if (C.lr + kSmallBand == D) {
          (void)small_chunk_go<kSmallBand, kMinDist, true, false>(C, c, lane, m1, D, x, y, z, kp.min_dist_sq, has_query, true, seed_bits, s_best, A);
        } else {
          (void)small_chunk_go<kSmallBand, kMinDist, false, false>(C, c, lane, m1, D, x, y, z, kp.min_dist_sq, has_query, true, seed_bits, s_best, A);
        }
      };
      // (band by band: the chunks that hang below chunk c and are worth a look, as bits)
      auto go_bands = [&](const SmallChunk &C, int c, bool mine) -> uint32_t {
        if (c == 0) {
          if (top_levels == D) {  // (no bands below)
            go_flat(C, 0);
            return 0u;
          }
          if (top_levels == 1) return small_chunk_go<1, kMinDist, false, true>(C, 0, lane, m1, D, x, y, z, kp.min_dist_sq, has_query, mine, seed_bits, s_best, A);
          if (top_levels == 2) return small_chunk_go<2, kMinDist, false, true>(C, 0, lane, m1, D, x, y, z, kp.min_dist_sq, has_query, mine, seed_bits, s_best, A);
          if (top_levels == 3) return small_chunk_go<3, kMinDist, false, true>(C, 0, lane, m1, D, x, y, z, kp.min_dist_sq, has_query, mine, seed_bits, s_best, A);
          return small_chunk_go<kSmallBand, kMinDist, false, true>(C, 0, lane, m1, D, x, y, z, kp.min_dist_sq, has_query, mine, seed_bits, s_best, A);
        }
        if (C.lr + kSmallBand == D) return small_chunk_go<kSmallBand, kMinDist, true, false>(C, c, lane, m1, D, x, y, z, kp.min_dist_sq, has_query, mine, seed_bits, s_best, A);
        return small_chunk_go<kSmallBand, kMinDist, false, true>(C, c, lane, m1, D, x, y, z, kp.min_dist_sq, has_query, mine, seed_bits, s_best, A);
      };
      SmallChunk CA, CB;
      CA.rec = CB.rec = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      CA.r = CB.r = 1u;
      CA.lr = CB.lr = 0;
      if (queued) {
        // A QUEUE (LDS) of the workgroup's chunks: the chunks of the tree's upper bands are on it from the start (this
        // workgroup's: every P-th), a chunk that could not be ruled out puts the chunks below it behind them (the bits of
        // small_chunk_go), whoever is free takes the next.  No list a band and no barrier between bands, nothing looked at
        // by two workgroups, nothing looked at below what was ruled out.  An entry is a claim first (a place in the
        // queue) and a chunk once its producer has written it; the queue is done with when every entry taken has been
        // gone through (whoever goes through a chunk counts it AFTER putting its children on) and none is left.
        volatile uint16_t *queue = &s_list[0][0];
        volatile int *q_head = &s_next[0], *q_tail = &s_cnt[0], *q_done = &s_cnt[1];
        auto claim = [&]() -> int {
          int h = 0;
          if (lane == 0) h = atomicAdd(const_cast<int *>(q_head), 1);
          return __builtin_amdgcn_readfirstlane(h);
        };
        // the chunk of claim h: -2 not there yet, -1 there will be none
        auto look = [&](int h) -> int {
          if (h < n_seeds) return p + h * P;
          const int at = h - n_seeds;
          if (at < 8192) {
            const uint16_t v = queue[at];
            if (v != 0xFFFFu) return (int)v;
          }
          const int d = *q_done, t = *q_tail;  // (done first: it only grows, and never past the tail)
          if ((d == t && t <= h) || at >= 8192) return -1;
          return -2;
        };
        auto wait_for = [&](int h) -> int {
          for (;;) {
            const int c = look(h);
            if (c != -2) return c;
            if (small_give_up(sy, wait)) {  // (bounded like every wait of the launch)
              alive = false;
              return -1;
            }
          }
        };
        auto through = [&](const SmallChunk &C, int c) {
          const int kb = C.lr == 0 ? top_levels : kSmallBand;
          const bool below = C.lr >= seed_lr && C.lr + kb < D;  // (the bands above: their children are on the queue from the start)
          uint32_t live = 0u;
          if (below) live = go_bands(C, c, true);
          else go_flat(C, c);
          if (live != 0u) {
            int start = 0;  // the chunks in front of c's band
            for (int lr = 0; lr < C.lr; lr += (lr == 0 ? top_levels : kSmallBand)) start += 1 << lr;
            const int fan = __popc(live), first_below = start + (1 << C.lr) + ((c - start) << kb);
            int at = 0;
            if (lane == 0) at = atomicAdd(const_cast<int *>(q_tail), fan);
            at = __builtin_amdgcn_readfirstlane(at) - n_seeds;
            if (lane < 16 && ((live >> lane) & 1u) != 0u) {
              const int slot = at + __popc(live & ((1u << lane) - 1u));
              if (slot < 8192) queue[slot] = (uint16_t)(first_below + lane);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          }
          if (lane == 0) atomicAdd(const_cast<int *>(q_done), 1);
        };
        int ha = claim();
        int ca = wait_for(ha);
        if (ca >= 0) small_chunk_fetch(tv.nodes, ca, top_levels, lane, CA);
        while (ca >= 0) {
          const int hb = claim();
          int cb = look(hb);
          if (cb >= 0) small_chunk_fetch(tv.nodes, cb, top_levels, lane, CB);
          asm volatile("" : "+v"(CA.rec.x), "+v"(CA.rec.y), "+v"(CA.rec.z), "+v"(CA.rec.w));
          through(CA, ca);
          if (cb == -2) {  // (not there when asked for: perhaps one of the chunk's own children)
            cb = wait_for(hb);
            if (cb >= 0) small_chunk_fetch(tv.nodes, cb, top_levels, lane, CB);
          }
          if (cb < 0) break;
          const int ha2 = claim();
          ca = look(ha2);
          if (ca >= 0) small_chunk_fetch(tv.nodes, ca, top_levels, lane, CA);
          asm volatile("" : "+v"(CB.rec.x), "+v"(CB.rec.y), "+v"(CB.rec.z), "+v"(CB.rec.w));
          through(CB, cb);
          if (ca == -2) {
            ca = wait_for(ha2);
            if (ca >= 0) small_chunk_fetch(tv.nodes, ca, top_levels, lane, CA);
          }
        }
      } else if (!hier) {
        // FLAT: the workgroup's chunks p, p + P, p + 2 P, ... taken by its waves as they come free
        auto take = [&]() -> int {
          int n = 0;
          if (lane == 0) n = atomicAdd(&s_next[0], 1);
          return __builtin_amdgcn_readfirstlane(n) * P + p;
        };
#if defined(PCGX_STAMPS)
        long long t_go = 0, t_wait = 0, n_go = 0;
        const long long t_loop0 = clock64();
#define PCGX_GO_TIMED(CX, cx)                                                                                  \
  {                                                                                                            \
    const long long q0 = clock64();                                                                            \
    asm volatile("" : "+v"(CX.rec.x), "+v"(CX.rec.y), "+v"(CX.rec.z), "+v"(CX.rec.w));                         \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                            \
    const long long q1 = clock64();                                                                            \
    go_flat(CX, cx);                                                                                           \
    const long long q2 = clock64();                                                                            \
    t_wait += q1 - q0;                                                                                         \
    t_go += q2 - q1;                                                                                           \
    n_go++;                                                                                                    \
  }
#else
#define PCGX_GO_TIMED(CX, cx)                                                          \
  {                                                                                    \
    asm volatile("" : "+v"(CX.rec.x), "+v"(CX.rec.y), "+v"(CX.rec.z), "+v"(CX.rec.w)); \
    go_flat(CX, cx);                                                                   \
  }
#endif
        int ca = take();
        if (ca < nchunks) small_chunk_fetch(tv.nodes, ca, top_levels, lane, CA);
        while (ca < nchunks) {
          const int cb = take();
          if (cb < nchunks) small_chunk_fetch(tv.nodes, cb, top_levels, lane, CB);
          PCGX_GO_TIMED(CA, ca);
          if (cb >= nchunks) break;
          ca = take();
          if (ca < nchunks) small_chunk_fetch(tv.nodes, ca, top_levels, lane, CA);
          PCGX_GO_TIMED(CB, cb);
        }
#undef PCGX_GO_TIMED
#if defined(PCGX_STAMPS)
        if (it == PCGX_STAMP_ITER && blockIdx.x == 3 && threadIdx.x == 0) {  // (workgroup 3's first wave; its spare slots and workgroup 4's)
          g_stamps_small_fit[3 * 8 + 6] = (unsigned long long)t_go;
          g_stamps_small_fit[3 * 8 + 7] = (unsigned long long)t_wait;
          g_stamps_small_fit[4 * 8 + 6] = (unsigned long long)n_go;
          g_stamps_small_fit[4 * 8 + 7] = (unsigned long long)(clock64() - t_loop0);
        }
#endif
      } else {
        // BAND BY BAND: a band's chunks are looked at only below chunks whose sub-tree could not be ruled out (every
        // workgroup of the group keeps its own list -- it knows its own best -- and goes through the nodes of every P-th
        // chunk).  A strip of 64 neighbouring targets needs the sub-trees near it: of the 2048 chunks of a 16384-point
        // tree's last band a few dozen, and to rule one out by itself costs a fetch.
        int lr = 0, start = 0;
        for (int b = 0; lr < D; b++) {
          __syncthreads();  // (band b's list is whole)
          const int cnt = s_cnt[b];
          const uint16_t *list = s_list[b & 1];
          uint16_t *below = s_list[(b + 1) & 1];
          const int kb = b == 0 ? top_levels : kSmallBand;
          auto take = [&]() -> int {
            int n = 0;
            if (lane == 0) n = atomicAdd(&s_next[b], 1);
            return __builtin_amdgcn_readfirstlane(n);
          };
          auto chunk_of = [&](int n) -> int { return start + (int)list[n]; };
          auto done_with = [&](int c, uint32_t live) {  // (live: uniform)
            if (live == 0u) return;
            const int fan = __popc(live);
            int at = 0;
            if (lane == 0) at = atomicAdd(&s_cnt[b + 1], fan);
            at = __builtin_amdgcn_readfirstlane(at);
            if (lane < 16 && ((live >> lane) & 1u) != 0u) below[at + __popc(live & ((1u << lane) - 1u))] = (uint16_t)(((c - start) << kb) | lane);
          };
          int na = take();
          int ca = na < cnt ? chunk_of(na) : 0;
          if (na < cnt) small_chunk_fetch(tv.nodes, ca, top_levels, lane, CA);
          while (na < cnt) {
            const int nb = take();
            const int cb = nb < cnt ? chunk_of(nb) : 0;
            if (nb < cnt) small_chunk_fetch(tv.nodes, cb, top_levels, lane, CB);
            asm volatile("" : "+v"(CA.rec.x), "+v"(CA.rec.y), "+v"(CA.rec.z), "+v"(CA.rec.w));
            done_with(ca, go_bands(CA, ca, ca % P == p));
            if (nb >= cnt) break;
            na = take();
            ca = na < cnt ? chunk_of(na) : 0;
            if (na < cnt) small_chunk_fetch(tv.nodes, ca, top_levels, lane, CA);
            asm volatile("" : "+v"(CB.rec.x), "+v"(CB.rec.y), "+v"(CB.rec.z), "+v"(CB.rec.w));
            done_with(cb, go_bands(CB, cb, cb % P == p));
          }
          start += 1 << lr;
          lr += kb;
        }
      }
    }
    // ---- the group's other workgroups' minima (the group's first workgroup: a wave a partner, all at once), then the
    // workgroup's waves meet in LDS; the others hand theirs over, the first one decides
    PCGX_STAMP_IF(it == PCGX_STAMP_ITER, small_fit, 8, blockIdx.x, 2);
    if (p == 0) {
      for (int pp = 1 + wave; pp < P && alive; pp += kSmallWaves) {
        const unsigned long long *o = part_w + ((size_t)(g * P + pp) * kPartWords) * 64 + lane;
        unsigned long long wd[kPartWords];
        for (;;) {
          bool ok = true;
#pragma unroll
          for (int k = 0; k < kPartWords; k++) wd[k] = word_in(o + k * 64);
#pragma unroll
          for (int k = 0; k < kPartWords; k++) ok = ok && (uint32_t)(wd[k] >> 32) == tag;
          if (__ballot(ok) == ~0ull) break;
          if (small_give_up(sy, wait)) {
            alive = false;
            break;
          }
        }
        if (alive)
          A.meet(((unsigned long long)(uint32_t)wd[1] << 32) | (uint32_t)wd[0], (uint32_t)wd[2], ((unsigned long long)(uint32_t)wd[4] << 32) | (uint32_t)wd[3],
                 (uint32_t)wd[5]);
      }
    }
    s_part[0][wave][lane] = (uint32_t)A.fd;
    s_part[1][wave][lane] = (uint32_t)(A.fd >> 32);
    s_part[2][wave][lane] = A.fd_id;
    s_part[3][wave][lane] = (uint32_t)A.ld;
    s_part[4][wave][lane] = (uint32_t)(A.ld >> 32);
    s_part[5][wave][lane] = A.ld_id;
    __syncthreads();  // (s_part is written again an iteration on: behind a pose that needs every workgroup's terms)
    if (wave == 0) {
      A.clear();
#pragma unroll
      for (int w = 0; w < kSmallWaves; w++)
        A.meet(((unsigned long long)s_part[1][w][lane] << 32) | s_part[0][w][lane], s_part[2][w][lane],
               ((unsigned long long)s_part[4][w][lane] << 32) | s_part[3][w][lane], s_part[5][w][lane]);
      // (for the next iteration -- which begins behind a pose that needs this workgroup's minima or terms, below)
      s_best[lane] = 0xFFFFFFFFu;
      reset_queue();
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // (in LDS before anything of this wave's leaves the workgroup)
      if (p != 0) {
        unsigned long long *mine = part_w + ((size_t)blockIdx.x * kPartWords) * 64 + lane;
        word_out(mine + 0 * 64, tagged((uint32_t)A.fd, tag));
        word_out(mine + 1 * 64, tagged((uint32_t)(A.fd >> 32), tag));
        word_out(mine + 2 * 64, tagged(A.fd_id, tag));
        word_out(mine + 3 * 64, tagged((uint32_t)A.ld, tag));
        word_out(mine + 4 * 64, tagged((uint32_t)(A.ld >> 32), tag));
        word_out(mine + 5 * 64, tagged(A.ld_id, tag));
      }
      if (p == 0 && alive) {
        // the head of this file: the walk's answer out of the minima; then the pair's terms (evaluator.go:130-144)
        float4 best = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(-1));
        float best_d = kp.max_dist_sq;  // nothing in range: {-1, maxRange^2} (kdtree.go:100-103)
        if (i < nt) {
          if (!finite) {
            small_walk<kMinDist>(tv.nodes, m1, x, y, z, kp.max_dist_sq, kp.min_dist_sq, best, best_d);
          } else {
            uint32_t id = 0u;  // the partner: 16 * chunk + place; 0: none
            const uint32_t dmin_bits = (uint32_t)(A.fd >> 32);
            if (kMinDist && dmin_bits == 0u) {
              id = A.fd_id;  // under the MinDistSq cut: the first such point in visit order
            } else {
              const float dmin = __uint_as_float(dmin_bits);
              if (!(dmin > kp.max_dist_sq)) {
                if (A.ld_id != 0u && (uint32_t)(A.ld >> 32) == dmin_bits) id = A.ld_id;
                else if (dmin != kp.max_dist_sq) id = A.fd_id;
              }
            }
            if (id != 0u) {
              uint32_t r;
              int lr;
              small_chunk_root((int)(id >> 4), top_levels, r, lr);
              const uint32_t t = id & 15u;
              const int k = 31 - __clz((int)t);
              const uint32_t wb = (r << k) | (t - (1u << k));
              __hip_atomic_store(&partner_of[i], wb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              best = node_at(tv.nodes, wb);
              const float dx = best.x - x, dy = best.y - y, dz = best.z - z;
              best_d = (dx * dx + dy * dy) + dz * dz;
            }
          }
        }
        const bool found = i < nt && __float_as_int(best.w) >= 0;
        TermSrc S;
        S.match = nullptr;
        S.pos_of = nullptr;
        S.xyz = nullptr;
        S.nt = nt;
        S.project = false;  // (the target is re-projected already)
        S.weight_fn = kp.weight_fn;
        S.weight_a = kp.weight_a;
        S.raw = nullptr;
        float t[kStrictRows];
        (void)pair_terms(S, x, y, z, make_float4(best.x, best.y, best.z, found ? best_d : -1.0f), t);  // (no pair, padding: -0.0f)
        const int64_t ic = (perm != nullptr && i < nt) ? (int64_t)perm[i] : i;
#pragma unroll
        for (int k = 0; k < kStrictRows; k++) word_out(&terms[(int64_t)k * ntp + ic], tagged(__float_as_uint(t[k]), tag));
        const unsigned long long bits = __ballot(found);
        if (lane == 0) {
          valid[g] = bits;
          word_out(&counts_w[g], tagged((uint32_t)__popcll(bits), tag));
        }
      }
    }
    PCGX_STAMP_IF(it == PCGX_STAMP_ITER, small_fit, 8, blockIdx.x, 3);
    if (!alive) break;
    // ---- the sums (evaluator.go:122-145), out of the terms' words as they come.  Row r's first tile (2048 terms) is its
    // wave's chain; every further tile is summarised by a wave of its own meanwhile (mini::tile_record) and the row's
    // wave walks through the records.  Roles in an order in which nobody waits for a later one (a launch of few
    // workgroups gives a wave several): the tiles' waves tile by tile, then the rows'.
    if (worker >= 0) {
      const int nT = (int)((ntp + ss::kTile - 1) / ss::kTile);  // (<= kSmallMaxTiles)
      const int n_tilers = nrows * (nT - 1), n_roles = n_tilers + nrows;
      unsigned long long *tsum_w = reinterpret_cast<unsigned long long *>(scratch + kSmallTileSumsAt);
      unsigned long long *recs_w = reinterpret_cast<unsigned long long *>(scratch + kSmallTileRecsAt);
      // terms [begin, end) of row T added to s one after the other (eight blocks of 64 asked for together, the next eight
      // under these blocks' adds)
      auto chain_over = [&](const unsigned long long *T, float s, int64_t begin, int64_t end) -> float {
        constexpr int kB = 8;
        unsigned long long v[kB], vn[kB];
#pragma unroll
        for (int j = 0; j < kB; j++) v[j] = begin + j * 64 < end ? word_in(T + begin + j * 64) : tagged(0x80000000u, tag);
        for (int64_t base = begin; base < end && alive; base += kB * 64) {
#pragma unroll
          for (int j = 0; j < kB; j++) vn[j] = base + (kB + j) * 64 < end ? word_in(T + base + (kB + j) * 64) : tagged(0x80000000u, tag);
          for (;;) {  // this batch: every word this iteration's?
            bool ok = true;
#pragma unroll
            for (int j = 0; j < kB; j++) ok = ok && (uint32_t)(v[j] >> 32) == tag;
            if (__ballot(ok) == ~0ull) break;
            if (small_give_up(sy, wait)) {
              alive = false;
              break;
            }
#pragma unroll
            for (int j = 0; j < kB; j++)
              if (base + j * 64 < end && (uint32_t)(v[j] >> 32) != tag) v[j] = word_in(T + base + j * 64);
          }
          if (!alive) break;
#pragma unroll
          for (int j = 0; j < kB; j++) {
            if (base + j * 64 >= end) break;  // uniform
            s = chain64(s, __uint_as_float((uint32_t)v[j]));
          }
#pragma unroll
          for (int j = 0; j < kB; j++) v[j] = vn[j];
        }
        return s;
      };
      // the lane's leaf of tile k of row T (terms 32 * lane ... of the tile; behind the row's end: -0.0f, which changes no sum)
      auto load_leaf = [&](const unsigned long long *T, int k, float (&t)[ss::kLeaf]) {
        const int64_t i0 = (int64_t)k * ss::kTile + (int64_t)lane * ss::kLeaf;
#pragma unroll
        for (int h = 0; h < 2; h++) {  // (sixteen words at a time: registers)
          unsigned long long w[16];
          for (;;) {
            bool ok = true;
#pragma unroll
            for (int j = 0; j < 16; j++) w[j] = i0 + h * 16 + j < ntp ? word_in(T + i0 + h * 16 + j) : tagged(0x80000000u, tag);
#pragma unroll
            for (int j = 0; j < 16; j++) ok = ok && (uint32_t)(w[j] >> 32) == tag;
            if (__ballot(ok) == ~0ull) break;
            if (small_give_up(sy, wait)) {
              alive = false;
              break;
            }
          }
#pragma unroll
          for (int j = 0; j < 16; j++) t[h * 16 + j] = __uint_as_float((uint32_t)w[j]);
        }
      };
      auto leaf_sum = [&](const float (&t)[ss::kLeaf]) -> double {
        double v = 0.0;
#pragma unroll
        for (int j = 0; j < ss::kLeaf; j++) v += (double)t[j];
        return v;
      };
      auto wave_sum = [&](double v) -> double {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        return v;
      };
      auto put_tile_sum = [&](int row, int k, double v) {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
        if (lane == 0) {
          word_out(&tsum_w[((size_t)row * kSmallMaxTiles + k) * 2], tagged((uint32_t)bits, tag));
          word_out(&tsum_w[((size_t)row * kSmallMaxTiles + k) * 2 + 1], tagged((uint32_t)(bits >> 32), tag));
        }
      };
      for (int role = worker; role < n_roles && alive; role += nworkers) {  // uniform per wave
        if (role < n_tilers) {
          // ---- tile k >= 1 of a row: its record
          const int k = 1 + role / nrows, row = role % nrows;
          const unsigned long long *T = terms + (int64_t)row * ntp;
          float t[ss::kLeaf];
          double base = 0.0;
          if (k == 1) {  // (the first tile's float64 sum: by the second tile's wave, for everybody behind it too)
            load_leaf(T, 0, t);
            if (!alive) break;
            base = wave_sum(leaf_sum(t));
            put_tile_sum(row, 0, base);
          }
          load_leaf(T, k, t);
          if (!alive) break;
          const double lsum = leaf_sum(t);
          put_tile_sum(row, k, wave_sum(lsum));
          if (k > 1) {  // the float64 sum of everything in front of the tile: the tiles' sums, by their waves
            unsigned long long lo = 0ull, hi = 0ull;
            for (;;) {
              lo = lane < k ? word_in(&tsum_w[((size_t)row * kSmallMaxTiles + lane) * 2]) : tagged(0u, tag);
              hi = lane < k ? word_in(&tsum_w[((size_t)row * kSmallMaxTiles + lane) * 2 + 1]) : tagged(0u, tag);
              if (__ballot((uint32_t)(lo >> 32) == tag && (uint32_t)(hi >> 32) == tag) == ~0ull) break;
              if (small_give_up(sy, wait)) {
                alive = false;
                break;
              }
            }
            if (!alive) break;
            const double mine = lane < k ? __longlong_as_double((long long)(((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo)) : 0.0;
            base = wave_sum(mine);
          }
          const mini::Leaf L = mini::tile_guesses(t, base, lsum, lane);
          const ss::TileRec R = mini::tile_record(t, L, lane);
          const uint32_t *rw = reinterpret_cast<const uint32_t *>(&R);
          uint32_t mine = 0u;
#pragma unroll
          for (int j = 0; j < 16; j++) mine = lane == j ? rw[j] : mine;
          if (lane < 16) word_out(&recs_w[((size_t)row * kSmallMaxTiles + k) * 16 + lane], tagged(mine, tag));
          continue;
        }
        // ---- a row: its first tile term by term from 0.0f (evaluator.go:122: the state runs through a binade every few terms
        // there, nothing to summarise), the others by their records
        const int row = role - n_tilers;
        const unsigned long long *T = terms + (int64_t)row * ntp;
        PCGX_STAMP_WAVE_IF(it == PCGX_STAMP_ITER && row == 0, small_fit, 8, 1, 6);  // (workgroup 1's spare slots: row 0's sums)
        uint32_t s_bits = __float_as_uint(chain_over(T + lane, 0.0f, 0, ntp < ss::kTile ? ntp : (int64_t)ss::kTile));
        if (!alive) break;
        PCGX_STAMP_WAVE_IF(it == PCGX_STAMP_ITER && row == 0, small_fit, 8, 1, 7);
#if defined(PCGX_STAMPS)
        unsigned long long n_fail = 0ull;
#endif
        for (int k = 1; k < nT && alive; k++) {
          unsigned long long w = 0ull;
          for (;;) {
            w = lane < 16 ? word_in(&recs_w[((size_t)row * kSmallMaxTiles + k) * 16 + lane]) : tagged(0u, tag);
            if (__ballot((uint32_t)(w >> 32) == tag) == ~0ull) break;
            if (small_give_up(sy, wait)) {
              alive = false;
              break;
            }
          }
          if (!alive) break;
          ss::TileRec R;
          uint32_t *rw = reinterpret_cast<uint32_t *>(&R);
#pragma unroll
          for (int j = 0; j < 16; j++) rw[j] = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)w, j);
          if (R.key >= 0 && ss::apply(s_bits, R.key, R.s)) continue;  // the record covers the state: proven the additions' result
          if (R.cons && R.in == s_bits) {                             // ... or its guesses were the states themselves
            s_bits = R.out;
            continue;
          }
#if defined(PCGX_STAMPS)
          n_fail += 256ull + (R.key < 0 ? 1ull : 0ull);
#endif
          // else the additions one by one: 11 us a tile.  (The tile leaf by leaf under windows of the leaves' own, runs of
          // equal windows composed -- ss_host_model's resolve_tile, a leaf a lane -- was built and was SLOWER, 22 us: the tiles
          // that get here are the ones whose sums hover around zero, where most leaves have no window either.)
          const int64_t b0 = (int64_t)k * ss::kTile, b1 = b0 + ss::kTile < ntp ? b0 + ss::kTile : ntp;
          s_bits = __float_as_uint(chain_over(T + lane, __uint_as_float(s_bits), b0, b1));
        }
        if (!alive) break;
        const float s = __uint_as_float(s_bits);
#if defined(PCGX_STAMPS)
        if (it == PCGX_STAMP_ITER && lane == 0) g_small_fail[row] = n_fail;
#endif
        PCGX_STAMP_WAVE_IF(it == PCGX_STAMP_ITER && row == 0, small_fit, 8, 2, 6);
        const int slot = row == 0 ? S_VALUE : (row <= 6 ? S_G0 + row - 1 : (row == 7 ? S_DIST_RMS : S_WEIGHT));
        if (row == 0) {  // the pair count rides with the first sum
          uint32_t np = 0u;
          for (int w0 = 0; w0 < Q && alive; w0 += 64) {
            unsigned long long cw = 0ull;
            for (;;) {
              cw = w0 + lane < Q ? word_in(&counts_w[w0 + lane]) : tagged(0u, tag);
              if (__ballot((uint32_t)(cw >> 32) == tag) == ~0ull) break;
              if (small_give_up(sy, wait)) {
                alive = false;
                break;
              }
            }
            np += (uint32_t)cw;
          }
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) np += __shfl_xor(np, o);
          if (lane == 0 && alive) word_out(&sums_w[S_PAIRS], tagged(np, tag));
        }
        if (lane == 0 && alive) word_out(&sums_w[slot], tagged(__float_as_uint(s), tag));
      }
    }
    PCGX_STAMP_IF(it == PCGX_STAMP_ITER, small_fit, 8, blockIdx.x, 4);
    if (!alive) break;
    // ---- evaluate tail + pose update (evaluator.go:156-186, updater.go:44-71), and the new pose to everybody
    if (blockIdx.x == 0 && wave == 0) {
      unsigned long long w = 0ull;
      for (;;) {
        const bool mine = lane < S_COUNT && (lane != S_WEIGHT || nrows == kStrictRows);
        w = mine ? word_in(&sums_w[lane]) : tagged(0u, tag);
        if (__ballot((uint32_t)(w >> 32) == tag) == ~0ull) break;
        if (small_give_up(sy, wait)) {
          alive = false;
          break;
        }
      }
      if (!alive) break;
      PCGX_STAMP_WAVE_IF(it == PCGX_STAMP_ITER, small_fit, 8, 2, 7);  // (the sums are in)
      double sums[S_COUNT];
#pragma unroll
      for (int k = 0; k < S_COUNT; k++) {
        const uint32_t bitsk = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)w, k);
        sums[k] = k == S_PAIRS ? (double)bitsk : (double)__uint_as_float(bitsk);
      }
      if (nrows < kStrictRows) {  // default weight: 0 + 1 + 1 + ... in float32 is the pair count up to 2^24, where it stays
        const unsigned long long n = (unsigned long long)sums[S_PAIRS];
        sums[S_WEIGHT] = (double)(n < (1ull << 24) ? n : (1ull << 24));
      }
      if (lane == 0) {
#pragma unroll
        for (int k = 0; k < S_COUNT; k++) sums10[k] = sums[k];
        icp_update_step(state, sums, kp);
        // the new pose as words of the next iteration (the lane's own stores read back: one thread's accesses stay in order)
        const uint32_t next_tag = (launch_no << kSmallTagIterBits) | (uint32_t)(it + 2);
#pragma unroll
        for (int k = 0; k < 16; k++) word_out(&pose_w[k], tagged(__float_as_uint(state->trans[k]), next_tag));
        word_out(&pose_w[16], tagged((uint32_t)state->iter, next_tag));
        word_out(&pose_w[17], tagged((uint32_t)state->done, next_tag));
      }
    }
    PCGX_STAMP_IF(it == PCGX_STAMP_ITER, small_fit, 8, blockIdx.x, 5);
  }
#if defined(PCGX_STAMPS)
  if (threadIdx.x == 0 && blockIdx.x == 0) g_small_iter_t[61] = wall_clock64();
#endif
  // ---- the Fit's result straight to the host (pcgx_icp_fit): the loop state into the context's pinned mailbox, the
  // sequence word last; the host polls that word instead of waiting for the stream and copying (a copy command, a blit
  // kernel and two waits: 25 us behind a 160 us Fit).  By the updater's wave: its first lane wrote the state and reads it
  // back (one thread's accesses stay in order), through LDS to all lanes, which store a word each.
  if (mailbox != nullptr && blockIdx.x == 0 && wave == 0) {
    constexpr int kWords = (int)(sizeof(IcpState) / 4);
    uint32_t *s_state = &s_part[0][0][0];  // (kPartWords * kSmallWaves * 64 words: nobody's any more)
    if (lane == 0) {
      if (!alive) {
        state->status = PCGX_E_HIP;
        state->done = 1;
      }
      const uint32_t *src = reinterpret_cast<const uint32_t *>(state);
      uint32_t buf[kWords];  // (every load under way before the first store: word by word it was 68 round trips, 17 us)
#pragma unroll
      for (int k = 0; k < kWords; k++) buf[k] = src[k];
#pragma unroll
      for (int k = 0; k < kWords; k++) s_state[k] = buf[k];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // (every word with the sequence number in ONE 64-bit store: the host takes the state when all of them carry it --
    // no order among the stores to keep, nothing to wait for here.  Data, a wait, then a sequence word: the wait was 15 us.)
    unsigned long long *mb = reinterpret_cast<unsigned long long *>(const_cast<uint32_t *>(mailbox) + 2);
    for (int k = lane; k < kWords; k += 64) __hip_atomic_store(&mb[k], tagged(s_state[k], mailbox_seq), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
#if defined(PCGX_STAMPS)
  if (threadIdx.x == 0 && blockIdx.x == 0) g_small_iter_t[62] = wall_clock64();
#endif
  // ---- out: the last WAVE to leave puts the words back to zero (the next launch starts from zero); a launch that gave
  // up ends the Fit
  if (lane == 0) {
    if (!alive) {
      state->status = PCGX_E_HIP;
      state->done = 1;
    }
    // (a workgroup's waves count themselves in LDS first, and the launch's count lives on a line of its own: 1152 returning
    // atomics on ONE word are served one after the other, 10 ns each, and for those 11 us every wave that looked at the
    // abort word on the same line -- everybody who waits for something, every 64th look -- stood still: the launch's
    // last iteration took 36 us instead of 20)
    if (atomicAdd(&s_exited, 1) == kSmallWaves - 1) {
      unsigned int *exited = reinterpret_cast<unsigned int *>(scratch + kSmallExitedAt);
      const unsigned int e = __hip_atomic_fetch_add(exited, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (e == G - 1u) {
        __hip_atomic_store(&sy->abort, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(exited, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

// Groups of 64 targets that lie TOGETHER, whatever order the caller has them in: what a group's waves can rule out of
// the tree (small_chunk_go) is what is far from ALL 64.  No sort -- a place in a coarse grid's Morton order is enough:
// one workgroup counts the targets per cell (16 x 16 x 16 over the base cloud's box, LDS), scans, and hands out places
// (which of a cell's targets comes first is left to the atomics: the grouping decides how much work a Fit is, never
// what comes out -- the terms go to the caller's places, the sums run in the caller's order).
constexpr int kSmallOrderCells = 4096;
__global__ __launch_bounds__(1024) void small_order_kernel(const float *__restrict__ q, int nt, float lo0, float lo1, float lo2, float sc0,
                                                           float sc1, float sc2, int32_t *__restrict__ perm) {
  __shared__ uint32_t s_cnt[kSmallOrderCells];
  __shared__ uint32_t s_tot[1024];
  for (int c = threadIdx.x; c < kSmallOrderCells; c += 1024) s_cnt[c] = 0u;
  __syncthreads();
  auto cell_of = [&](int i) -> uint32_t {
    const float x = q[3 * i], y = q[3 * i + 1], z = q[3 * i + 2];
    auto axis = [](float v, float lo, float sc) -> uint32_t {
      const float f = (v - lo) * sc;  // (NaN, outside the box: the ends)
      return f >= 15.0f ? 15u : (f > 0.0f ? (uint32_t)f : 0u);
    };
    const uint32_t cx = axis(x, lo0, sc0), cy = axis(y, lo1, sc1), cz = axis(z, lo2, sc2);
    uint32_t m = 0u;
#pragma unroll
    for (int b = 0; b < 4; b++) m |= (((cx >> b) & 1u) << (3 * b)) | (((cy >> b) & 1u) << (3 * b + 1)) | (((cz >> b) & 1u) << (3 * b + 2));
    return m;
  };
  for (int i = threadIdx.x; i < nt; i += 1024) atomicAdd(&s_cnt[cell_of(i)], 1u);
  __syncthreads();
  // exclusive scan of the 4096 counts: four a thread, the threads' totals by a scan over the workgroup
  const int c0 = threadIdx.x * 4;
  const uint32_t a0 = s_cnt[c0], a1 = s_cnt[c0 + 1], a2 = s_cnt[c0 + 2], a3 = s_cnt[c0 + 3];
  s_tot[threadIdx.x] = a0 + a1 + a2 + a3;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const uint32_t add = (int)threadIdx.x >= o ? s_tot[threadIdx.x - o] : 0u;
    __syncthreads();
    s_tot[threadIdx.x] += add;
    __syncthreads();
  }
  const uint32_t before = threadIdx.x > 0 ? s_tot[threadIdx.x - 1] : 0u;
  s_cnt[c0] = before;
  s_cnt[c0 + 1] = before + a0;
  s_cnt[c0 + 2] = before + a0 + a1;
  s_cnt[c0 + 3] = before + a0 + a1 + a2;
  __syncthreads();
  for (int i = threadIdx.x; i < nt; i += 1024) perm[atomicAdd(&s_cnt[cell_of(i)], 1u)] = i;
}

// A small session's start in one launch: the loop state of a fresh Fit (icp.go:47: identity, counters zero), the
// target's coordinates by component in the caller's order, and zeroes in every word a launch reads before it writes
// (the tags say "not this iteration's" of anything else, but a zero says it of every launch there will ever be).
__global__ __launch_bounds__(256) void small_prepare_kernel(const float *__restrict__ q, const int32_t *__restrict__ perm, int64_t nt, float *__restrict__ x,
                                                            float *__restrict__ y, float *__restrict__ z, uint32_t *__restrict__ pos_of,
                                                            IcpState *__restrict__ state, uint4 *__restrict__ terms16, int64_t n_terms16,
                                                            uint4 *__restrict__ sync16, int64_t n_sync16) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
  if (t == 0) {
    IcpState h;
    memset(&h, 0, sizeof h);
    const Mat4 id = mat4_translate(0.0f, 0.0f, 0.0f);
    for (int k = 0; k < 16; k++) h.trans[k] = id.m[k];
    *state = h;
  }
  for (int64_t pos = t; pos < nt; pos += stride) {
    const int64_t i = perm ? (int64_t)perm[pos] : pos;
    pos_of[i] = (uint32_t)pos;  // where the caller's target i sits in the session's order
    x[pos] = q[3 * i];
    y[pos] = q[3 * i + 1];
    z[pos] = q[3 * i + 2];
  }
  const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
  for (int64_t i = t; i < n_terms16; i += stride) terms16[i] = zero;
  for (int64_t i = t; i < n_sync16; i += stride) sync16[i] = zero;
}

static std::atomic<long long> g_small_launches[3];

// ---- host side --------------------------------------------------------------------------------------------------
static int64_t small_knob(const char *name, int64_t def) {
  if (const char *e = getenv(name)) return (int64_t)atoll(e);
  return def;
}
// The chunks of a tree, the workgroups a group of 64 targets gets, the chunks a wave then has to look at
struct SmallShape {
  int64_t nchunks;
  int P;
  int64_t per_wave;
};
static SmallShape small_shape(const TreeView &tv, int64_t nt) {
  SmallShape S;
  const int Q = (int)((nt + 63) / 64);
  const int top_levels = tv.depth % kSmallBand == 0 ? kSmallBand : tv.depth % kSmallBand;
  S.nchunks = 1;
  for (int lr = top_levels; lr < tv.depth; lr += kSmallBand) S.nchunks += (int64_t)1 << lr;
  // workgroups per group: as many as the chip has room for (256 resident for certain) and the group has chunks for
  S.P = Q >= 1 && Q <= 256 ? 256 / Q : 1;
  const int64_t p_chunks = (S.nchunks + kSmallWaves - 1) / kSmallWaves;
  if ((int64_t)S.P > p_chunks) S.P = (int)p_chunks;
  if (S.P < 1) S.P = 1;
  const int p_forced = (int)small_knob("PCGX_ICP_SMALL_P", 0);  // (tests: read at every launch)
  if (p_forced > 0 && p_forced * Q <= 256) S.P = p_forced;
  S.per_wave = (S.nchunks + (int64_t)S.P * kSmallWaves - 1) / ((int64_t)S.P * kSmallWaves);
  return S;
}
// A session whose Fit runs in one launch: up to 256 groups of 64 targets, a visit-order key of 32 bits -- and where it is
// the faster way.  An iteration here is ~12 us of hand-overs, deciding and the pose update, 4.9 ns a target for the sums'
// chains and ~0.7 us for every chunk a wave has to look at; the general path's is ~42 us whatever the sizes ON CLOUDS
// ITS WALK LIKES (tools/small_vs_general.py, random surfaces, 20-iteration host-pointer Fits, ms, here / there: 1000 x
// 1000 0.39 / 0.77, 2000 x 2000 0.56 / 0.80, 4000 x 4000 0.89 / 0.86, 8000 x 8000 1.09 / 0.89) and several times that
// on clouds whose coordinates repeat (pcgx_kdtree::many_ties -- the reference's own benchmark's ground plane,
// icp_test.go:100-142, 10 iterations: 4096 points 0.58 / 1.84, 16384 points 1.24 / 3.9).  PCGX_ICP_SMALL_TARGET / _BASE /
// _PAIRS set limits of their own (all three: the one launch wherever it can run -- the tests).
bool small_fit_eligible(const TreeView &tv, int64_t nt, bool many_ties) {
  if (!(tv.n >= 1 && tv.n <= 65535 && tv.depth <= 16 && nt >= 1 && nt <= 16384)) return false;
  const int64_t max_base = small_knob("PCGX_ICP_SMALL_BASE", -1), max_nt = small_knob("PCGX_ICP_SMALL_TARGET", -1);
  const int64_t max_pairs = small_knob("PCGX_ICP_SMALL_PAIRS", (int64_t)1 << 28);
  if (nt * (int64_t)tv.n > max_pairs) return false;
  if (max_base >= 0 || max_nt >= 0) return (max_base < 0 || tv.n <= max_base) && (max_nt < 0 || nt <= max_nt);
  if (many_ties) return tv.n <= 32767;
  const SmallShape S = small_shape(tv, nt);
  return 12.0 + 0.0049 * (double)nt + 0.7 * (double)S.per_wave <= 36.0;
}
size_t small_fit_sync_bytes() { return (kSmallScratchBytes + 15) & ~(size_t)15; }
size_t small_fit_terms_bytes(int64_t nt) { return (size_t)kStrictRows * (size_t)((nt + 63) & ~(int64_t)63) * sizeof(unsigned long long); }
int small_fit_max_iters() { return (1 << kSmallTagIterBits) - 2; }
// a target large enough for the grouping to matter (and for a 20 us launch in front of the Fit not to)
bool small_fit_wants_order(int64_t nt) { return nt > small_knob("PCGX_ICP_SMALL_ORDER_FROM", 2048); }

pcgx_status small_fit_prepare(const float *d_target_aos, int64_t nt, const float box_lo[3], const float box_hi[3], int32_t *d_perm, float *d_xyz,
                               uint32_t *d_pos_of, IcpState *state, void *terms, void *sync, hipStream_t st) {
  if (d_perm) {  // (small_fit_wants_order)
    float sc[3], lo[3];
    for (int k = 0; k < 3; k++) {
      const float ext = box_hi[k] - box_lo[k];
      lo[k] = box_lo[k] == box_lo[k] ? box_lo[k] : 0.0f;
      sc[k] = (ext > 0.0f && ext < 3.0e38f) ? 16.0f / ext : 0.0f;
    }
    hipLaunchKernelGGL(small_order_kernel, dim3(1), dim3(1024), 0, st, d_target_aos, (int)nt, lo[0], lo[1], lo[2], sc[0], sc[1], sc[2], d_perm);
  }
  const int64_t n_terms16 = (int64_t)(small_fit_terms_bytes(nt) / 16), n_sync16 = (int64_t)(small_fit_sync_bytes() / 16);
  static_assert(kSmallPartnersAt % 16 == 0, "the scratch block in 16-byte pieces");
  const int64_t most = n_terms16 > n_sync16 ? n_terms16 : n_sync16;
  int64_t blocks = (most + 255) / 256;
  if (blocks > 512) blocks = 512;
  hipLaunchKernelGGL(small_prepare_kernel, dim3((unsigned)blocks), dim3(256), 0, st, d_target_aos, (const int32_t *)d_perm, nt, d_xyz, d_xyz + nt,
                     d_xyz + 2 * nt, d_pos_of, state, (uint4 *)terms, n_terms16, (uint4 *)sync, n_sync16);
  PCGX_HIP_TRY(hipGetLastError());
  return PCGX_OK;
}

pcgx_status small_fit_enqueue(const TreeView &tv, const float *tx, const float *ty, const float *tz, int64_t nt, IcpState *state,
                              const IcpKernelParams &kp, void *terms, unsigned long long *valid, double *sums10, void *sync,
                              uint32_t launch_no, int iters, const int32_t *perm, hipStream_t st, volatile uint32_t *mailbox,
                              uint32_t mailbox_seq) {
  static_assert(sizeof(IcpState) % 4 == 0 && 2 * sizeof(IcpState) + 8 <= kMailboxBytes, "the loop state, a tag a word, fits the mailbox");
  static_assert(sizeof(SmallSync) <= kSmallPoseAt && kSmallPoseAt + kSmallPoseWords * 8 <= kSmallSumsAt &&
                    kSmallSumsAt + kSmallSumsWords * 8 <= kSmallCountsAt && kSmallCountsAt + 256 * 8 <= kSmallExitedAt && kSmallExitedAt + 64 <= kSmallPartAt && S_COUNT <= kSmallSumsWords,
                "the scratch block's layout");
  if (iters < 1 || iters > small_fit_max_iters()) return fail(PCGX_E_INVALID, "icp (one launch): %d iterations in a launch", iters);
  const int64_t ntp = (nt + 63) & ~(int64_t)63;
  const int Q = (int)(ntp / 64);
  const SmallShape S = small_shape(tv, nt);
  const int P = S.P;
  const int64_t nchunks = S.nchunks;
  if (Q > 256 || Q * P > 256) return fail(PCGX_E_INVALID, "icp (one launch): %d groups of targets", Q);
  // How a group's waves go through the tree's chunks (PCGX_ICP_SMALL_HIER: 0 / 2 / 1 force one; tests, measurements):
  //   chunk after chunk   every chunk of the workgroup's share is looked at (most are ruled out by the plane test's bound):
  //                       where a wave has fewer than eight to look at;
  //   a queue             the upper bands' chunks from the start, below them only what hangs under a chunk that could not
  //                       be ruled out, whoever is free takes the next: from eight chunks a wave on;
  //   band by band        the same with a list a band and a barrier between bands (every one of the group's P workgroups
  //                       keeps a list of its own): trees a workgroup has to itself, from 128 chunks a wave on.
  // Measured (10- / 20-iteration host-pointer Fits, ms; chunk after chunk / queue / band by band): the benchmark's plane
  // at 1024 points (1 chunk a wave) 0.22 / 0.25 / 0.36, at 4096 (17) 0.65 / 0.57 / 0.76, at 16384 (273) 5.9 / 1.27 / 1.23; a
  // random surface 8000 x 8000 (35) 1.77 / 1.09 / 1.06.  The bound from last time's partner costs a dependent fetch in
  // front of the first chunk: where a wave has eight chunks or more to rule out with it.
  const int hier_forced = (int)small_knob("PCGX_ICP_SMALL_HIER", -1);  // (read at every launch)
  const int64_t per_wave = S.per_wave;
  (void)nchunks;
  const int hier = hier_forced >= 0 ? (hier_forced == 1) : (per_wave >= 128);
  const int queued = hier_forced >= 0 ? (hier_forced == 2 || hier_forced == 3) : (per_wave >= 8 && per_wave < 128);
  const int queued_flat = hier_forced == 3;  // (every chunk on the queue from the start)
  const int seeded = hier || queued || per_wave >= 8;
  g_small_launches[0]++;
  if (hier || queued) g_small_launches[1]++;
  if (perm) g_small_launches[2]++;
  launch_no &= (1u << (32 - kSmallTagIterBits)) - 1u;
  if (kp.min_dist_sq > 0.0f)
    hipLaunchKernelGGL(icp_small_fit_kernel<true>, dim3((unsigned)(Q * P)), dim3(kSmallBlock), 0, st, tv, tx, ty, tz, nt, ntp, state, kp,
                       (unsigned long long *)terms, valid, sums10, (char *)sync, launch_no, P, hier | (seeded << 1) | (queued << 2) | (queued_flat << 3) | ((int)small_knob("PCGX_ICP_SMALL_SEEDS", 4) << 8), iters, perm, mailbox, mailbox_seq);
  else
    hipLaunchKernelGGL(icp_small_fit_kernel<false>, dim3((unsigned)(Q * P)), dim3(kSmallBlock), 0, st, tv, tx, ty, tz, nt, ntp, state, kp,
                       (unsigned long long *)terms, valid, sums10, (char *)sync, launch_no, P, hier | (seeded << 1) | (queued << 2) | (queued_flat << 3) | ((int)small_knob("PCGX_ICP_SMALL_SEEDS", 4) << 8), iters, perm, mailbox, mailbox_seq);
  PCGX_HIP_TRY(hipGetLastError());
  return PCGX_OK;
}

}  // namespace pcgx

extern "C" pcgx_status pcgx_debug_icp_one_launch(int64_t out[3], int32_t reset) {
  if (!out) return pcgx::fail(PCGX_E_INVALID, "pcgx_debug_icp_one_launch: NULL argument");
  for (int k = 0; k < 3; k++) {
    out[k] = (int64_t)pcgx::g_small_launches[k].load();
    if (reset) pcgx::g_small_launches[k].store(0);
  }
  return PCGX_OK;
}
