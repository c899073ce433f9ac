// icp_small.hip -- PointToPointICPGradient.Fit for SMALL clouds in ONE launch.
//
// Reference: pc/registration/icp/icp.go:23-67 (Fit), correspondence.go:22-37 (Pairs), evaluator.go:91-189 (Evaluate),
// updater.go:44-71 (Update); pc/storage/kdtree/kdtree.go:83-146,199-222 (Nearest).  The reference's own benchmark of
// the path, BenchmarkPointToPointICPGradient (icp_test.go:100-142), runs 1024 ... 16384 points with MinDistSq = res^2:
// the approximate search, whose answer depends on the walk's visit order.  The general path (icp.hip + strict.hip)
// takes four to five dependent launches per iteration, ~0.1 ms however few the points, and its walk kernel is built
// for a million queries: 96 us per iteration at 1024 targets, 388 at 16384 (profiles/r05e_rows.json).
//
// Here a Fit is one persistent launch of G = ceil(nt / 512) workgroups (G <= 64: they are resident together on the
// chip's 256 CUs) that loops over the iterations with two grid barriers in each:
//   walk   every lane its target: re-projection from the ORIGINAL target (icp.go:62-64, mat/mat4.go:130-137) and the
//          reference's walk as an in-order traversal WITHOUT a stack -- in the implicit BFS tree a node's parent is a
//          shift of its index, and which side of the parent it hangs on follows from the parent's split value, so
//          "where did I come from" is recomputed instead of stored.  The split values of all inner nodes live in LDS
//          (4 B per BFS slot of the levels above the last: 64 KB at 16384 ... 32767 points), so a descent step, a
//          pruned pivot (kdtree.go:111-115) and an unwind step never leave the CU; only a pivot or leaf whose distance
//          is evaluated reads its 16-byte record (L2-resident: 512 KB at 16k points).  Same visits, same order, same
//          float32 expressions as kdtree.go:94-146: ids, DistSq bits, tie winners and the MinDistSq cut are the
//          reference's.  Then the pair's nine float32 terms (evaluator.go:130-144; strict_terms.h, pair_terms) into
//          rows of the caller's target order -- small sessions keep that order, there is no Morton pass.
//   -- grid barrier --
//   sums   evaluator.go:122-145 adds the terms up in float32, one after the other from 0.0f: row r's chain is ONE wave's
//          (a wave of its own on a SIMD of its own where there are enough workgroups), 64 terms per coalesced load, the
//          adds by v_readlane + v_add_f32: a dependent add every 8 cycles, 3.3 ns -- 3.4 us at 1024 targets, 55 us at
//          16384.  (The summaries of strict_sum.h pay from ~10^5 terms on; below that their launches cost more than the
//          chain itself.)  The row that finishes last runs the evaluate tail and the pose update (evaluator.go:156-186,
//          updater.go:44-71: icp_update_step, the code the other paths run).
//   -- grid barrier --
// A barrier is an arrival count in device memory (agent-scope atomics, release / acquire fences); every wait is bounded
// by wall-clock time and looks at an abort word: a workgroup that gives up raises it, everybody leaves the kernel and
// the Fit ends with PCGX_E_HIP -- the grid drains whatever happens.
#include "knn_walk.h"
#include "strict_terms.h"

namespace pcgx {

#ifndef PCGX_SMALL_BLOCK
#define PCGX_SMALL_BLOCK 512
#endif
constexpr int kSmallBlock = PCGX_SMALL_BLOCK;
constexpr long long kSmallBarrierTicks = 200000000;  // 2 s (s_memrealtime: 100 MHz)

struct SmallSync {  // device words, zero between launches
  unsigned int arrived;   // barrier arrivals since the launch began
  unsigned int abort;     // a workgroup gave up
  unsigned int rows_done; // ticket of the sums' rows
  unsigned int exited;    // workgroups that have left the loop (the last one zeroes the block)
};

__device__ __forceinline__ bool small_barrier(SmallSync *sy, unsigned int &target, unsigned int nblocks) {
  __shared__ int s_ok;
  __syncthreads();
  if (threadIdx.x == 0) {
    target += nblocks;
    __threadfence();  // the workgroup's stores out (and, behind the wait, the others' in)
    __hip_atomic_fetch_add(&sy->arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int ok = 1;
    long long t_first = 0;
    for (int spins = 0;; spins++) {
      if (__hip_atomic_load(&sy->arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) break;
      if ((spins & 63) == 63) {
        if (__hip_atomic_load(&sy->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
          ok = 0;
          break;
        }
        const long long now = (long long)wall_clock64();
        if (t_first == 0) t_first = now;
        if (now - t_first > kSmallBarrierTicks) {
          __hip_atomic_store(&sy->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = 0;
          break;
        }
      }
      __builtin_amdgcn_s_sleep(1);
    }
    __threadfence();
    s_ok = ok;
  }
  __syncthreads();
  return s_ok != 0;
}

// kdtree.go:94-146 on the implicit tree (pcgx_internal.h: node b's children are 2b and 2b + 1, its depth floor(log2 b),
// its size a closed form of b: node_size), as the in-order walk  visit(near) ; test node ; visit(far)  with one running
// best (knn_walk.h says why that is the reference's recursion) -- and no stack: see the head of this file.
// One turn of the loop = ONE point whose distance is evaluated: the moves that need split values only -- the descent to a
// leaf, the way up past sub-trees that are done and pivots the plane test drops (kdtree.go:111-115) -- run out of LDS
// in front of it, then every lane fetches its one record together (the records of the tree's upper levels are in LDS
// too: s_rec, 2^rec_levels slots; below them: the L2).  As two branches with a fetch each, a wave paid both round trips
// every turn: 96 us per iteration at 1024 points.
template <bool kMinDist>
__device__ __forceinline__ void small_walk(const float4 *__restrict__ nodes, const float *s_split, const float4 *s_rec, uint32_t rec_slots,
                                           uint32_t m1, float qx, float qy, float qz, float max_dist_sq, float min_dist_sq,
                                           float4 &best, float &best_d) {
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
        b = sz == 2u ? 2u * b : (s_split[b] > qv ? 2u * b : 2u * b + 1u);  // one child: that one; else pivot > p -> child0 (:216)
        d++;
      }
    } else {
      bool at_pivot = false;
      while (b != 1u) {
        const uint32_t p = b >> 1;
        const int dp = d - 1;
        szp = node_size(p, dp, m1);
        const float qv = sel3(dp % 3, qx, qy, qz), sv = s_split[p];
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
    const float4 nd = b < rec_slots ? s_rec[b] : node_at(nodes, b);
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

// One launch = `iters` iterations of Fit's loop (icp.go:48-65) from the state in *state; see the head of this file.
// tx / ty / tz: the ORIGINAL target in the caller's order; terms: [kStrictRows][ntp] float32, ntp = nt rounded up to 64
// (<= gridDim.x * kSmallBlock); valid: [ntp / 64] matched-target bits; sums10: the session's sums (device memory).
template <bool kMinDist>
__global__ __launch_bounds__(kSmallBlock) void icp_small_fit_kernel(TreeView tv, const float *__restrict__ tx,
                                                                    const float *__restrict__ ty, const float *__restrict__ tz,
                                                                    int64_t nt, int64_t ntp, IcpState *__restrict__ state,
                                                                    IcpKernelParams kp, float *__restrict__ terms,
                                                                    unsigned long long *__restrict__ valid,
                                                                    double *__restrict__ sums10, SmallSync *__restrict__ sy,
                                                                    int iters, int rec_slots_arg) {
  // LDS: the records {x, y, z, id} of the BFS slots below rec_slots (the whole tree up to 8191 points, its upper twelve
  // levels beyond), then the split value of every node that can have children, [2^(depth - 1)]
  extern __shared__ float4 s_rec[];
  const uint32_t rec_slots = (uint32_t)rec_slots_arg;
  float *s_split = reinterpret_cast<float *>(s_rec + rec_slots);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned int G = gridDim.x;
  const uint32_t m1 = (uint32_t)tv.n + 1u;
  {
    const uint32_t inner = tv.depth > 1 ? 1u << (tv.depth - 1) : 1u;
    for (uint32_t b = threadIdx.x; b < inner; b += kSmallBlock)
      s_split[b] = b >= 1u ? node_comp(tv.nodes, b, (31 - __clz((int)b)) % 3) : 0.0f;  // (slots of absent nodes: never looked at)
    for (uint32_t b = threadIdx.x; b < rec_slots; b += kSmallBlock) s_rec[b] = node_at(tv.nodes, b);
  }
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * kSmallBlock + threadIdx.x;
  float x0 = 0.0f, y0 = 0.0f, z0 = 0.0f;
  if (i < nt) {
    x0 = tx[i];
    y0 = ty[i];
    z0 = tz[i];
  }
  const int nrows = kp.weight_fn == PCGX_WEIGHT_ONE ? kStrictRows - 1 : kStrictRows;
  unsigned int bar_target = 0u;
  bool alive = true;
  for (int it = 0; it < iters; it++) {
    // ---- the loop state (the update of the iteration before: behind the barrier, past the caches)
    float m[16];
#pragma unroll
    for (int k = 0; k < 16; k++) m[k] = __hip_atomic_load(&state->trans[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int upd_iter = __hip_atomic_load(&state->iter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int done = __hip_atomic_load(&state->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (done) break;  // uniform over the grid: every workgroup reads the same state
    // ---- correspondence (correspondence.go:25-36) and the pair's terms (evaluator.go:130-144)
    {
      float x = x0, y = y0, z = z0;
      if (upd_iter > 0) mat4_transform(m, x0, y0, z0, x, y, z);  // icp.go:27-30,62-64
      float4 best = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(-1));
      float best_d = kp.max_dist_sq;
      if (i < nt) small_walk<kMinDist>(tv.nodes, s_split, s_rec, rec_slots, m1, x, y, z, kp.max_dist_sq, kp.min_dist_sq, best, best_d);
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
      if (i < ntp) {
#pragma unroll
        for (int k = 0; k < kStrictRows; k++) terms[(int64_t)k * ntp + i] = t[k];
      }
      const unsigned long long bits = __ballot(found);
      if (lane == 0 && i < ntp) valid[i >> 6] = bits;
    }
    if (!small_barrier(sy, bar_target, G)) {
      alive = false;
      break;
    }
    // ---- the sums: row r by worker r (workers: wave w of workgroup g is w * G + g -- a wave of its own workgroup, hence
    // of a SIMD of its own, wherever there are nine workgroups), evaluator.go:122-145
    const int worker = wave * (int)G + (int)blockIdx.x, nworkers = (kSmallBlock / 64) * (int)G;
    for (int row = worker; row < nrows; row += nworkers) {  // uniform per wave
      const float *T = terms + (int64_t)row * ntp;
      float s = 0.0f;  // evaluator.go:122
      float v = ntp > 0 ? T[lane] : -0.0f;
      for (int64_t base = 0; base < ntp; base += 64) {
        const float cur = v;
        if (base + 64 < ntp) v = T[base + 64 + lane];  // (the next 64 terms are on their way while these are added)
#pragma unroll
        for (int k = 0; k < 64; k++) s = s + __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cur), k));
      }
      unsigned long long np = 0ull;
      if (row == 0) {  // the pair count rides with the first sum
        for (int64_t w = lane; w < (ntp >> 6); w += 64) np += (unsigned long long)__popcll(valid[w]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) np += __shfl_xor(np, o);
      }
      unsigned int ticket = 0u;
      if (lane == 0) {
        const int slot = row == 0 ? S_VALUE : (row <= 6 ? S_G0 + row - 1 : (row == 7 ? S_DIST_RMS : S_WEIGHT));
        __hip_atomic_store(&sums10[slot], (double)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (row == 0) __hip_atomic_store(&sums10[S_PAIRS], (double)np, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        ticket = __hip_atomic_fetch_add(&sy->rows_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ticket == (unsigned)nrows - 1u) {  // the last row: evaluate tail + pose update (evaluator.go:156-186, updater.go:44-71)
          __threadfence();
          __hip_atomic_store(&sy->rows_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          double sums[S_COUNT];
          for (int k = 0; k < S_COUNT; k++) sums[k] = __hip_atomic_load(&sums10[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (nrows < kStrictRows) {  // default weight: 0 + 1 + 1 + ... in float32 is the pair count up to 2^24, where it stays
            const unsigned long long n = (unsigned long long)sums[S_PAIRS];
            sums[S_WEIGHT] = (double)(n < (1ull << 24) ? n : (1ull << 24));
            sums10[S_WEIGHT] = sums[S_WEIGHT];
          }
          icp_update_step(state, sums, kp);
        }
      }
    }
    if (!small_barrier(sy, bar_target, G)) {
      alive = false;
      break;
    }
  }
  // ---- out: the last workgroup to leave puts the words back to zero (the next launch starts from zero); a launch that
  // gave up ends the Fit
  __syncthreads();
  if (threadIdx.x == 0) {
    if (!alive) {
      state->status = PCGX_E_HIP;
      state->done = 1;
    }
    __threadfence();
    const unsigned int e = __hip_atomic_fetch_add(&sy->exited, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (e == G - 1u) {
      __hip_atomic_store(&sy->arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&sy->abort, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&sy->rows_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&sy->exited, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ---- host side --------------------------------------------------------------------------------------------------
static int64_t small_knob(const char *name, int64_t def) {
  if (const char *e = getenv(name)) return (int64_t)atoll(e);
  return def;
}
// a session whose Fit runs in one launch: a base tree whose inner levels fit LDS, a target of at most 64 workgroups
bool small_fit_eligible(const TreeView &tv, int64_t nt) {
  // Measured (tests/perf_rows_ref.py, the reference's benchmark shapes, 10 iterations, session-resident): 1024 points
  // 0.79 ms against the general path's 0.96; 4096: 1.98 against 1.84; 16384: 6.7 against 3.9 -- a lane walks its query
  // alone here (a wave takes as long as its slowest lane: ~500 visits of ~200 cycles on that data, whose ground plane
  // ties every third level of the tree), where the general path's waves refill finished lanes with new queries.  So
  // the one launch is for what it wins: up to 2048 targets on a tree that fits LDS whole (PCGX_ICP_SMALL_TARGET /
  // _BASE widen it: the tests run it up to 32768 x 32767).
  const int64_t max_base = small_knob("PCGX_ICP_SMALL_BASE", 8191), max_nt = small_knob("PCGX_ICP_SMALL_TARGET", 2048);
  return tv.n >= 1 && tv.n <= max_base && tv.n <= 32767 && tv.depth <= 16 && nt >= 1 && nt <= max_nt && nt <= 64 * kSmallBlock;
}
size_t small_fit_sync_bytes() { return sizeof(SmallSync); }

pcgx_status small_fit_enqueue(const TreeView &tv, const float *tx, const float *ty, const float *tz, int64_t nt, IcpState *state,
                              const IcpKernelParams &kp, float *terms, unsigned long long *valid, double *sums10, void *sync,
                              int iters, hipStream_t st) {
  const int64_t ntp = (nt + 63) & ~(int64_t)63;
  const unsigned G = (unsigned)((nt + kSmallBlock - 1) / kSmallBlock);
  const size_t split_bytes = (size_t)(tv.depth > 1 ? 1u << (tv.depth - 1) : 1u) * sizeof(float);
  // the records of as many upper levels as fit beside the split values (a workgroup per CU: 144 KB of its 160)
  constexpr size_t kLdsBudget = 144 * 1024;
  int rec_levels = tv.depth;
  while (rec_levels > 0 && ((size_t)16 << rec_levels) + split_bytes > kLdsBudget) rec_levels--;
  const int rec_slots = 1 << rec_levels;
  const size_t lds = (size_t)rec_slots * 16 + split_bytes;
  static const bool attr_ok = [] {  // (dynamic LDS beyond 64 KB has to be asked for, once per kernel)
    return hipFuncSetAttribute(reinterpret_cast<const void *>(&icp_small_fit_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudget) == hipSuccess &&
           hipFuncSetAttribute(reinterpret_cast<const void *>(&icp_small_fit_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudget) == hipSuccess;
  }();
  if (!attr_ok) (void)hipGetLastError();
  if (lds > 64 * 1024 && !attr_ok) return fail(PCGX_E_HIP, "icp (one launch): %zu bytes of LDS refused", lds);
  if (kp.min_dist_sq > 0.0f)
    hipLaunchKernelGGL(icp_small_fit_kernel<true>, dim3(G), dim3(kSmallBlock), lds, st, tv, tx, ty, tz, nt, ntp, state, kp, terms, valid,
                       sums10, (SmallSync *)sync, iters, rec_slots);
  else
    hipLaunchKernelGGL(icp_small_fit_kernel<false>, dim3(G), dim3(kSmallBlock), lds, st, tv, tx, ty, tz, nt, ntp, state, kp, terms, valid,
                       sums10, (SmallSync *)sync, iters, rec_slots);
  PCGX_HIP_TRY(hipGetLastError());
  return PCGX_OK;
}

}  // namespace pcgx
