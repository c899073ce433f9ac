// icp.hip -- point-to-point ICP (gradient) on gfx950: fused
// transform + nearest + reduction kernel, device-side evaluate-tail/update
// kernel, the device-resident session and the pcgx_icp_* C ABI.
//
// Reference: pc/registration/icp/icp.go:23-67 (Fit), correspondence.go:22-37
// (Pairs), evaluator.go:91-189 (Evaluate), updater.go:44-71 (Update).
//
// kPlane variants: the point-to-plane / Gauss-Newton extension (6x6 normal equations,
// SURVEY 8(f) N5; pcgx_math.h).  Same correspondence phase; the reduction accumulates the
// 30 sums {r^2, J r, upper triangle of J J^T, w, pairs} instead of the reference's 10.
// The reference has no such evaluator (only the Hessian slot): no reference parity.
#include <string.h>

#include <array>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>

#include <vector>

#include "knn_grid.h"
#include "knn_xwalk.h"
#include "strict_terms.h"

namespace pcgx {

constexpr int kIcpBlock = kKnnBlock;

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  return v;  // lane 0 holds the sum (fixed tree order -> deterministic)
}

__device__ __forceinline__ void accumulate_terms(double *acc, float x0, float y0, float z0, const float4 &bp,
                                                 const IcpKernelParams &kp) {
  // evaluator.go:130-144; every term is formed in float32 as the reference forms it (w = 1: exact)
  const float x1 = bp.x, y1 = bp.y, z1 = bp.z;
  const float w = eval_weight_fn(kp.weight_fn, kp.weight_a, bp.w);
  acc[S_VALUE] += (double)(w * bp.w);
  acc[S_G0 + 0] += (double)(w * (x0 - x1));
  acc[S_G0 + 1] += (double)(w * (y0 - y1));
  acc[S_G0 + 2] += (double)(w * (z0 - z1));
  acc[S_G0 + 3] += (double)(w * (z0 * y1 - y0 * z1));
  acc[S_G0 + 4] += (double)(w * (x0 * z1 - z0 * x1));
  acc[S_G0 + 5] += (double)(w * (y0 * x1 - x0 * y1));
  acc[S_DIST_RMS] += (double)(w * norm_sq3(x0, y0, z0));
  acc[S_WEIGHT] += (double)w;
  acc[S_PAIRS] += 1.0;
}

__device__ __forceinline__ void accumulate_plane_terms(double *acc, float x0, float y0, float z0, const float4 &bp,
                                                       const float4 &nrm) {
  float J[6], r;
  plane_terms(x0, y0, z0, bp.x, bp.y, bp.z, nrm.x, nrm.y, nrm.z, J, r);
  acc[P_VALUE] += (double)(r * r);
#pragma unroll
  for (int a = 0; a < 6; a++) acc[P_G0 + a] += (double)(J[a] * r);
  int k = 0;
#pragma unroll
  for (int a = 0; a < 6; a++)
#pragma unroll
    for (int b = a; b < 6; b++) {
      acc[P_H0 + k] += (double)(J[a] * J[b]);
      k++;
    }
  acc[P_WEIGHT] += 1.0;
  acc[P_PAIRS] += 1.0;
}

// Phase 2 of the correspondence kernels (evaluator.go:122-145): the workgroup streams the targets of
// its own chunk range in a fixed thread assignment and accumulates the evaluator's sums in float64;
// s_scratch: >= (kIcpBlock / 64) * NS doubles of LDS no longer in use.
template <bool kPlane, bool kFlagged = false>
__device__ __forceinline__ void reduce_block_range(uint32_t *s_scratch, const float *__restrict__ tx,
                                                   const float *__restrict__ ty, const float *__restrict__ tz,
                                                   int64_t nt, uint32_t chunk_begin, uint32_t chunk_end, bool project,
                                                   const float (&m)[16], const float4 *__restrict__ match,
                                                   const uint32_t *__restrict__ match_id,
                                                   const float4 *__restrict__ normals, const IcpKernelParams &kp,
                                                   double *__restrict__ block_partials,
                                                   uint32_t *__restrict__ flags = nullptr, double extra = 0.0) {
  // extra: added to component threadIdx.x of the row (threads < NS)
  // kFlagged: only the targets whose flags[] word has bit 31 set (cleared here)
  constexpr int NS = kPlane ? (int)P_COUNT : (int)S_COUNT;
  double acc[NS];
#pragma unroll
  for (int k = 0; k < NS; k++) acc[k] = 0.0;
  const int64_t r_begin = (int64_t)chunk_begin * 64;
  int64_t r_end = (int64_t)chunk_end * 64;
  if (r_end > nt) r_end = nt;
  for (int64_t i = r_begin + threadIdx.x; i < r_end; i += kIcpBlock) {
    if (kFlagged) {
      const uint32_t f = flags[i];
      if (!(f >> 31)) continue;
      flags[i] = f & 0x7fffffffu;
    }
    const float4 bp = match[i];
    if (bp.w >= 0.0f) {  // correspondence.go:27-29
      float x0 = tx[i], y0 = ty[i], z0 = tz[i];
      if (project) {  // icp.go:62-64
        float px, py, pz;
        mat4_transform(m, x0, y0, z0, px, py, pz);
        x0 = px; y0 = py; z0 = pz;
      }
      if (kPlane) {
        const float4 nrm = normals[match_id[i]];
        accumulate_plane_terms(acc, x0, y0, z0, bp, nrm);
      } else {
        accumulate_terms(acc, x0, y0, z0, bp, kp);
      }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double(*s_red)[NS] = reinterpret_cast<double(*)[NS]>(s_scratch);
#pragma unroll
  for (int k = 0; k < NS; k++) {
    double v = wave_sum_f64(acc[k]);
    if (lane == 0) s_red[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < NS) {
    double v = 0.0;
    for (int w = 0; w < kIcpBlock / 64; w++) v += s_red[w][threadIdx.x];
    block_partials[(int64_t)blockIdx.x * NS + threadIdx.x] = v + extra;
  }
}

// One ICP iteration's correspondence + reduction for a tile of targets.
//
// Phase 1, correspondence (correspondence.go:22-37 for every target at once): targets are
// stored SoA (x[], y[], z[]) in Morton order of the ORIGINAL target; every iteration
// re-projects the original by the accumulated transform (icp.go:62-64) in registers and walks
// the tree (walk_queries).  Result per target: match = {base x, y, z, DistSq}, DistSq < 0 = no
// pair.  Phase 2, reduction (evaluator.go:122-145): after a workgroup barrier the workgroup
// streams the targets of ITS OWN (static) chunk range again in a fixed thread assignment and
// accumulates the evaluator's 9 sums + the pair count in float64.  The order of every addition
// is fixed by the launch geometry alone, so the sums are bitwise reproducible although the
// walk hands queries to lanes dynamically.
// kPlane: match_id[i] additionally records the matched base id; the reduction gathers that
// point's normal (normals[id], float4 per base point in id order) and accumulates the 30 sums.
// kGrid (exact mode only): icp_grid_kernel ran before and answered every target the uniform grid
// could certify (knn_grid.h); only the rest (walk_list, this workgroup's segment, walk_count[slot]
// entries) is walked here.
constexpr int kIcpGridBlock = 256;
constexpr int kIcpStrictGridBlock = 64;  // strict sessions: no workgroup reduction in the grid pass, one wave per workgroup

template <bool kMinDist, bool kPlane, bool kGrid, bool kSums = true>
__global__ __launch_bounds__(kIcpBlock) void icp_corr_kernel(
    TreeView tv, const float *__restrict__ tx, const float *__restrict__ ty,
    const float *__restrict__ tz, int64_t nt, const IcpState *__restrict__ state,
    IcpKernelParams kp, float4 *__restrict__ match, uint32_t *__restrict__ first_leaf,
    double *__restrict__ block_partials, uint32_t *__restrict__ match_id,
    const float4 *__restrict__ normals, uint32_t *__restrict__ walk_list, uint32_t *__restrict__ walk_count,
    int32_t n_grid_rows, const uint32_t *__restrict__ orig_of = nullptr, float4 *__restrict__ match_caller = nullptr,
    StrictWork strict_w = StrictWork(), int32_t tile_sums = 0, float *__restrict__ match_cert = nullptr,
    const float *__restrict__ cert_by_id = nullptr) {
  static_assert(!(kGrid && kMinDist), "the grid answers exact-mode queries only");
  extern __shared__ uint32_t s_stack[];
  __shared__ uint32_t s_next_chunk;
  __shared__ double s_tile_part[kIcpBlock / 64][kStrictRows];
  if (kGrid && !kSums) {  // (both words in one round trip: a strict session's launch usually finds nothing to walk)
    const uint32_t left0 = walk_count[block_slot(blockIdx.x, gridDim.x)];
    const int done0 = state->done;
    if (done0 || (left0 == 0 && !tile_sums)) return;  // uniform
  }
  if (state->done) return;  // uniform
  // strict sessions (kGrid, no sums here): on its way out the workgroup forms the float64 tile sums the
  // summary kernel's guesses start from (strict_terms.h) -- the pairs are all in place (icp_grid_kernel)
  // except those of the few targets this launch still walks, and only guesses depend on the sums
  auto strict_tile_sums = [&]() {
    if (!tile_sums) return;  // uniform
    const TermSrc S = make_term_src(match_caller, nullptr, state, strict_w);
    for (int64_t tile = blockIdx.x; tile < strict_w.ntiles; tile += gridDim.x) tile_sums_block<kIcpBlock>(S, strict_w, tile, s_tile_part);
  };
  // grid mode: what this workgroup has to do is known before any of the walk's set-up -- usually
  // nothing but folding its share of icp_grid_kernel's rows
  constexpr int NS = kPlane ? (int)P_COUNT : (int)S_COUNT;
  uint32_t slot = 0, left = 0;
  double grid_part = 0.0;
  if (kGrid && !kSums) {
    slot = block_slot(blockIdx.x, gridDim.x);
    left = walk_count[slot];  // uniform
    if (left == 0) {
      strict_tile_sums();
      return;
    }
  } else if (kGrid) {
    slot = block_slot(blockIdx.x, gridDim.x);
    left = walk_count[slot];  // uniform
    // this workgroup's share of icp_grid_kernel's rows, folded into its own row in a fixed order:
    // kFold lanes per component take every kFold-th row each (their loads in flight together: one
    // lane per component would wait for 8 dependent round trips), then fold across the lanes
    constexpr int kFold = kPlane ? 16 : 32;
    static_assert(NS * kFold <= kIcpBlock, "one lane per (component, residue)");
    __shared__ double s_fold[NS];
    if (threadIdx.x < NS * kFold) {
      const int k = threadIdx.x / kFold, j = threadIdx.x % kFold;
      const int64_t g0 = (int64_t)n_grid_rows * blockIdx.x / gridDim.x, g1 = (int64_t)n_grid_rows * (blockIdx.x + 1) / gridDim.x;
      double v = 0.0;
      for (int64_t r = g0 + j; r < g1; r += kFold) v += block_partials[((int64_t)gridDim.x + r) * NS + k];
#pragma unroll
      for (int o = kFold / 2; o > 0; o >>= 1) v += __shfl_down(v, o, kFold);
      if (j == 0) s_fold[k] = v;
    }
    __syncthreads();
    grid_part = threadIdx.x < NS ? s_fold[threadIdx.x] : 0.0;
    if (left == 0) {
      if (threadIdx.x < NS) block_partials[(int64_t)blockIdx.x * NS + threadIdx.x] = grid_part;
      return;
    }
  }
  uint32_t *queue = s_stack + (size_t)(tv.depth > 1 ? tv.depth - 1 : 1) * kIcpBlock +
                    (threadIdx.x >> 6) * (kWalkQueueBytesPerWave / 4);
  float *top = reinterpret_cast<float *>(s_stack + (size_t)(tv.depth > 1 ? tv.depth - 1 : 1) * kIcpBlock +
                                         (kIcpBlock / 64) * (kWalkQueueBytesPerWave / 4));
  load_top_levels(tv, top);
  uint32_t chunk_begin, chunk_end;
  block_chunk_range(nt, blockIdx.x, gridDim.x, chunk_begin, chunk_end);
  if (threadIdx.x == 0) s_next_chunk = chunk_begin;
  __syncthreads();
  float m[16];
#pragma unroll
  for (int i = 0; i < 16; i++) m[i] = state->trans[i];
  // Before the first update targetTransformed is a plain copy (icp.go:27-30).
  const bool project = state->iter > 0;
  // From the second iteration of a Fit on, match[i] still holds the base point matched in the
  // previous iteration: its distance to the re-projected target bounds the new nearest distance
  // from above and seeds the walk's pruning bound (knn_walk.h, "Pruning bound"; exact mode only).
  // A session's match[] starts out invalid (w = NaN) and only ever holds points of its tree.
  // Likewise first_leaf[i] holds the leaf the target's first descent ended in last time (0: none):
  // the re-projected target has hardly moved, so it predicts this iteration's descent far better
  // than the grid directory does (the prediction is verified either way).
  auto load_query = [&](int64_t i, float &x, float &y, float &z, float &ub, uint32_t &pred) {
    // every load is issued before the first use: one memory round trip per chunk
    const float x0 = tx[i], y0 = ty[i], z0 = tz[i];
    float4 pm = make_float4(0.0f, 0.0f, 0.0f, -1.0f);
    pred = 0u;
    if (project) {  // uniform
      pred = first_leaf[i] & 0x7fffffffu;
      if (!kMinDist) pm = match[i];
    }
    x = x0; y = y0; z = z0;
    if (project) mat4_transform(m, x0, y0, z0, x, y, z);
    // The distance is used arithmetically whatever w says, so the compiler cannot split the
    // 16-byte load into "w first, xyz if valid" (two dependent round trips).  dm + 0 is dm
    // exactly; an invalid record gives +inf or NaN, neither tightens the bound.
    const float dx = pm.x - x, dy = pm.y - y, dz = pm.z - z;
    const float dm = (dx * dx + dy * dy) + dz * dz;
    ub = dm + (pm.w >= 0.0f ? 0.0f : __builtin_inff());
  };
  auto emit = [&](int64_t i, const float4 &bp, float best_d) {
    const float4 rec = make_float4(bp.x, bp.y, bp.z, __float_as_int(bp.w) >= 0 ? best_d : -1.0f);
    match[i] = rec;
    if (match_caller) match_caller[orig_of[i]] = rec;  // strict sums: see icp_grid_kernel
    if (kPlane) match_id[i] = __float_as_uint(bp.w);
    // (the walked partner's certificate, for the next iteration's grid pass: icp_grid_kernel)
    if (match_cert) match_cert[i] = (cert_by_id && __float_as_int(bp.w) >= 0) ? cert_by_id[__float_as_uint(bp.w)] : 0.0f;
  };
  if (kGrid) {
    // icp_grid_kernel answered what the grid could certify and summed those terms; this workgroup
    // walks what was left in its segment of walk_list (usually nothing) and sums the terms of
    // exactly those targets (flag in first_leaf[], bit 31) in the fixed order of its range
    const int64_t r_begin = (int64_t)chunk_begin * 64;
    __syncthreads();
    if (threadIdx.x == 0) {
      walk_count[slot] = 0;  // for the next iteration's grid pass
      s_next_chunk = 0;
    }
    __syncthreads();
    walk_queries<kMinDist>(
        tv, s_stack + threadIdx.x, kIcpBlock, queue, top, (int64_t)left, &s_next_chunk, (left + 63u) / 64u, 0,
        kp.max_dist_sq, kp.min_dist_sq,
        [&](int64_t j, float &x, float &y, float &z, float &ub, uint32_t &pred) {
          load_query(r_begin + walk_list[r_begin + j], x, y, z, ub, pred);
          pred &= 0x7fffffffu;
        },
        [&](int64_t j, const float4 &bp, float best_d) { emit(r_begin + walk_list[r_begin + j], bp, best_d); },
        [&](int64_t j, uint32_t leaf) { first_leaf[r_begin + walk_list[r_begin + j]] = kSums ? (leaf | 0x80000000u) : leaf; });
    if (!kSums) {
      __syncthreads();
      strict_tile_sums();
      return;
    }
    __threadfence_block();
    __syncthreads();
    reduce_block_range<kPlane, true>(s_stack, tx, ty, tz, nt, chunk_begin, chunk_end, project, m, match, match_id,
                                     normals, kp, block_partials, first_leaf, grid_part);
    return;
  } else {
    walk_queries<kMinDist>(
        tv, s_stack + threadIdx.x, kIcpBlock, queue, top, nt, &s_next_chunk, chunk_end, (int64_t)chunk_begin * 64,
        kp.max_dist_sq, kp.min_dist_sq, load_query, emit, [&](int64_t i, uint32_t leaf) { first_leaf[i] = leaf; });
  }

  if (!kSums) return;
  // ---- phase 2: this workgroup's range, fixed order
  __threadfence_block();
  __syncthreads();  // all match[] of the range are written; stacks / queues are free for reuse
  reduce_block_range<kPlane>(s_stack, tx, ty, tz, nt, chunk_begin, chunk_end, project, m, match, match_id, normals,
                             kp, block_partials);
}

// Grid pass of an iteration (exact mode): one target per lane asks the uniform grid (knn_grid.h)
// with the distance to its previous match as bound.  Certified answers go to match[] (the same
// record the walk writes) and, in the lanes' fixed order, into this workgroup's row of partial sums
// (rows n_corr_blocks .. of block_partials).  The others are flagged in first_leaf[] (bit 31) and
// queued in the walk_list segment of the icp_corr_kernel workgroup (n_corr_blocks of them) that owns
// the target; that kernel walks them and adds their terms to ITS row.
// kTrace (measurement aid): nothing is stored; trace[0..2] += targets left to the walk, point
// records read, cell-bound words read.
// kSums false (strict sessions: the sums are formed by strict.hip from match[]): the terms, their LDS
// reduction and the walk flags are left out.
template <bool kPlane, bool kTrace = false, bool kSums = true>
__global__ __launch_bounds__(kIcpGridBlock) void icp_grid_kernel(
    GridView grid, const float *__restrict__ tx, const float *__restrict__ ty, const float *__restrict__ tz, int64_t nt,
    const IcpState *__restrict__ state, IcpKernelParams kp, float4 *__restrict__ match,
    uint32_t *__restrict__ match_id, const float4 *__restrict__ normals, uint32_t *__restrict__ first_leaf,
    uint32_t *__restrict__ walk_list, uint32_t *__restrict__ walk_count, uint32_t n_corr_blocks,
    double *__restrict__ block_partials, unsigned long long *__restrict__ trace = nullptr,
    const uint32_t *__restrict__ orig_of = nullptr, float4 *__restrict__ match_caller = nullptr,
    float *__restrict__ match_cert = nullptr, int caller_has_pairs = 0, int walk_follows = 1, int test_force_walk = 0) {
  constexpr int NS = kPlane ? (int)P_COUNT : (int)S_COUNT;
  __shared__ float s_terms[NS][kIcpGridBlock + 16];  // + 16: the kSub-lane groups of one wave land on different banks
  // ("done" is looked at behind the target's loads, which it would only hold up: a workgroup of this kernel is five
  // dependent round trips long, this was one of them; nothing is written before)
  const int done = state->done;
  // targets are stored in cell order: an XCD takes a contiguous eighth of them (pcgx_internal.h, xcd_tile)
  // (without the sums a workgroup may be smaller than kIcpGridBlock: kIcpStrictGridBlock)
  const uint32_t block = kSums ? (uint32_t)kIcpGridBlock : blockDim.x;
  const uint32_t n_tiles = (uint32_t)((nt + block - 1) / block);
  const uint32_t tile = xcd_tile(blockIdx.x, n_tiles);
  if (tile >= n_tiles) return;  // uniform
  const int64_t i = (int64_t)tile * block + threadIdx.x;
  double acc[NS];
#pragma unroll
  for (int k = 0; k < NS; k++) acc[k] = 0.0;
  const bool project = state->iter > 0;  // icp.go:27-30
  float x = 0.0f, y = 0.0f, z = 0.0f;
  float4 pm = make_float4(0.0f, 0.0f, 0.0f, -1.0f);
  float pm_cert = 0.0f;
  const bool use_cert = !kTrace && match_cert != nullptr && grid.cert != nullptr;
  if (i < nt) {
    x = tx[i];
    y = ty[i];
    z = tz[i];
    if (project) {
      pm = match[i];  // previous iteration's pair (w = NaN before the first one)
      if (use_cert) pm_cert = match_cert[i];  // ... and its partner's certificate (0: none)
    }
  }
  if (done) return;  // uniform
  if (i < nt) {
    float ub = __builtin_inff();
    bool kept = false;  // last iteration's partner is this one's, by its certificate
    float dm = 0.0f;
    if (project) {
      const float *m = state->trans;
      float px, py, pz;
      mat4_transform(m, x, y, z, px, py, pz);
      x = px; y = py; z = pz;
      const float dx = pm.x - x, dy = pm.y - y, dz = pm.z - z;
      dm = (dx * dx + dy * dy) + dz * dz;  // (the scan's expression for this point: knn_grid.h)
      if (pm.w >= 0.0f && dm == dm) ub = dm;
      // Every other base point is strictly farther from the moved target than last iteration's partner when its DistSq
      // is below the partner's certificate (knn_grid.hip, grid_cert_kernel): Nearest returns that point again, with
      // this distance -- no search.  (C4: 54 % of the targets in a Fit's second iteration, 98.5 % in its third, 99.9 %
      // from the tenth on, tools/cert_probe.py; the search is 5.8 scattered loads per target, this is none.)
      kept = pm.w >= 0.0f && dm < pm_cert && dm < kp.max_dist_sq;
      if (test_force_walk && i % test_force_walk == 0) kept = false;  // (tests: this target is searched for and handed to the walk below)
      if (kTrace && match_cert != nullptr && grid.cert != nullptr && pm.w >= 0.0f && dm < match_cert[i] && dm < kp.max_dist_sq)
        atomicAdd(&trace[36], 1ull);  // (the trace counts what a certificate would keep and searches all the same)
    }
    float4 best = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(-1));
    float best_d = kp.max_dist_sq;
    GridTrace tr;
    GridVerdict v = GRID_FOUND;
    if (kept) {
      best = make_float4(pm.x, pm.y, pm.z, __int_as_float(0));  // (the id is not looked at below: match_id[] keeps it)
      best_d = dm;
    } else {
      v = grid_nearest(grid, x, y, z, kp.max_dist_sq, ub, best, best_d, kTrace ? &tr : nullptr);
      if (test_force_walk && project && i % test_force_walk == 0) v = GRID_WALK;  // (tests: targets for the walk where the grid leaves none)
    }
    if (use_cert && !kept)  // the new partner's certificate (none: not found, left to the walk)
      match_cert[i] = (v == GRID_FOUND && __float_as_int(best.w) >= 0) ? grid.cert[__float_as_uint(best.w)] : 0.0f;
    if (kTrace) {
      if (v == GRID_WALK) atomicAdd(&trace[0], 1ull);
      atomicAdd(&trace[1], (unsigned long long)tr.points);
      atomicAdd(&trace[2], (unsigned long long)tr.words);
      if (tr.wave_slots) atomicAdd(&trace[3], (unsigned long long)tr.wave_slots);
      atomicAdd(&trace[4 + (tr.rounds4 < 0 ? 0 : min(tr.rounds4 + 1, 15))], 1ull);
      atomicAdd(&trace[20 + (tr.rounds9 < 0 ? 0 : min(tr.rounds9 + 1, 15))], 1ull);
    } else if (v == GRID_WALK) {
      uint32_t begin;
      const uint32_t slot = slot_of_query(nt, i, n_corr_blocks, begin);
      const int64_t r_begin = (int64_t)begin * 64;
      walk_list[r_begin + atomicAdd(&walk_count[slot], 1u)] = (uint32_t)(i - r_begin);
      first_leaf[i] = kSums ? 0x80000000u : 0u;  // "walked this iteration" (icp_corr_kernel adds its terms)
      // No walk was launched behind this pass (enqueue_corr: it finds nothing to do in iteration after iteration, and
      // the host cannot know): the step ends here -- `done` 2 makes every kernel behind this one return, this step's
      // and the later ones' -- and the host enqueues it again, with the walk, when it next looks (settle()).
      if (!walk_follows) const_cast<IcpState *>(state)->done = 2;
      // strict sums: the float64 tile sums (guesses only) are formed by icp_corr_kernel's workgroups while others
      // of them still walk; until the walk's answer arrives the target stands with the best point the grid has
      // seen -- usually the answer -- instead of last iteration's pair (none at all in a Fit's first iteration)
      if (!kSums && match_caller) {
        const bool seen = __float_as_int(best.w) >= 0;
        match_caller[orig_of[i]] = make_float4(best.x, best.y, best.z, seen ? best_d : -1.0f);
      }
    } else {
      const bool found = __float_as_int(best.w) >= 0;
      const float4 bp = make_float4(best.x, best.y, best.z, found ? best_d : -1.0f);
      // (a kept pair: the point is in place, only the distance is new -- 4 bytes of each record instead of 16)
      if (kept) reinterpret_cast<float *>(&match[i])[3] = bp.w;
      else match[i] = bp;
      // strict sums add the pairs' terms in the CALLER's target order: the pair goes there as well (a
      // 16-byte scatter here, +6 us at C4, instead of a gather through pos_of in the terms kernel, 10 us)
      if (!kSums && match_caller && !(kept && caller_has_pairs)) match_caller[orig_of[i]] = bp;  // (a kept pair is there, and its reader forms the distance from the points: strict_terms.h, pair_terms)
      if (kPlane && !kept) match_id[i] = __float_as_uint(best.w);
      if (found && kSums) {  // correspondence.go:27-29
        if (kPlane) accumulate_plane_terms(acc, x, y, z, bp, normals[kept ? match_id[i] : __float_as_uint(best.w)]);
        else accumulate_terms(acc, x, y, z, bp, kp);
      }
    }
  }
  if (kTrace || !kSums) return;  // uniform
  // every lane holds the float32 terms of (at most) one pair: through LDS, component k is added
  // up by kSub lanes, each over every kSub-th target, then across those lanes -- a fixed order
  constexpr int kSub = kPlane ? 8 : 16, kRun = kIcpGridBlock / kSub;
  static_assert(NS * kSub <= kIcpGridBlock, "one lane per (component, run)");
#pragma unroll
  for (int k = 0; k < NS; k++) s_terms[k][threadIdx.x] = (float)acc[k];  // exact: acc[k] is one float32 term or 0
  __syncthreads();
  if (threadIdx.x < NS * kSub) {
    const int k = threadIdx.x / kSub, j = threadIdx.x % kSub;
    double v = 0.0;
    for (int u = 0; u < kRun; u++) v += (double)s_terms[k][u * kSub + j];  // lanes j side by side: no bank conflicts
#pragma unroll
    for (int o = kSub / 2; o > 0; o >>= 1) v += __shfl_down(v, o, kSub);
    if (j == 0) block_partials[((int64_t)n_corr_blocks + tile) * NS + k] = v;
  }
}

// The same iteration on a base handle that has seen DeletePoint: correspondence by the reference's
// own walk of its patched tree (knn_xwalk.h; one target per lane, static assignment, no hints from
// the previous iteration -- the tree may have changed in between), then the same phase 2.
template <bool kMinDist, bool kPlane, bool kSums = true>
__global__ __launch_bounds__(kIcpBlock) void icp_corr_xkernel(
    XTreeView xv, const float *__restrict__ tx, const float *__restrict__ ty, const float *__restrict__ tz,
    int64_t nt, const IcpState *__restrict__ state, IcpKernelParams kp, float4 *__restrict__ match,
    double *__restrict__ block_partials, uint32_t *__restrict__ match_id, const float4 *__restrict__ normals,
    int64_t guard) {
  extern __shared__ uint32_t s_stack[];
  if (state->done) return;  // uniform
  uint32_t chunk_begin, chunk_end;
  block_chunk_range(nt, blockIdx.x, gridDim.x, chunk_begin, chunk_end);
  float m[16];
#pragma unroll
  for (int i = 0; i < 16; i++) m[i] = state->trans[i];
  const bool project = state->iter > 0;
  const int64_t r_begin = (int64_t)chunk_begin * 64;
  int64_t r_end = (int64_t)chunk_end * 64;
  if (r_end > nt) r_end = nt;
  for (int64_t i = r_begin + threadIdx.x; i < r_end; i += kIcpBlock) {
    float qx = tx[i], qy = ty[i], qz = tz[i];
    if (project) {
      float px, py, pz;
      mat4_transform(m, qx, qy, qz, px, py, pz);
      qx = px; qy = py; qz = pz;
    }
    float4 best = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(-1));
    float best_d = kp.max_dist_sq;
    xwalk(
        xv, s_stack + threadIdx.x, kIcpBlock, qx, qy, qz, guard, [&]() { return best_d; },
        [&](const float4 &nd, float d) {  // kdtree.go:95-106
          if (!(d > best_d)) {
            best = nd;
            best_d = d;
          }
          return !(kMinDist && best_d < kp.min_dist_sq);
        },
        [&](const float4 &nd, float d) {  // kdtree.go:116-123
          if (d < best_d) {
            best = nd;
            best_d = d;
            if (kMinDist && best_d < kp.min_dist_sq) return false;
          }
          return true;
        });
    match[i] = make_float4(best.x, best.y, best.z, __float_as_int(best.w) >= 0 ? best_d : -1.0f);
    if (kPlane) match_id[i] = __float_as_uint(best.w);
  }
  if (!kSums) return;
  __threadfence_block();
  __syncthreads();
  reduce_block_range<kPlane>(s_stack, tx, ty, tz, nt, chunk_begin, chunk_end, project, m, match, match_id, normals,
                             kp, block_partials);
}

// Plane sessions: evaluate tail (finish_evaluate_plane) + Gauss-Newton update; one thread.
__device__ __forceinline__ void icp_plane_update_step(IcpState *__restrict__ state, const double *__restrict__ sums30,
                                                      const IcpKernelParams &kp) {
  state->num_iteration += 1;
  const int64_t npairs = (int64_t)sums30[P_PAIRS];
  if (npairs < (int64_t)kp.min_pairs) {
    state->ev.num_pairs = npairs;
    state->status = PCGX_E_NOT_ENOUGH_PAIRS;
    state->done = 1;
    return;
  }
  EvaluatedPlane ev;
  finish_evaluate_plane(sums30, ev);
  state->ev.value = ev.value;
  for (int i = 0; i < 6; i++) state->ev.gradient[i] = ev.gradient[i];
  state->ev.dist_rms = 0.0f;
  state->ev.num_pairs = ev.num_pairs;
  for (int i = 0; i < 36; i++) state->hessian[i] = ev.hessian[i];
  Mat4 t;
  for (int i = 0; i < 16; i++) t.m[i] = state->trans[i];
  int32_t it = state->iter;
  const int rc = gauss_newton_update(kp.gn, it, ev, t);
  if (rc < 0) {
    state->status = PCGX_E_SINGULAR;
    state->done = 1;
    return;
  }
  for (int i = 0; i < 16; i++) state->trans[i] = t.m[i];
  state->iter = it;
  if (rc > 0) state->done = 1;
}

// Sums the per-workgroup partials in a fixed order -> sums: wave w owns components w, w + waves, ...
// (launched with one wave per component for the reference's 10 sums).  With kFuseUpdate (single
// GPU: no exchange between reduce and update) thread 0 then runs the evaluate tail + pose update
// in the same launch.
template <bool kFuseUpdate, bool kPlane>
__global__ __launch_bounds__(1024) void icp_final_reduce_kernel(const double *__restrict__ block_partials,
                                                                int nblocks, IcpState *__restrict__ state,
                                                                double *__restrict__ sums,
                                                                IcpKernelParams kp) {
  constexpr int NS = kPlane ? (int)P_COUNT : (int)S_COUNT;
  __shared__ double s_sums[NS];
  if (state->done) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, waves = blockDim.x >> 6;
  for (int k = wave; k < NS; k += waves) {
    double v = 0.0;
    for (int b = lane; b < nblocks; b += 64) v += block_partials[(int64_t)b * NS + k];
    v = wave_sum_f64(v);
    if (lane == 0) {
      sums[k] = v;
      s_sums[k] = v;
    }
  }
  if (kFuseUpdate) {
    __syncthreads();
    if (threadIdx.x == 0) {
      if (kPlane) icp_plane_update_step(state, s_sums, kp);
      else icp_update_step(state, s_sums, kp);
    }
  }
}

template <bool kPlane>
__global__ void icp_update_kernel(IcpState *__restrict__ state, const double *__restrict__ sums,
                                  IcpKernelParams kp) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (state->done) return;
  if (kPlane) icp_plane_update_step(state, sums, kp);
  else icp_update_step(state, sums, kp);
}

// The update behind a sharded step's all-reduce: xchg = the summed sums + the ranks' error flag behind them.  A flag
// that is up ends the Fit on every rank in this same iteration (state->status PCGX_E_RCCL); the sums also go to the
// session's own buffer (pcgx_icp_session_read_sums).
template <bool kPlane>
__global__ void icp_update_sharded_kernel(IcpState *__restrict__ state, const double *__restrict__ xchg, int n,
                                          double *__restrict__ sums_out, IcpKernelParams kp) {
  if (blockIdx.x != 0) return;
  if (state->done) return;
  if ((int)threadIdx.x < n && sums_out != xchg) sums_out[threadIdx.x] = xchg[threadIdx.x];
  if (threadIdx.x != 0) return;
  if (xchg[n] != 0.0) {
    state->status = PCGX_E_RCCL;
    state->done = 1;
    return;
  }
  if (kPlane) icp_plane_update_step(state, xchg, kp);
  else icp_update_step(state, xchg, kp);
}

// normals (packed xyz, base id order) -> float4 per base point
__global__ __launch_bounds__(256) void pack_normals_kernel(const float *__restrict__ n3, int64_t n,
                                                           float4 *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = make_float4(n3[3 * i], n3[3 * i + 1], n3[3 * i + 2], 0.0f);
}

__global__ __launch_bounds__(256) void invert_positions_kernel(const uint32_t *__restrict__ pos_of, int64_t n,
                                                               uint32_t *__restrict__ orig_of) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) orig_of[pos_of[i]] = (uint32_t)i;
}

__global__ __launch_bounds__(256) void gather_soa_kernel(const float *__restrict__ q,
                                                         const int32_t *__restrict__ perm, int64_t n,
                                                         float *__restrict__ x, float *__restrict__ y,
                                                         float *__restrict__ z, uint32_t *__restrict__ pos_of) {
  const int64_t pos = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (pos >= n) return;
  const int64_t i = perm ? (int64_t)perm[pos] : pos;
  pos_of[i] = (uint32_t)pos;  // where the caller's target i sits in the session's (Morton) order
  x[pos] = q[3 * i];
  y[pos] = q[3 * i + 1];
  z[pos] = q[3 * i + 2];
}

}  // namespace pcgx

using namespace pcgx;

struct pcgx_icp_session {
  const pcgx_kdtree *base = nullptr;
  bool patched = false;  // base had deletions at creation: walk its patched explicit tree (icp_corr_xkernel)
  int64_t nt = 0;
  float *d_xyz = nullptr;  // SoA: x[nt] | y[nt] | z[nt], Morton order of the original target
  IcpState *d_state = nullptr;
  double *d_partials = nullptr;
  uint32_t *d_pos_of = nullptr;  // [nt] position of the caller's target i in the session's order
  uint32_t *d_orig_of = nullptr;      // strict sums: [nt] the caller's index of the target at a position
  float4 *d_match_caller = nullptr;   // strict sums: match[] in the caller's target order
  bool caller_order_fresh = false;    // the last correspondence pass wrote d_match_caller as well
  bool tile_sums_fresh = false;       // ... and formed the strict sums' float64 tile sums on its way out
  int strict = 0;                // sequential float32 sums: 1 = in parallel (strict.hip), 2 = one wave (icp_strict_sums_kernel)
  bool strict_explicit = false;  // asked for by name (set_strict, PCGX_SUMS_REFERENCE_CHAIN, the environment): a sharded step refuses
                                 // it; the default (sums_mode 0) quietly becomes float64 sums there
  pcgx::StrictBuffers *strict_buf = nullptr;  // strict 1
  float *d_terms = nullptr;      // strict 2: [9][nt_pad] float32 terms in the caller's target order
  unsigned long long *d_valid = nullptr;  // strict: [nt_pad / 64] matched-target bits
  int64_t nt_pad = 0;
  float4 *d_match = nullptr;       // [nt] matched base point + DistSq per target
  float *d_match_cert = nullptr;   // [nt] the matched point's certificate (GridView::cert; 0: none): icp_grid_kernel
  uint32_t *d_first_leaf = nullptr;  // [nt] leaf the target's first descent ended in (0: unknown)
  uint32_t *d_walk_list = nullptr;   // [nt] per workgroup segment: targets the grid pass left to the walk
  uint32_t *d_walk_count = nullptr;  // [grid] entries in each segment (zero between iterations)
  double *d_sums = nullptr;  // caller's buffer, or own
  bool own_sums = false;
  double *d_xchg = nullptr;  // sharded float64 steps: the sums + the ranks' error flag, what the all-reduce carries
  int32_t steps_sharded = 0; // sharded steps enqueued (fault injection of the tests counts them)
  int32_t host_iter = 0;     // Evaluates enqueued since the device's loop state was last WRITTEN (session made, reset, set_pose):
                             // what settle() compares with the device's num_iteration, which those writes zero -- not the updater's
                             // `iter`, which set_pose may start anywhere
  bool spec_walk = true;     // the leftover walk is not launched behind a grid pass from a Fit's second Evaluate on (enqueue_corr)
  bool spec_pending = false; // ... and steps enqueued that way have not been looked at yet (settle())
  hipStream_t spec_stream = nullptr;  // ... on this stream (entry points without a stream argument settle there)
  hipStream_t used[4] = {nullptr, nullptr, nullptr, nullptr};  // the streams work on this session's buffers was enqueued on
  int n_used = 0;                                               // (5: more than four -- pcgx_icp_session_free waits for the device)
  void touch(hipStream_t st) {
    for (int k = 0; k < n_used && k < 4; k++)
      if (used[k] == st) return;
    // a stream the session has not been used on: the set-up (upload, order, gather: enqueued on used[0], not waited
    // for) must be through before work there reads the session's buffers
    if (n_used > 0) (void)hipStreamSynchronize(used[0]);
    if (n_used < 4) used[n_used] = st;
    n_used++;
  }
  bool shard_failed = false; // this rank could not go on: it keeps calling the collectives with its flag up
  bool ring_fit_open = false; // pcgx_icp_fit_sharded has told the communicator that a Fit begins (comm_ring_new_fit: once per Fit)
  bool general_ready = true;       // d_match / d_match_cert / d_first_leaf / d_walk_count hold their start values (a small session: at its first step outside the one launch)
  bool small = false;              // both clouds small: a step, or a whole Fit, is ONE launch (icp_small.hip); the target stays in the caller's order
  int32_t *d_small_perm = nullptr; // ... its order of the targets (position -> caller's index; nullptr: the caller's order)
  void *d_small_sync = nullptr;    // ... that launch's barrier words (zero between launches)
  bool plane = false;              // point-to-plane / Gauss-Newton session (30 sums)
  uint32_t *d_match_id = nullptr;  // plane: [nt] matched base id
  float4 *d_normals = nullptr;     // plane: [base n] unit normals in base id order
  int n_sums() const { return plane ? (int)P_COUNT : (int)S_COUNT; }
  int grid = 1;
  IcpKernelParams kp;
  int32_t max_iteration = 20;
};

static pcgx_status settle(pcgx_icp_session *s, hipStream_t st);
static int icp_knob(const char *name, int def, int lo, int hi);

static IcpKernelParams make_kernel_params(const pcgx_icp_params *p) {
  IcpKernelParams kp;
  memset(&kp, 0, sizeof kp);
  kp.max_dist_sq = p->max_dist * p->max_dist;  // kdtree.go:91 via correspondence.go:26
  kp.min_dist_sq = p->min_dist_sq;
  kp.min_pairs = p->min_pairs == 0 ? 6 : p->min_pairs;  // evaluator.go:92-95
  kp.upd = resolve_updater(p->weight, p->threshold, p->max_iteration);
  kp.gn = resolve_gauss_newton(p->threshold, 0.0f, p->max_iteration);
  kp.weight_fn = p->weight_fn;
  kp.weight_a = p->weight_fn_param;
  return kp;
}

static int icp_grid(int64_t nt, const TreeView &tv) {
  int64_t blocks = (nt + kIcpBlock - 1) / kIcpBlock;
  int64_t cap = (int64_t)ctx().num_cu * walk_blocks_per_cu(tv) * walk_oversubscribe();
  if (blocks > cap) blocks = cap;
  if (blocks >= 8) blocks &= ~(int64_t)7;  // multiple of 8: see block_chunk_range
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

// Loop state of a fresh Fit (icp.go:47): identity transform, counters zero.  On the device, in
// stream order: a reset between two Fits costs a launch, not a host synchronisation.
__global__ void icp_reset_kernel(IcpState *__restrict__ state) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  IcpState h;
  memset(&h, 0, sizeof h);
  const Mat4 id = mat4_translate(0.0f, 0.0f, 0.0f);
  for (int i = 0; i < 16; i++) h.trans[i] = id.m[i];
  *state = h;
}

// The general path's per-target arrays at their start values: no previous match yet (w = NaN: icp_corr_kernel takes pruning
// hints from match[] only when w >= 0), no certificate, no first leaf, empty walk lists.  A session made for the one
// launch (icp_small.hip) has not paid for this when it is made: four launches in front of a 0.16 ms Fit.
static pcgx_status general_prepare(pcgx_icp_session *s, hipStream_t st) {
  if (s->general_ready) return PCGX_OK;
  const size_t n1 = (size_t)(s->nt ? s->nt : 1);
  PCGX_HIP_TRY(hipMemsetAsync(s->d_match, 0xFF, n1 * sizeof(float4), st));
  PCGX_HIP_TRY(hipMemsetAsync(s->d_match_cert, 0, n1 * sizeof(float), st));
  PCGX_HIP_TRY(hipMemsetAsync(s->d_first_leaf, 0, n1 * sizeof(uint32_t), st));
  PCGX_HIP_TRY(hipMemsetAsync(s->d_walk_count, 0, (size_t)s->grid * sizeof(uint32_t), st));
  s->general_ready = true;
  return PCGX_OK;
}

static pcgx_status reset_state(pcgx_icp_session *s, hipStream_t st) {
  hipLaunchKernelGGL(icp_reset_kernel, dim3(1), dim3(64), 0, st, s->d_state);
  PCGX_HIP_TRY(hipGetLastError());
  s->host_iter = 0;
  if (s->spec_pending) {  // (steps enqueued without the leftover walk and never looked at: whatever they left in the lists)
    s->spec_pending = false;
    if (s->d_walk_count) PCGX_HIP_TRY(hipMemsetAsync(s->d_walk_count, 0, (size_t)s->grid * sizeof(uint32_t), st));
  }
  if (s->shard_failed || s->steps_sharded > 0) {  // (a sharded Fit may have ended inside a launch: its counters)
    PCGX_TRY(strict_reset(s->strict_buf, st));
    s->shard_failed = false;
    s->steps_sharded = 0;
  }
  return PCGX_OK;
}

extern "C" pcgx_status pcgx_icp_session_reset(pcgx_icp_session *s, void *stream) {
  PCGX_API_LOCK();
  if (!s) return fail(PCGX_E_INVALID, "pcgx_icp_session_reset: NULL session");
  s->touch(pick_stream(stream));
  return reset_state(s, pick_stream(stream));
}

extern "C" pcgx_status pcgx_icp_session_set_pose(pcgx_icp_session *s, const float trans16[16],
                                                 int32_t iter, void *stream) {
  PCGX_API_LOCK();
  if (!s || !trans16 || iter < 0) return fail(PCGX_E_INVALID, "pcgx_icp_session_set_pose: bad argument");
  hipStream_t st = pick_stream(stream);
  s->touch(st);
  PCGX_TRY(settle(s, st));  // (steps enqueued without the leftover walk: enqueue_corr)
  IcpState h;
  memset(&h, 0, sizeof h);
  memcpy(h.trans, trans16, sizeof h.trans);
  h.iter = iter;
  s->host_iter = 0;  // (num_iteration restarts at 0 with the state written below; the next step is not speculated on: enqueue_corr)
  PCGX_HIP_TRY(hipMemcpyAsync(s->d_state, &h, sizeof h, hipMemcpyHostToDevice, st));
  PCGX_HIP_TRY(hipStreamSynchronize(st));
  return PCGX_OK;
}

extern "C" pcgx_status pcgx_icp_session_read_sums(pcgx_icp_session *s, double sums10[10], void *stream) {
  PCGX_API_LOCK();
  if (!s || !sums10) return fail(PCGX_E_INVALID, "pcgx_icp_session_read_sums: bad argument");
  if (s->plane) return fail(PCGX_E_INVALID, "pcgx_icp_session_read_sums: plane session (30 sums): use pcgx_icp_session_read_sums_n");
  hipStream_t st = pick_stream(stream);
  s->touch(st);
  PCGX_TRY(settle(s, st));  // (steps enqueued without the leftover walk: enqueue_corr)
  PCGX_HIP_TRY(hipMemcpyAsync(sums10, s->d_sums, S_COUNT * sizeof(double), hipMemcpyDeviceToHost, st));
  PCGX_HIP_TRY(hipStreamSynchronize(st));
  return PCGX_OK;
}

extern "C" pcgx_status pcgx_icp_session_set_strict(pcgx_icp_session *s, int32_t on) {
  PCGX_API_LOCK();
  if (!s) return fail(PCGX_E_INVALID, "pcgx_icp_session_set_strict: NULL session");
  if (on && s->plane) return fail(PCGX_E_INVALID, "pcgx_icp_session_set_strict: point-to-plane sessions have no reference sums to reproduce");
  if (on < 0 || on > 2) return fail(PCGX_E_INVALID, "pcgx_icp_session_set_strict: mode must be 0, 1 or 2");
  // steps enqueued without the leftover walk are replayed by settle() in the mode they were asked for in: before it changes
  if (s->spec_pending) PCGX_TRY(settle(s, s->spec_stream));
  s->strict = on;
  s->strict_explicit = on != 0;
  return PCGX_OK;
}

extern "C" pcgx_status pcgx_icp_session_sums_count(const pcgx_icp_session *s, int32_t *count) {
  PCGX_API_LOCK();
  if (!s || !count) return fail(PCGX_E_INVALID, "pcgx_icp_session_sums_count: bad argument");
  *count = s->n_sums();
  return PCGX_OK;
}

extern "C" pcgx_status pcgx_icp_session_read_sums_n(pcgx_icp_session *s, double *sums, int32_t cap, void *stream) {
  PCGX_API_LOCK();
  if (!s || !sums || cap < s->n_sums()) return fail(PCGX_E_INVALID, "pcgx_icp_session_read_sums_n: bad argument");
  hipStream_t st = pick_stream(stream);
  s->touch(st);
  PCGX_TRY(settle(s, st));  // (steps enqueued without the leftover walk: enqueue_corr)
  PCGX_HIP_TRY(hipMemcpyAsync(sums, s->d_sums, (size_t)s->n_sums() * sizeof(double), hipMemcpyDeviceToHost, st));
  PCGX_HIP_TRY(hipStreamSynchronize(st));
  return PCGX_OK;
}

extern "C" pcgx_status pcgx_icp_session_free(pcgx_icp_session *s) {
  PCGX_API_LOCK();
  if (!s) return PCGX_OK;
  if (s->base) const_cast<pcgx_kdtree *>(s->base)->sessions.fetch_sub(1);
  // the buffers go back to the block cache and may be handed out again at once: work enqueued on them must have
  // finished -- on the streams this session was used on (a session knows them: every entry point names its stream),
  // not on the whole device: that wait ended only when every OTHER context's Fit had drained too
  if (s->n_used > 4) (void)hipDeviceSynchronize();
  else
    for (int k = 0; k < s->n_used; k++) (void)hipStreamSynchronize(s->used[k]);
  dev_cache_free(s->d_xyz);
  dev_cache_free(s->d_state);
  dev_cache_free(s->d_partials);
  dev_cache_free(s->d_pos_of);
  dev_cache_free(s->d_orig_of);
  dev_cache_free(s->d_match_caller);
  dev_cache_free(s->d_terms);
  dev_cache_free(s->d_valid);
  strict_destroy(s->strict_buf);
  dev_cache_free(s->d_match);
  dev_cache_free(s->d_match_cert);
  dev_cache_free(s->d_first_leaf);
  dev_cache_free(s->d_walk_list);
  dev_cache_free(s->d_walk_count);
  dev_cache_free(s->d_match_id);
  dev_cache_free(s->d_normals);
  if (s->own_sums) dev_cache_free(s->d_sums);
  dev_cache_free(s->d_xchg);
  dev_cache_free(s->d_small_sync);
  dev_cache_free(s->d_small_perm);
  delete s;
  return PCGX_OK;
}

// normals == nullptr: the reference's point-to-point session; else a plane session (normals:
// packed xyz per base point in id order, host or device memory like the target).
static pcgx_status session_create(const pcgx_kdtree *base, const float *normals, float damping,
                                  const float *target, int64_t nt, int32_t target_on_device,
                                  const pcgx_icp_params *params, double *d_sums,
                                  pcgx_icp_session **out) {
  if (!out) return fail(PCGX_E_INVALID, "pcgx_icp_session_create: out is NULL");
  *out = nullptr;
  if (!base || !params || nt < 0 || (nt > 0 && !target))
    return fail(PCGX_E_INVALID, "pcgx_icp_session_create: bad argument");
  if (params->weight_fn < 0 || params->weight_fn >= PCGX_WEIGHT_KINDS)
    return fail(PCGX_E_INVALID, "pcgx_icp_session_create: weight_fn %d is none of the built-in forms (a custom Go closure "
                                "cannot run on the device)", params->weight_fn);
  if (normals && params->weight_fn != PCGX_WEIGHT_ONE)
    return fail(PCGX_E_INVALID, "the point-to-plane extension takes the default weight only");
  if (params->sums_mode < 0 || params->sums_mode >= PCGX_SUMS_KINDS)
    return fail(PCGX_E_INVALID, "pcgx_icp_session_create: sums_mode %d is none of PCGX_SUMS_*", params->sums_mode);
  PCGX_TRY(ensure_init());
  hipStream_t st = ctx().stream;
  const int64_t n_base_ids = base->n;  // normals are indexed by the original ids
  // after DeletePoint the handle holds the reference's patched tree: the session walks that one
  // (knn_explicit.hip); no node left: Pairs() finds nothing (correspondence.go:27-29, kdtree.go:84-86)
  const bool patched = base->n_deleted > 0;
  if (patched && base->n_deleted >= base->n)
    return fail(PCGX_E_NOT_ENOUGH_PAIRS, "not enough correspondence pairs (every base point was deleted)");
  static const bool trace = getenv("PCGX_FIT_TRACE") != nullptr;  // (where a session's set-up time goes, per phase)
  auto now_us = []() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_phase[6] = {0, 0, 0, 0, 0, 0};
  t_phase[0] = trace ? now_us() : 0.0;
  pcgx_icp_session *s = new pcgx_icp_session();
  s->base = base;
  s->touch(st);
  s->patched = patched;
  const_cast<pcgx_kdtree *>(base)->sessions.fetch_add(1);
  s->nt = nt;
  s->plane = normals != nullptr;
  // the reference's own sums unless the caller asks otherwise (include/pcgx.h, PCGX_SUMS_*); the
  // point-to-plane extension has no reference sums to reproduce
  s->strict = s->plane ? 0 : (params->sums_mode == PCGX_SUMS_REFERENCE ? 1 : (params->sums_mode == PCGX_SUMS_F64_TREE ? 0 : 2));
  s->strict_explicit = s->strict == 2;
  if (const char *e = getenv("PCGX_ICP_STRICT")) {  // experiments: overrides sums_mode
    s->strict = s->plane ? 0 : (e[0] == '1' ? 1 : (e[0] == '2' ? 2 : 0));
    s->strict_explicit = s->strict != 0;
  }
  s->kp = make_kernel_params(params);
  s->kp.gn.damping = damping;
  s->max_iteration = s->kp.upd.max_iteration;
  s->grid = icp_grid(nt, base->view());
  pcgx_status rc = PCGX_OK;
  auto bail = [&](pcgx_status code) {
    pcgx_icp_session_free(s);
    return code;
  };
  hipError_t e;
  if ((e = dev_cache_alloc((void **)&s->d_xyz, (size_t)(nt ? nt : 1) * 12)) != hipSuccess ||
      (e = dev_cache_alloc((void **)&s->d_state, sizeof(IcpState))) != hipSuccess ||
      (e = dev_cache_alloc((void **)&s->d_partials,
                           ((size_t)s->grid + (size_t)(nt / kIcpGridBlock) + 1) * s->n_sums() * sizeof(double))) != hipSuccess ||
      (e = dev_cache_alloc((void **)&s->d_pos_of, (size_t)(nt ? nt : 1) * sizeof(uint32_t))) != hipSuccess ||
      (e = dev_cache_alloc((void **)&s->d_match, (size_t)(nt ? nt : 1) * sizeof(float4))) != hipSuccess ||
      (e = dev_cache_alloc((void **)&s->d_match_cert, (size_t)(nt ? nt : 1) * sizeof(float))) != hipSuccess ||
      (e = dev_cache_alloc((void **)&s->d_first_leaf, (size_t)(nt ? nt : 1) * sizeof(uint32_t))) != hipSuccess ||
      (e = dev_cache_alloc((void **)&s->d_walk_list, (size_t)(nt ? nt : 1) * sizeof(uint32_t))) != hipSuccess ||
      (e = dev_cache_alloc((void **)&s->d_walk_count, (size_t)s->grid * sizeof(uint32_t))) != hipSuccess)
    return bail(fail(PCGX_E_OOM, "icp session allocation failed: %s", hipGetErrorString(e)));
  if (d_sums) {
    s->d_sums = d_sums;
  } else {
    if ((e = dev_cache_alloc((void **)&s->d_sums, (size_t)s->n_sums() * sizeof(double))) != hipSuccess)
      return bail(fail(PCGX_E_OOM, "icp session allocation failed: %s", hipGetErrorString(e)));
    s->own_sums = true;
  }
  if (s->plane) {
    const int64_t nb = n_base_ids;
    if ((e = dev_cache_alloc((void **)&s->d_match_id, (size_t)(nt ? nt : 1) * sizeof(uint32_t))) != hipSuccess ||
        (e = dev_cache_alloc((void **)&s->d_normals, (size_t)nb * sizeof(float4))) != hipSuccess)
      return bail(fail(PCGX_E_OOM, "icp session allocation failed: %s", hipGetErrorString(e)));
    Arena &ar = ctx().arena;
    if ((rc = ar.begin(st)) != PCGX_OK) return bail(rc);
    const float *d_n3 = normals;
    if (!target_on_device) {
      float *stage = nullptr;
      if ((rc = ar.alloc_n((size_t)nb * 3, &stage)) != PCGX_OK) return bail(rc);
      if ((rc = staged_upload(stage, normals, (size_t)nb * 12, st)) != PCGX_OK) return bail(rc);
      d_n3 = stage;
    }
    hipLaunchKernelGGL(pack_normals_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, st, d_n3, nb,
                       s->d_normals);
    if ((e = hipStreamSynchronize(st)) != hipSuccess)
      return bail(fail(PCGX_E_HIP, "icp session setup failed: %s", hipGetErrorString(e)));
  }
  t_phase[1] = trace ? now_us() : 0.0;  // buffers
  // Small clouds (the reference's own benchmark shapes, icp_test.go:100-142): the whole Fit in one launch
  // (icp_small.hip), the target in the caller's order -- the sums run in that order.  Such a session's start values,
  // its words' zeroes and its target's coordinates are ONE launch's work (small_prepare_kernel) behind the upload: a
  // host-pointer Fit is then upload, that launch, the Fit's, the result -- it was ten operations on the stream, 5 us apart.
  static const bool small_on = icp_knob("PCGX_ICP_SMALL", 1, 0, 1) != 0;
  s->small = small_on && nt > 0 && !s->plane && !patched && s->strict == 1 && !base->has_nan && small_fit_eligible(base->view(), nt, base->many_ties);
  if (s->small) {
    s->general_ready = false;
    s->host_iter = 0;
  } else {
    if ((rc = reset_state(s, st)) != PCGX_OK) return bail(rc);
    s->general_ready = false;
    if ((rc = general_prepare(s, st)) != PCGX_OK) return bail(rc);
  }
  if (nt > 0) {
    Arena &ar = ctx().arena;
    if ((rc = ar.begin(st)) != PCGX_OK) return bail(rc);
    const float *d_q = target;
    bool from_pinned = false;
    if (!target_on_device && s->small) {  // (a small target: read by small_prepare_kernel out of the context's pinned memory)
      if (const void *up = small_upload(target, (size_t)nt * 12)) {
        d_q = static_cast<const float *>(up);
        from_pinned = true;
        t_phase[2] = t_phase[3] = trace ? now_us() : 0.0;
      }
    }
    if (!target_on_device && !from_pinned) {
      float *stage = nullptr;
      if ((rc = ar.alloc_n((size_t)nt * 3, &stage)) != PCGX_OK) return bail(rc);
      t_phase[2] = trace ? now_us() : 0.0;  // memsets, arena
      if ((rc = staged_upload(stage, target, (size_t)nt * 12, st)) != PCGX_OK) return bail(rc);
      d_q = stage;
      t_phase[3] = trace ? now_us() : 0.0;  // upload enqueued (pageable memory: staged by the runtime)
    }
    if (s->small) {
      s->nt_pad = (nt + 63) & ~(int64_t)63;
      if ((e = dev_cache_alloc((void **)&s->d_terms, small_fit_terms_bytes(nt))) != hipSuccess ||
          (e = dev_cache_alloc((void **)&s->d_valid, (size_t)(s->nt_pad / 64) * sizeof(unsigned long long))) != hipSuccess ||
          (e = dev_cache_alloc(&s->d_small_sync, small_fit_sync_bytes())) != hipSuccess)
        return bail(fail(PCGX_E_OOM, "icp session allocation failed: %s", hipGetErrorString(e)));
      if (small_fit_wants_order(nt) && (e = dev_cache_alloc((void **)&s->d_small_perm, (size_t)nt * sizeof(int32_t))) != hipSuccess)
        return bail(fail(PCGX_E_OOM, "icp session allocation failed: %s", hipGetErrorString(e)));
      if ((rc = small_fit_prepare(d_q, nt, base->bbox_lo, base->bbox_hi, s->d_small_perm, s->d_xyz, s->d_pos_of, s->d_state, s->d_terms,
                                  s->d_small_sync, st)) != PCGX_OK)
        return bail(rc);
      if (from_pinned) small_upload_read(st);
    }
    int32_t *perm = nullptr;
    if (nt > 1 && !s->small) {
      if ((rc = ar.alloc_n((size_t)nt, &perm)) != PCGX_OK) return bail(rc);
      if ((rc = morton_order(d_q, nt, base->bbox_lo, base->bbox_hi, perm, st)) != PCGX_OK) return bail(rc);
    }
    if (!s->small)
      hipLaunchKernelGGL(gather_soa_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, st, d_q, perm, nt,
                         s->d_xyz, s->d_xyz + nt, s->d_xyz + 2 * nt, s->d_pos_of);
    t_phase[4] = trace ? now_us() : 0.0;  // order + gather enqueued
    // No wait here: what follows on this session is enqueued on this stream, behind the gather, or on another stream
    // through an entry point that names it -- those wait for this one first (touch(): a session's first use on a stream
    // that is not the one it was made on).  The host's twenty step enqueues (0.3 ms) used to start only when upload,
    // order and gather had drained (0.25 ms of idle host and, behind it, idle GPU per host-pointer Fit).
    if ((e = hipGetLastError()) != hipSuccess)
      return bail(fail(PCGX_E_HIP, "icp session setup failed: %s", hipGetErrorString(e)));
  }
  if (trace)
    fprintf(stderr, "pcgx session trace: buffers %.0f us, memsets + arena %.0f, upload %.0f, order + gather enqueued %.0f\n",
            t_phase[1] - t_phase[0], t_phase[2] - t_phase[1], t_phase[3] - t_phase[2], t_phase[4] - t_phase[3]);
  *out = s;
  return PCGX_OK;
}

extern "C" pcgx_status pcgx_icp_session_create(const pcgx_kdtree *base, const float *target,
                                               int64_t nt, int32_t target_on_device,
                                               const pcgx_icp_params *params, double *d_sums10,
                                               pcgx_icp_session **out) {
  PCGX_API_LOCK();
  return session_create(base, nullptr, 0.0f, target, nt, target_on_device, params, d_sums10, out);
}

extern "C" pcgx_status pcgx_icp_plane_session_create(const pcgx_kdtree *base, const float *base_normals,
                                                     const float *target, int64_t nt, int32_t on_device,
                                                     const pcgx_icp_params *params, float damping,
                                                     double *d_sums30, pcgx_icp_session **out) {
  PCGX_API_LOCK();
  if (!base_normals) return fail(PCGX_E_INVALID, "pcgx_icp_plane_session_create: base_normals is NULL");
  if (params && params->min_dist_sq > 0.0f)
    return fail(PCGX_E_INVALID, "pcgx_icp_plane_session_create: MinDistSq > 0 (approximate search) is not offered here");
  if (!(damping >= 0.0f)) return fail(PCGX_E_INVALID, "pcgx_icp_plane_session_create: damping must be >= 0");
  return session_create(base, base_normals, damping, target, nt, on_device, params, d_sums30, out);
}

// Walk knobs of the ICP kernel (the hinted walk of iterations >= 1 profits from resolving wrong
// leaf predictions in lockstep during the chunk preparation; the cold C2 walk does not):
// PCGX_ICP_TIGHT (default 32: the whole descent), PCGX_ICP_CHUNKS (default 2 chunks per refill section).
static int icp_knob(const char *name, int def, int lo, int hi) {
  if (const char *e = getenv(name)) {
    const int v = atoi(e);
    if (v >= lo && v <= hi) return v;
  }
  return def;
}

static pcgx_status enqueue_corr_patched(pcgx_icp_session *s, hipStream_t st) {
  XTreeView xv;
  PCGX_TRY(xtree_view(s->base, &xv, st));
  const size_t lds = (size_t)(xv.depth > 1 ? xv.depth : 2) * kIcpBlock * sizeof(uint32_t);
  const int64_t guard = 4 * s->base->n + 8;
  const float *x = s->d_xyz, *y = s->d_xyz + s->nt, *z = s->d_xyz + 2 * s->nt;
  ProfScope prof(PCGX_PROF_ICP_WALK, st);
  if (s->plane)
    hipLaunchKernelGGL((icp_corr_xkernel<false, true>), dim3(s->grid), dim3(kIcpBlock), lds, st, xv, x, y, z, s->nt,
                       s->d_state, s->kp, s->d_match, s->d_partials, s->d_match_id, (const float4 *)s->d_normals, guard);
  else if (s->kp.min_dist_sq > 0.0f && s->strict)
    hipLaunchKernelGGL((icp_corr_xkernel<true, false, false>), dim3(s->grid), dim3(kIcpBlock), lds, st, xv, x, y, z, s->nt,
                       s->d_state, s->kp, s->d_match, s->d_partials, (uint32_t *)nullptr, (const float4 *)nullptr, guard);
  else if (s->kp.min_dist_sq > 0.0f)
    hipLaunchKernelGGL((icp_corr_xkernel<true, false>), dim3(s->grid), dim3(kIcpBlock), lds, st, xv, x, y, z, s->nt,
                       s->d_state, s->kp, s->d_match, s->d_partials, (uint32_t *)nullptr, (const float4 *)nullptr, guard);
  else if (s->strict)
    hipLaunchKernelGGL((icp_corr_xkernel<false, false, false>), dim3(s->grid), dim3(kIcpBlock), lds, st, xv, x, y, z, s->nt,
                       s->d_state, s->kp, s->d_match, s->d_partials, (uint32_t *)nullptr, (const float4 *)nullptr, guard);
  else
    hipLaunchKernelGGL((icp_corr_xkernel<false, false>), dim3(s->grid), dim3(kIcpBlock), lds, st, xv, x, y, z, s->nt,
                       s->d_state, s->kp, s->d_match, s->d_partials, (uint32_t *)nullptr, (const float4 *)nullptr, guard);
  PCGX_HIP_TRY(hipGetLastError());
  return PCGX_OK;
}

static pcgx_status enqueue_corr(pcgx_icp_session *s, hipStream_t st, bool may_speculate = false) {
  static const int tight = icp_knob("PCGX_ICP_TIGHT", 32, 0, 32), chunks = icp_knob("PCGX_ICP_CHUNKS", 2, 1, 64);
  // a deletion made after the session was created: from now on the reference's patched tree is walked
  // (the same handle's Nearest / Range already do), without hints from earlier iterations
  if (!s->patched && s->base->n_deleted > 0) s->patched = true;
  PCGX_TRY(general_prepare(s, st));
  const bool caller_had_pairs = s->caller_order_fresh;  // the pass before this one left every pair in the caller's order too
  s->caller_order_fresh = false;
  s->tile_sums_fresh = false;
  if (s->patched) return enqueue_corr_patched(s, st);
  if (s->strict == 1 && !s->plane && s->nt > 0) {
    // the strict sums run in the caller's target order: the kernels below also leave every pair there
    if (!s->d_match_caller) {
      hipError_t e = dev_cache_alloc((void **)&s->d_orig_of, (size_t)s->nt * sizeof(uint32_t));
      if (e == hipSuccess) e = dev_cache_alloc((void **)&s->d_match_caller, (size_t)s->nt * sizeof(float4));
      if (e != hipSuccess) {  // not required: strict.hip then gathers through pos_of
        (void)hipGetLastError();
        dev_cache_free(s->d_orig_of);
        dev_cache_free(s->d_match_caller);
        s->d_orig_of = nullptr;
        s->d_match_caller = nullptr;
      } else {
        hipLaunchKernelGGL(invert_positions_kernel, dim3((unsigned)((s->nt + 255) / 256)), dim3(256), 0, st,
                           (const uint32_t *)s->d_pos_of, s->nt, s->d_orig_of);
      }
    }
    s->caller_order_fresh = s->d_match_caller != nullptr;
    if (!s->strict_buf)
      PCGX_TRY(strict_create(s->nt, s->d_xyz, s->d_xyz + s->nt, s->d_xyz + 2 * s->nt, (const uint32_t *)s->d_pos_of,
                             &s->strict_buf, st));
  }
  TreeView tv = s->base->view();
  tv.tight_levels = tight;
  tv.chunks_per_refill = chunks;
  const size_t lds = walk_lds_bytes(tv, kIcpBlock);
  const float *x = s->d_xyz, *y = s->d_xyz + s->nt, *z = s->d_xyz + 2 * s->nt;
  const bool grid = grid_enabled(s->base) && !(s->kp.min_dist_sq > 0.0f) && s->nt > 0;
  // with the grid pass before it the correspondence kernel finds (nearly) every pair in place: its workgroups
  // form the strict sums' tile sums on their way out (else strict_tilesum_kernel does, after this launch)
  // (unless the summary kernel forms and exchanges them itself, StrictWork::exchange: the default)
  s->tile_sums_fresh = grid && s->caller_order_fresh && s->strict == 1 && !s->plane && !strict_work(s->strict_buf, s->kp)->exchange;
  const StrictWork strict_w = s->tile_sums_fresh ? *strict_work(s->strict_buf, s->kp) : StrictWork();
  // Strict sessions behind a grid pass: the correspondence kernel is there for the few targets the grid could not
  // certify (none at C4) and forms no sums -- 128 workgroups instead of two per CU: the launch of 512 of them, 66 KB
  // of LDS each, cost 4-6 us per iteration to find nothing to do.
  static const int left_blocks = icp_knob("PCGX_ICP_LEFTOVER_BLOCKS", 128, 8, 4096) & ~7;
  const int n_corr = (grid && s->strict && !s->plane && s->grid > left_blocks) ? left_blocks : s->grid;
  static const bool cert_on = icp_knob("PCGX_ICP_CERT", 1, 0, 1) != 0;  // (0: every pair is searched for, as before round 5)
  float *cert = (grid && cert_on && s->base->grid.cert) ? s->d_match_cert : nullptr;
  // The leftover walk behind the grid pass finds nothing to do in iteration after iteration (C4: never anything), and
  // its launch is 5 us of a 70 us step.  From a Fit's second Evaluate on it is therefore NOT launched behind a strict
  // session's grid pass -- on the speculation that the grid answers every target; a target it cannot answer ends the
  // step on the device (icp_grid_kernel: `done` 2) and settle() enqueues it again with the walk, as every step after it.
  static const bool spec_on = icp_knob("PCGX_ICP_SPEC_WALK", 1, 0, 1) != 0;
  static const int test_force_walk = icp_knob("PCGX_TEST_ICP_FORCE_WALK", 0, 0, 1 << 30);
  const bool no_walk = may_speculate && spec_on && s->spec_walk && grid && s->strict == 1 && !s->plane && s->host_iter >= 1 &&
                       !s->tile_sums_fresh;
  if (no_walk) {
    s->spec_pending = true;
    s->spec_stream = st;
  }
  if (grid) {
    ProfScope prof_grid(PCGX_PROF_ICP_GRID, st);
    const unsigned gb = (unsigned)((s->nt + kIcpGridBlock - 1) / kIcpGridBlock);
    if (s->plane)
      hipLaunchKernelGGL(icp_grid_kernel<true>, dim3(xcd_grid(gb)), dim3(kIcpGridBlock), 0, st, s->base->grid, x, y, z, s->nt,
                         s->d_state, s->kp, s->d_match, s->d_match_id, (const float4 *)s->d_normals, s->d_first_leaf,
                         s->d_walk_list, s->d_walk_count, (uint32_t)s->grid, s->d_partials, (unsigned long long *)nullptr,
                         (const uint32_t *)nullptr, (float4 *)nullptr, cert);
    else if (s->strict)
      hipLaunchKernelGGL((icp_grid_kernel<false, false, false>),
                         dim3(xcd_grid((unsigned)((s->nt + kIcpStrictGridBlock - 1) / kIcpStrictGridBlock))),
                         dim3(kIcpStrictGridBlock), 0, st, s->base->grid, x, y, z,
                         s->nt, s->d_state, s->kp, s->d_match, s->d_match_id, (const float4 *)s->d_normals,
                         s->d_first_leaf, s->d_walk_list, s->d_walk_count, (uint32_t)n_corr, s->d_partials,
                         (unsigned long long *)nullptr, (const uint32_t *)(s->caller_order_fresh ? s->d_orig_of : nullptr),
                         s->caller_order_fresh ? s->d_match_caller : nullptr, cert, caller_had_pairs ? 1 : 0, no_walk ? 0 : 1, test_force_walk);
    else
      hipLaunchKernelGGL(icp_grid_kernel<false>, dim3(xcd_grid(gb)), dim3(kIcpGridBlock), 0, st, s->base->grid, x, y, z, s->nt,
                         s->d_state, s->kp, s->d_match, s->d_match_id, (const float4 *)s->d_normals, s->d_first_leaf,
                         s->d_walk_list, s->d_walk_count, (uint32_t)s->grid, s->d_partials, (unsigned long long *)nullptr,
                         (const uint32_t *)nullptr, (float4 *)nullptr, cert);
  }
  // timed (pcgx_prof_enable) when it is the kernel that does the work: with the grid pass before it
  // it walks next to nothing, and a second pair of events per step costs more than it
  if (no_walk) return PCGX_OK;  // (nothing behind the grid pass)
  ProfScope prof(grid ? PCGX_PROF_ICP_LEFTOVER : PCGX_PROF_ICP_WALK, st);
#define PCGX_LAUNCH_CORR(MD, PL, GR)                                                                                  \
  do {                                                                                                                \
  if (s->strict && !PL)                                                                                               \
    hipLaunchKernelGGL((icp_corr_kernel<MD, false, GR, false>), dim3(n_corr), dim3(kIcpBlock), lds, st, tv, x, y, z, \
                       s->nt, s->d_state, s->kp, s->d_match, s->d_first_leaf, s->d_partials, s->d_match_id,           \
                       (const float4 *)s->d_normals, s->d_walk_list, s->d_walk_count,                                 \
                       (int32_t)((s->nt + kIcpGridBlock - 1) / kIcpGridBlock),                                        \
                       (const uint32_t *)(s->caller_order_fresh ? s->d_orig_of : nullptr),                            \
                       s->caller_order_fresh ? s->d_match_caller : nullptr, strict_w,                                 \
                       (int32_t)((GR) && s->tile_sums_fresh ? 1 : 0), cert, s->base->grid.cert);                      \
  else                                                                                                                \
  hipLaunchKernelGGL((icp_corr_kernel<MD, PL, GR>), dim3(s->grid), dim3(kIcpBlock), lds, st, tv, x, y, z, s->nt,     \
                     s->d_state, s->kp, s->d_match, s->d_first_leaf, s->d_partials, s->d_match_id,                   \
                     (const float4 *)s->d_normals, s->d_walk_list, s->d_walk_count,                                  \
                     (int32_t)((s->nt + kIcpGridBlock - 1) / kIcpGridBlock), (const uint32_t *)nullptr,               \
                     (float4 *)nullptr, StrictWork(), 0, cert, s->base->grid.cert);                                   \
  } while (0)
  if (s->plane) {
    if (grid) PCGX_LAUNCH_CORR(false, true, true);
    else PCGX_LAUNCH_CORR(false, true, false);
  } else if (s->kp.min_dist_sq > 0.0f) {
    PCGX_LAUNCH_CORR(true, false, false);
  } else {
    if (grid) PCGX_LAUNCH_CORR(false, false, true);
    else PCGX_LAUNCH_CORR(false, false, false);
  }
#undef PCGX_LAUNCH_CORR
  return PCGX_OK;
}

template <bool kFuseUpdate>
static pcgx_status enqueue_strict(pcgx_icp_session *s, hipStream_t st) {
  if (s->strict == 1) {  // the whole GPU: strict.hip
    if (!s->strict_buf)
      PCGX_TRY(strict_create(s->nt, s->d_xyz, s->d_xyz + s->nt, s->d_xyz + 2 * s->nt, (const uint32_t *)s->d_pos_of,
                             &s->strict_buf, st));
    const bool first_iter = s->host_iter++ == 0;
    if (s->caller_order_fresh)  // pairs already in the caller's order: no gather through pos_of
      PCGX_TRY(strict_enqueue(s->strict_buf, (const float4 *)s->d_match_caller, (const uint32_t *)nullptr, s->d_state,
                              s->d_sums, s->kp, kFuseUpdate, s->tile_sums_fresh, first_iter, st));
    else
      PCGX_TRY(strict_enqueue(s->strict_buf, (const float4 *)s->d_match, (const uint32_t *)s->d_pos_of, s->d_state,
                              s->d_sums, s->kp, kFuseUpdate, false, first_iter, st));
    return PCGX_OK;
  }
  // strict 2: the plain dependent chain, one wave (kept as the on-device cross-check of strict 1)
  if (!s->d_terms) {  // first strict launch of the session
    s->nt_pad = (s->nt + 63) & ~(int64_t)63;
    const size_t np = (size_t)(s->nt_pad ? s->nt_pad : 64);
    PCGX_HIP_TRY(dev_cache_alloc((void **)&s->d_terms, 9 * np * sizeof(float)));
    PCGX_HIP_TRY(dev_cache_alloc((void **)&s->d_valid, (np / 64) * sizeof(unsigned long long)));
  }
  return strict_check_enqueue(s->d_xyz, s->nt, s->nt_pad, (const float4 *)s->d_match, (const uint32_t *)s->d_pos_of, s->d_state, s->kp,
                              s->d_terms, s->d_valid, s->d_sums, kFuseUpdate, st);
}

// Steps enqueued without the leftover walk (enqueue_corr) are looked at: if one of them met a target the grid could not
// answer, the device stopped there (`done` 2) -- that step and the ones enqueued behind it are enqueued again, with the
// walk, and the session keeps the walk from then on.  Called by everything that reads the session's state.
static pcgx_status settle(pcgx_icp_session *s, hipStream_t st) {
  if (!s->spec_pending) return PCGX_OK;
  s->spec_pending = false;
  IcpState h;
  PCGX_HIP_TRY(hipMemcpyAsync(&h, s->d_state, sizeof h, hipMemcpyDeviceToHost, st));
  PCGX_HIP_TRY(hipStreamSynchronize(st));
  if (h.done != 2) return PCGX_OK;
  s->spec_walk = false;
  const int32_t missing = s->host_iter - h.num_iteration;  // (Evaluates enqueued, Evaluates carried out)
  if (getenv("PCGX_ICP_SPEC_TRACE"))
    fprintf(stderr, "pcgx icp: a step without the leftover walk met a target the grid could not answer: %d of %d Evaluates enqueued again, with the walk\n",
            (int)missing, (int)s->host_iter);
  const int32_t zero = 0;
  PCGX_HIP_TRY(hipMemcpyAsync(&s->d_state->done, &zero, sizeof zero, hipMemcpyHostToDevice, st));
  PCGX_HIP_TRY(hipMemsetAsync(s->d_walk_count, 0, (size_t)s->grid * sizeof(uint32_t), st));
  s->host_iter = h.num_iteration;
  for (int32_t k = 0; k < missing; k++) {
    PCGX_TRY(enqueue_corr(s, st, false));
    PCGX_TRY(enqueue_strict<true>(s, st));
  }
  PCGX_HIP_TRY(hipGetLastError());
  return PCGX_OK;
}

// Measurement aid: counters of the strict chain since the last call (see include/pcgx.h).
extern "C" pcgx_status pcgx_debug_icp_strict_stats(pcgx_icp_session *s, void *stream, int64_t out[64]) {
  PCGX_API_LOCK();
  if (!s || !out) return fail(PCGX_E_INVALID, "pcgx_debug_icp_strict_stats: NULL argument");
  for (int k = 0; k < 64; k++) out[k] = 0;
  if (!s->strict_buf) return PCGX_OK;
  PCGX_TRY(settle(s, pick_stream(stream)));  // (the counters of steps the device skipped would be missing)
  unsigned long long h[64];
  PCGX_TRY(strict_read_debug(s->strict_buf, h, pick_stream(stream)));
  for (int k = 0; k < 64; k++) out[k] = (int64_t)h[k];
  return PCGX_OK;
}

extern "C" pcgx_status pcgx_icp_session_partials(pcgx_icp_session *s, void *stream) {
  PCGX_API_LOCK();
  if (!s) return fail(PCGX_E_INVALID, "pcgx_icp_session_partials: NULL session");
  hipStream_t st = pick_stream(stream);
  s->touch(st);
  PCGX_TRY(settle(s, st));  // (steps enqueued without the leftover walk: enqueue_corr)
  PCGX_TRY(enqueue_corr(s, st));
  if (s->strict)
    PCGX_TRY(enqueue_strict<false>(s, st));
  else if (s->plane)
    hipLaunchKernelGGL((icp_final_reduce_kernel<false, true>), dim3(1), dim3(1024), 0, st, s->d_partials, s->grid,
                       s->d_state, s->d_sums, s->kp);
  else
    hipLaunchKernelGGL((icp_final_reduce_kernel<false, false>), dim3(1), dim3(64 * S_COUNT), 0, st, s->d_partials,
                       s->grid, s->d_state, s->d_sums, s->kp);
  PCGX_HIP_TRY(hipGetLastError());
  return PCGX_OK;
}

extern "C" pcgx_status pcgx_icp_session_update(pcgx_icp_session *s, void *stream) {
  PCGX_API_LOCK();
  if (!s) return fail(PCGX_E_INVALID, "pcgx_icp_session_update: NULL session");
  hipStream_t st = pick_stream(stream);
  s->touch(st);
  PCGX_TRY(settle(s, st));  // (steps enqueued without the leftover walk: enqueue_corr)
  if (s->plane)
    hipLaunchKernelGGL(icp_update_kernel<true>, dim3(1), dim3(64), 0, st, s->d_state, s->d_sums, s->kp);
  else
    hipLaunchKernelGGL(icp_update_kernel<false>, dim3(1), dim3(64), 0, st, s->d_state, s->d_sums, s->kp);
  PCGX_HIP_TRY(hipGetLastError());
  return PCGX_OK;
}

// A small session's steps in one launch (icp_small.hip) -- while it is what it was made as: the reference's sums, the
// canonical tree (a deletion since sends the session to the patched tree's walk, enqueue_corr)
static bool small_now(const pcgx_icp_session *s) {
  return s->small && s->strict == 1 && !s->plane && !s->patched && s->base->n_deleted == 0 && !s->spec_pending &&
         !(s->kp.max_dist_sq < s->kp.min_dist_sq);  // (maxRange^2 < MinDistSq: the walk ends at its first leaf, kdtree.go:100-106)
}
static pcgx_status small_steps(pcgx_icp_session *s, hipStream_t st, int iters, uint32_t *mail_seq = nullptr) {
  s->caller_order_fresh = false;
  s->tile_sums_fresh = false;
  s->host_iter += iters;
  static std::atomic<uint32_t> launches{0};  // (one count for the process: a block of the cache keeps its last session's words)
  // ONE such launch at a time on a device: its workgroups wait for each other inside the launch, all of them have to be
  // on the chip together -- and two launches dealt out side by side from two streams' queues may each get a part of
  // the chip and wait for the rest (until their 2 s run out and the Fits end with PCGX_E_HIP).  A launch starts behind
  // the one before it, whatever stream that was on (an event; the hosts' threads do not wait for each other's Fits).
  // (Kernels that end by themselves may share the chip with it: they make room.)
  static std::mutex one_at_a_time;
  static hipEvent_t last_done[16] = {};
  std::lock_guard<std::mutex> lk(one_at_a_time);
  const int slot = current_slot() & 15;
  if (!last_done[slot]) PCGX_HIP_TRY(hipEventCreateWithFlags(&last_done[slot], hipEventDisableTiming));
  else PCGX_HIP_TRY(hipStreamWaitEvent(st, last_done[slot], 0));
  struct Record {  // (whatever way this function is left: the next launch waits for what was enqueued here)
    hipEvent_t ev;
    hipStream_t st;
    ~Record() { (void)hipEventRecord(ev, st); }
  } record{last_done[slot], st};
  while (iters > 0) {
    const int now = iters < small_fit_max_iters() ? iters : small_fit_max_iters();
    // (mail_seq: the launch that ends the Fit leaves the loop state in the context's mailbox -- pcgx_icp_fit)
    const bool mail = mail_seq != nullptr && iters == now && ctx().mailbox != nullptr;
    if (mail) *mail_seq = mailbox_next_seq();
    PCGX_TRY(small_fit_enqueue(s->base->view(), s->d_xyz, s->d_xyz + s->nt, s->d_xyz + 2 * s->nt, s->nt, s->d_state, s->kp, s->d_terms,
                               s->d_valid, s->d_sums, s->d_small_sync, ++launches, now, s->d_small_perm, st, mail ? ctx().mailbox : nullptr,
                               mail ? *mail_seq : 0u));
    iters -= now;
  }
  return PCGX_OK;
}

// partials + update with no exchange in between (single GPU): two launches per iteration.
extern "C" pcgx_status pcgx_icp_session_step(pcgx_icp_session *s, void *stream) {
  PCGX_API_LOCK();
  if (!s) return fail(PCGX_E_INVALID, "pcgx_icp_session_step: NULL session");
  hipStream_t st = pick_stream(stream);
  s->touch(st);
  if (small_now(s)) return small_steps(s, st, 1);
  PCGX_TRY(enqueue_corr(s, st, true));
  if (s->strict)
    PCGX_TRY(enqueue_strict<true>(s, st));
  else if (s->plane)
    hipLaunchKernelGGL((icp_final_reduce_kernel<true, true>), dim3(1), dim3(1024), 0, st, s->d_partials, s->grid,
                       s->d_state, s->d_sums, s->kp);
  else
    hipLaunchKernelGGL((icp_final_reduce_kernel<true, false>), dim3(1), dim3(64 * S_COUNT), 0, st, s->d_partials,
                       s->grid, s->d_state, s->d_sums, s->kp);
  PCGX_HIP_TRY(hipGetLastError());
  return PCGX_OK;
}

// One iteration of a Fit whose target is spread over the ranks of `c`.  Two numeric modes:
//  * the reference's sums (the default, sums_mode 0 / set_strict 1): its sequential float32 additions over the ranks'
//    tiles one after the other, rank 0's first (strict.hip, strict_enqueue_sharded): the Fit of the concatenated
//    target bit for bit, 2 + world small collectives per iteration;
//  * float64 sums (PCGX_SUMS_F64_TREE): one all-reduce of the 10 (plane: 30) sums.
// Either way every collective carries the ranks' error flag, and a rank that cannot go on (*local_rc) keeps calling
// them with its flag up: all ranks end the Fit in the same iteration, nobody waits for a peer that has left.
// Returns the COMMUNICATOR's status (a failed collective is the one thing that may not be survived).
static pcgx_status step_sharded_impl(pcgx_icp_session *s, pcgx_comm *c, void *stream, pcgx_status *local_rc) {
  int32_t rank = 0, world = 1;
  PCGX_TRY(pcgx_comm_rank(c, &rank, &world));
  hipStream_t st = pick_stream(stream);
  s->touch(st);
  PCGX_TRY(settle(s, st));  // (steps enqueued without the leftover walk: enqueue_corr)
  const int step = s->steps_sharded++;
  // the first sharded step since the session was made or reset: a Fit begins on the communicator -- whatever an earlier
  // Fit left in the ring's abort words is not this one's business (every rank gets here at the same point of its calls)
  if (step == 0 && world > 1 && s->strict == 1 && !s->plane && !s->ring_fit_open) comm_ring_new_fit(c);
  s->ring_fit_open = false;
  if (const char *e = getenv("PCGX_TEST_FAIL_RANK")) {  // fault injection (tests/test_gpu_multi.py)
    const char *it = getenv("PCGX_TEST_FAIL_ITER");
    if (atoi(e) == rank && it && atoi(it) == step && !s->shard_failed) {
      *local_rc = fail(PCGX_E_HIP, "injected failure of rank %d in iteration %d (PCGX_TEST_FAIL_RANK)", rank, step);
      s->shard_failed = true;
    }
  }
  const bool reference = s->strict == 1 && !s->plane;
  if (reference) {
    // the ring form (strict.hip, strict_enqueue_ring) wherever the ranks can share memory: no collective per step.
    // (Asked for first, by every rank whatever its own state: making the ring is collective.)
    RingView ring;
    const bool have_ring = comm_ring_step(c, step, &ring);
    if (!s->shard_failed) {
      const pcgx_status rc = enqueue_corr(s, st);
      if (rc != PCGX_OK) {
        *local_rc = rc;
        s->shard_failed = true;
      }
    }
    if (!s->strict_buf) {
      const pcgx_status rc = strict_create(s->nt, s->d_xyz, s->d_xyz + s->nt, s->d_xyz + 2 * s->nt, (const uint32_t *)s->d_pos_of,
                                           &s->strict_buf, st);
      if (rc != PCGX_OK) return rc;  // (no buffers at all: this rank cannot even raise its flag)
    }
    if (have_ring) {
      const bool first_iter = s->host_iter++ == 0;
      if (s->caller_order_fresh)
        return strict_enqueue_ring(s->strict_buf, (const float4 *)s->d_match_caller, (const uint32_t *)nullptr, s->d_state,
                                   s->d_sums, s->kp, ring, s->shard_failed, first_iter, st);
      return strict_enqueue_ring(s->strict_buf, (const float4 *)s->d_match, (const uint32_t *)s->d_pos_of, s->d_state,
                                 s->d_sums, s->kp, ring, s->shard_failed, first_iter, st);
    }
    if (s->caller_order_fresh)
      return strict_enqueue_sharded(s->strict_buf, (const float4 *)s->d_match_caller, (const uint32_t *)nullptr, s->d_state,
                                    s->d_sums, s->kp, c, rank, world, s->shard_failed, st);
    return strict_enqueue_sharded(s->strict_buf, (const float4 *)s->d_match, (const uint32_t *)s->d_pos_of, s->d_state,
                                  s->d_sums, s->kp, c, rank, world, s->shard_failed, st);
  }
  const int n = s->n_sums();
  if (!s->d_xchg) {
    if (dev_cache_alloc((void **)&s->d_xchg, (size_t)(n + 2) * sizeof(double)) != hipSuccess)
      return fail(PCGX_E_OOM, "pcgx_icp_session_step_sharded: no memory for the exchange");
  }
  PCGX_HIP_TRY(hipMemsetAsync(s->d_xchg, 0, (size_t)(n + 2) * sizeof(double), st));
  if (!s->shard_failed) {
    double *keep = s->d_sums;
    s->d_sums = s->d_xchg;  // the reduction writes where the all-reduce reads
    const pcgx_status rc = pcgx_icp_session_partials(s, stream);
    s->d_sums = keep;
    if (rc != PCGX_OK) {
      *local_rc = rc;
      s->shard_failed = true;
    }
  }
  if (s->shard_failed) {
    static const double one = 1.0;
    PCGX_HIP_TRY(hipMemcpyAsync(s->d_xchg + n, &one, sizeof one, hipMemcpyHostToDevice, st));
  }
  PCGX_TRY(pcgx_comm_allreduce_f64(c, s->d_xchg, n + 1, stream));
  if (s->plane)
    hipLaunchKernelGGL(icp_update_sharded_kernel<true>, dim3(1), dim3(64), 0, st, s->d_state, (const double *)s->d_xchg, n,
                       s->d_sums, s->kp);
  else
    hipLaunchKernelGGL(icp_update_sharded_kernel<false>, dim3(1), dim3(64), 0, st, s->d_state, (const double *)s->d_xchg, n,
                       s->d_sums, s->kp);
  PCGX_HIP_TRY(hipGetLastError());
  return PCGX_OK;
}

extern "C" pcgx_status pcgx_icp_session_step_sharded(pcgx_icp_session *s, pcgx_comm *c, void *stream) {
  PCGX_API_LOCK();
  if (!s || !c) return fail(PCGX_E_INVALID, "pcgx_icp_session_step_sharded: NULL argument");
  int32_t rank = 0, world = 1;
  PCGX_TRY(pcgx_comm_rank(c, &rank, &world));
  if (world == 1 && !(getenv("PCGX_COMM_FORCE_COLLECTIVE") && atoi(getenv("PCGX_COMM_FORCE_COLLECTIVE")) != 0))
    return pcgx_icp_session_step(s, stream);
  if (s->strict == 2)
    return fail(PCGX_E_INVALID, "the one-wave chain (PCGX_SUMS_REFERENCE_CHAIN) is not offered on a sharded target");
  pcgx_status local = PCGX_OK;
  const pcgx_status comm_rc = step_sharded_impl(s, c, stream, &local);
  return local != PCGX_OK ? local : comm_rc;
}

extern "C" pcgx_status pcgx_icp_fit_sharded(const pcgx_kdtree *base, const float *tile, int64_t nt,
                                            const pcgx_icp_params *params, pcgx_comm *c, float trans16[16],
                                            pcgx_icp_stat *stat) {
  PCGX_API_LOCK();
  if (!base || !params || !trans16 || !c) return fail(PCGX_E_INVALID, "pcgx_icp_fit_sharded: NULL argument");
  int32_t rank = 0, world = 1;
  PCGX_TRY(pcgx_comm_rank(c, &rank, &world));  // (before anything is made: nothing to give back, no collective missed)
  pcgx_icp_session *s = nullptr;
  pcgx_status rc = pcgx_icp_session_create(base, tile, nt, 0, params, nullptr, &s);
  std::string first_error = rc != PCGX_OK ? std::string(last_error_text()) : std::string();
  if (world > 1) comm_ring_new_fit(c);  // (makes the communicator's ring on first use -- collective; a rank without a session counts the Fit too)
  if (s) s->ring_fit_open = true;
  if (world > 1) {
    // A rank whose session could not be made (out of memory, a bad argument) must not leave the others
    // waiting in the first all-reduce: every rank reaches ONE exchange of an error flag first -- out of a host word
    // if even the device word cannot be had -- and all of them give up together if any has failed.
    double *d_flag = nullptr;
    const double mine = rc == PCGX_OK ? 0.0 : 1.0;
    double all = 1.0;
    pcgx_status rx = PCGX_OK;
    if (dev_cache_alloc((void **)&d_flag, sizeof(double)) != hipSuccess) {
      (void)hipGetLastError();
      d_flag = nullptr;
      rx = pcgx_comm_allreduce_host_f64(c, &all, 1);  // (all == 1: this rank reports a failure)
      if (rc == PCGX_OK) rc = fail(PCGX_E_OOM, "pcgx_icp_fit_sharded: no device memory for the ranks' error exchange");
    } else {
      if (hipMemcpyAsync(d_flag, &mine, sizeof mine, hipMemcpyHostToDevice, ctx().stream) != hipSuccess) rx = PCGX_E_HIP;
      if (rx == PCGX_OK) rx = pcgx_comm_allreduce_f64(c, d_flag, 1, nullptr);
      if (rx == PCGX_OK && (hipMemcpyAsync(&all, d_flag, sizeof all, hipMemcpyDeviceToHost, ctx().stream) != hipSuccess ||
                            hipStreamSynchronize(ctx().stream) != hipSuccess))
        rx = PCGX_E_HIP;
      dev_cache_free(d_flag);
    }
    if (rc == PCGX_OK && rx != PCGX_OK) rc = fail(rx, "pcgx_icp_fit_sharded: the ranks' error exchange failed");
    if (rc == PCGX_OK && all != 0.0) rc = fail(PCGX_E_RCCL, "pcgx_icp_fit_sharded: another rank could not set up its session");
  }
  if (rc != PCGX_OK) {
    if (s) pcgx_icp_session_free(s);
    if (!first_error.empty()) return fail(rc, "%s", first_error.c_str());
    return rc;
  }
  // Every rank enqueues MaxIteration steps: the loop state is the same on all of them (same sums), so they stop
  // together, and a step after `done` is a no-op on the device (its collectives still run).  A rank whose step fails
  // goes on calling the collectives with its flag up (step_sharded_impl), so the others see the failure in a
  // collective they all reach and end their Fits in that same iteration -- nobody is left inside an all-reduce.
  pcgx_status local = PCGX_OK;
  for (int it = 0; it < s->max_iteration && rc == PCGX_OK; it++) {
    pcgx_status l = PCGX_OK;
    if (world == 1 && !(getenv("PCGX_COMM_FORCE_COLLECTIVE") && atoi(getenv("PCGX_COMM_FORCE_COLLECTIVE")) != 0)) {
      rc = pcgx_icp_session_step(s, nullptr);
      continue;
    }
    rc = step_sharded_impl(s, c, nullptr, &l);
    if (l != PCGX_OK && local == PCGX_OK) {
      local = l;
      first_error = last_error_text();
    }
  }
  if (rc == PCGX_OK) rc = pcgx_icp_session_result(s, nullptr, trans16, stat, nullptr);
  pcgx_icp_session_free(s);
  if (local != PCGX_OK) return fail(local, "%s", first_error.c_str());
  return rc;
}

// ---- one process, several GPUs: a host thread per device slot, the exchange between them in host memory ---------
namespace {
struct LocalExchange {  // all-reduce (sum, rank order: bitwise reproducible) of `count` doubles between the threads of one process
  std::mutex mu;
  std::condition_variable cv;
  int world = 1, arrived = 0;
  long generation = 0;
  std::vector<std::vector<double>> parts;
  std::vector<double> total;
  bool broken = false;
};
struct LocalRank {
  LocalExchange *x;
  int rank;
};
int32_t local_allreduce(double *buf, int32_t count, void *user) {
  LocalRank *me = (LocalRank *)user;
  LocalExchange &x = *me->x;
  std::unique_lock<std::mutex> lk(x.mu);
  if (x.broken) return 2;
  x.parts[(size_t)me->rank].assign(buf, buf + count);
  const long gen = x.generation;
  if (++x.arrived == x.world) {
    x.total.assign((size_t)count, 0.0);
    for (int r = 0; r < x.world; r++) {
      if ((int)x.parts[(size_t)r].size() != count) x.broken = true;  // the ranks disagree about what they exchange
      else
        for (int k = 0; k < count; k++) x.total[(size_t)k] += x.parts[(size_t)r][(size_t)k];
    }
    x.arrived = 0;
    x.generation++;
    x.cv.notify_all();
  } else {
    // a peer that never arrives (it left its Fit on a path that skips a collective: a bug) must not hang the process
    if (!x.cv.wait_for(lk, std::chrono::seconds(60), [&] { return x.generation != gen || x.broken; })) x.broken = true;
  }
  if (x.broken) {
    x.cv.notify_all();
    return 2;
  }
  memcpy(buf, x.total.data(), (size_t)count * sizeof(double));
  return 0;
}
}  // namespace

// Fit with the target spread over n device slots of THIS process (pcgx_init_devices): bases[r] is the tree replica made
// on slot r, tiles[r] / nt[r] slot r's part of the target (host memory).  A thread per slot runs pcgx_icp_fit_sharded
// with an exchange in host memory; the result is every slot's (they agree), the status the first slot's that failed.
extern "C" pcgx_status pcgx_icp_fit_multi(int32_t n, const pcgx_kdtree *const *bases, const float *const *tiles,
                                          const int64_t *nt, const pcgx_icp_params *params, float trans16[16],
                                          pcgx_icp_stat *stat) {
  if (n < 1 || !bases || !tiles || !nt || !params || !trans16)
    return fail(PCGX_E_INVALID, "pcgx_icp_fit_multi: bad argument");
  LocalExchange x;
  x.world = n;
  x.parts.resize((size_t)n);
  // the ring of the reference-sums steps: every slot's inbox in pinned host memory that all the process's GPUs write
  // and poll (strict.hip, strict_enqueue_ring); without it (no such memory to be had, PCGX_SHARD_RING=0) the slots
  // exchange through the host all-reduce above
  unsigned long long *ring_block = nullptr;
  const RingLayout RL{n};
  {
    // (slots that share one HIP device -- a one-GPU test box -- have a hardware queue each for this: pcgx_init_devices)
    const char *off = getenv("PCGX_SHARD_RING");
    if (n > 1 && n <= 64 && !(off && off[0] == '0') && ensure_init() == PCGX_OK) {
      const size_t bytes = (size_t)n * RL.words() * sizeof(unsigned long long);
      // (Coherent by name: fine-grained memory the GPUs never cache.  Left to the runtime's default the block may be
      // coarse-grained -- a walker that polled a word before it arrived then kept reading the stale line out of its L2:
      // one Fit in a fresh process out of a few ended in the walk's time-out.)
      if (hipHostMalloc((void **)&ring_block, bytes, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess)
        memset(ring_block, 0, bytes);
      else {
        (void)hipGetLastError();
        ring_block = nullptr;
      }
    }
  }
  std::vector<LocalRank> ranks((size_t)n);
  std::vector<pcgx_status> rc((size_t)n, PCGX_OK);
  std::vector<std::string> msg((size_t)n);
  std::vector<std::array<float, 16>> tr((size_t)n);
  std::vector<pcgx_icp_stat> stv((size_t)n);
  std::vector<std::thread> th;
  for (int r = 0; r < n; r++) {
    ranks[(size_t)r] = LocalRank{&x, r};
    th.emplace_back([&, r]() {
      pcgx_status e = pcgx_set_device(r);
      pcgx_comm *c = nullptr;
      if (e == PCGX_OK) e = pcgx_comm_init_callback(r, n, local_allreduce, &ranks[(size_t)r], &c);
      if (e == PCGX_OK && ring_block) comm_attach_local_ring(c, ring_block, RL.words());
      if (e == PCGX_OK) e = pcgx_icp_fit_sharded(bases[r], tiles[r], nt[r], params, c, tr[(size_t)r].data(), &stv[(size_t)r]);
      if (e != PCGX_OK) {
        msg[(size_t)r] = last_error_text();
        if (ring_block) {  // (... nor in a kernel's wait for a word of this slot's: the ring is broken from its first step on)
          RingView v;
          v.words = v.host = ring_block;
          v.words_per_rank = RL.words();
          v.rank = r;
          v.world = n;
          v.epoch = 1u << kRingTagStepBits | 1u;  // (the communicators of this call are fresh: Fit 1, its first step)
          ring_abort_from_host(v, 1u);
        }
        std::lock_guard<std::mutex> lk(x.mu);  // (a rank that could not even start must not leave the others waiting)
        if (!c) {
          x.broken = true;
          x.cv.notify_all();
        }
      }
      if (c) pcgx_comm_free(c);
      rc[(size_t)r] = e;
    });
  }
  for (auto &t : th) t.join();
  unsigned long long ring_abort_word = 0ull;  // (what broke the ring, if anything did: reason | step << 32)
  if (ring_block) {
    ring_abort_word = ring_block[RL.abort()];
    (void)hipHostFree(ring_block);
  }
  // the rank that failed by itself speaks first; PCGX_E_RCCL ("another rank ...") only if nobody has a better story
  int pick = -1;
  for (int r = 0; r < n && pick < 0; r++)
    if (rc[(size_t)r] != PCGX_OK && rc[(size_t)r] != PCGX_E_RCCL) pick = r;
  for (int r = 0; r < n && pick < 0; r++)
    if (rc[(size_t)r] != PCGX_OK) pick = r;
  if (pick >= 0) {
    if (ring_abort_word)
      return fail(rc[(size_t)pick], "pcgx_icp_fit_multi: slot %d: %s (the ring's abort word: reason 0x%x in step %u)", pick,
                  msg[(size_t)pick].c_str(), (unsigned)(ring_abort_word & 0xffffffffu), (unsigned)(ring_abort_word >> 32));
    return fail(rc[(size_t)pick], "pcgx_icp_fit_multi: slot %d: %s", pick, msg[(size_t)pick].c_str());
  }
  memcpy(trans16, tr[0].data(), 16 * sizeof(float));
  if (stat) *stat = stv[0];
  return PCGX_OK;
}

static pcgx_status result_of(const pcgx_icp_session *s, const IcpState &h, float trans16[16], pcgx_icp_stat *stat, int32_t *converged);

extern "C" pcgx_status pcgx_icp_session_result(pcgx_icp_session *s, void *stream, float trans16[16],
                                               pcgx_icp_stat *stat, int32_t *converged) {
  PCGX_API_LOCK();
  if (!s) return fail(PCGX_E_INVALID, "pcgx_icp_session_result: NULL session");
  hipStream_t st = pick_stream(stream);
  s->touch(st);
  PCGX_TRY(settle(s, st));  // (steps enqueued without the leftover walk: enqueue_corr)
  IcpState h;
  PCGX_TRY(read_back_small(s->d_state, sizeof h, &h, st));  // (the stream is not waited for: a word from behind its last kernel is)
  return result_of(s, h, trans16, stat, converged);
}

static pcgx_status result_of(const pcgx_icp_session *s, const IcpState &h, float trans16[16], pcgx_icp_stat *stat, int32_t *converged) {
  if (trans16) memcpy(trans16, h.trans, sizeof h.trans);
  if (stat) {
    stat->evaluated.value = h.ev.value;
    memcpy(stat->evaluated.gradient, h.ev.gradient, sizeof h.ev.gradient);
    stat->evaluated.dist_rms = h.ev.dist_rms;
    stat->evaluated.num_pairs = h.ev.num_pairs;
    stat->num_iteration = h.num_iteration;
  }
  if (converged) *converged = (h.done == 1 && h.status == PCGX_OK) ? 1 : 0;
  if (h.status == PCGX_E_NOT_ENOUGH_PAIRS)
    return fail(PCGX_E_NOT_ENOUGH_PAIRS, "not enough correspondence pairs (%lld < %d) at iteration %d",
                (long long)h.ev.num_pairs, s->kp.min_pairs, h.num_iteration);
  if (h.status == PCGX_E_SINGULAR)
    return fail(PCGX_E_SINGULAR, "normal equations are not positive definite at iteration %d", h.num_iteration);
  if (h.status == PCGX_E_RCCL)  // (a sharded Fit: strict_finish_kernel / the ring's abort word, strict.hip)
    return fail(PCGX_E_RCCL, "the sharded Fit ended after %d iterations: another rank could not go on, or a wait for another rank's "
                             "state ran out of time", h.num_iteration);
  return PCGX_OK;
}

extern "C" pcgx_status pcgx_icp_session_hessian(pcgx_icp_session *s, void *stream, float hessian36[36]) {
  PCGX_API_LOCK();
  if (!s || !hessian36) return fail(PCGX_E_INVALID, "pcgx_icp_session_hessian: bad argument");
  if (!s->plane) return fail(PCGX_E_INVALID, "pcgx_icp_session_hessian: not a plane session (HasHessian() == false)");
  hipStream_t st = pick_stream(stream);
  s->touch(st);
  PCGX_TRY(settle(s, st));  // (steps enqueued without the leftover walk: enqueue_corr)
  IcpState h;
  PCGX_HIP_TRY(hipMemcpyAsync(&h, s->d_state, sizeof h, hipMemcpyDeviceToHost, st));
  PCGX_HIP_TRY(hipStreamSynchronize(st));
  memcpy(hessian36, h.hessian, sizeof h.hessian);
  return PCGX_OK;
}

extern "C" pcgx_status pcgx_icp_plane_fit(const pcgx_kdtree *base, const float *base_normals, const float *target,
                                          int64_t nt, const pcgx_icp_params *params, float damping,
                                          float trans16[16], pcgx_icp_stat *stat, float hessian36[36]) {
  PCGX_API_CALL();
  if (!base || !params || !trans16) return fail(PCGX_E_INVALID, "pcgx_icp_plane_fit: NULL argument");
  pcgx_icp_session *s = nullptr;
  PCGX_TRY(pcgx_icp_plane_session_create(base, base_normals, target, nt, 0, params, damping, nullptr, &s));
  pcgx_status rc = PCGX_OK;
  for (int it = 0; it < s->max_iteration && rc == PCGX_OK; it++) rc = pcgx_icp_session_step(s, nullptr);
  if (rc == PCGX_OK) rc = pcgx_icp_session_result(s, nullptr, trans16, stat, nullptr);
  if (rc == PCGX_OK && hessian36) rc = pcgx_icp_session_hessian(s, nullptr, hessian36);
  pcgx_icp_session_free(s);
  return rc;
}

extern "C" pcgx_status pcgx_icp_fit(const pcgx_kdtree *base, const float *target, int64_t nt,
                                    const pcgx_icp_params *params, float trans16[16],
                                    pcgx_icp_stat *stat) {
  PCGX_API_CALL();
  if (!base || !params || !trans16) return fail(PCGX_E_INVALID, "pcgx_icp_fit: NULL argument");
  {  // what Fit returns when its first Evaluate fails (icp.go:47-53): identity, NumIteration 1
    const Mat4 id = mat4_translate(0.0f, 0.0f, 0.0f);
    memcpy(trans16, id.m, sizeof id.m);
    if (stat) {
      memset(stat, 0, sizeof *stat);
      stat->num_iteration = 1;
    }
  }
  // (PCGX_FIT_TRACE: where a host-pointer Fit's wall time goes, per thread -- tools/conc4_probe.py)
  static const bool trace = getenv("PCGX_FIT_TRACE") != nullptr;
  auto now_us = []() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = trace ? now_us() : 0.0;
  pcgx_icp_session *s = nullptr;
  PCGX_TRY(pcgx_icp_session_create(base, target, nt, 0, params, nullptr, &s));
  const double t1 = trace ? now_us() : 0.0;
  pcgx_status rc = PCGX_OK;
  // At most MaxIteration evaluations can happen (updater.go:69-70); once the
  // device-side state is `done` the remaining launches return immediately.
  uint32_t mail_seq = 0u;
  if (small_now(s) && s->max_iteration >= 1) rc = small_steps(s, ctx().stream, s->max_iteration, &mail_seq);  // (small clouds: the whole loop in one launch)
  else
    for (int it = 0; it < s->max_iteration && rc == PCGX_OK; it++) rc = pcgx_icp_session_step(s, nullptr);
  const double t2 = trace ? now_us() : 0.0;
  if (rc == PCGX_OK && mail_seq != 0u) {  // (the launch mails the loop state itself)
    IcpState h;
    rc = mailbox_wait_tagged(mail_seq, (int)(sizeof h / 4), reinterpret_cast<uint32_t *>(&h), ctx().stream);
    if (rc == PCGX_OK) rc = result_of(s, h, trans16, stat, nullptr);
  } else if (rc == PCGX_OK) {
    rc = pcgx_icp_session_result(s, nullptr, trans16, stat, nullptr);
  }
  const double t3 = trace ? now_us() : 0.0;
  pcgx_icp_session_free(s);
  if (trace)
    fprintf(stderr, "pcgx fit trace: stream %p begin %.0f us: create %.0f, enqueue %.0f, result (wait) %.0f, free %.0f\n", (void *)ctx().stream,
            fmod(t0, 1e8), t1 - t0, t2 - t1, t3 - t2, now_us() - t3);
  return rc;
}

extern "C" pcgx_status pcgx_icp_evaluate_params(const pcgx_kdtree *base, const float *target, int64_t nt,
                                                const pcgx_icp_params *params, pcgx_icp_evaluated *out) {
  PCGX_API_CALL();
  if (!base || !out || !params) return fail(PCGX_E_INVALID, "pcgx_icp_evaluate: NULL argument");
  pcgx_icp_session *s = nullptr;
  PCGX_TRY(pcgx_icp_session_create(base, target, nt, 0, params, nullptr, &s));
  pcgx_status rc = pcgx_icp_session_partials(s, nullptr);
  double sums[S_COUNT];
  if (rc == PCGX_OK) {
    hipError_t e = hipMemcpyAsync(sums, s->d_sums, sizeof sums, hipMemcpyDeviceToHost, ctx().stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx().stream);
    if (e != hipSuccess) rc = fail(PCGX_E_HIP, "pcgx_icp_evaluate: %s", hipGetErrorString(e));
  }
  pcgx_icp_session_free(s);
  if (rc != PCGX_OK) return rc;
  return pcgx_icp_finish_evaluate(sums, params->min_pairs, out);
}

extern "C" pcgx_status pcgx_icp_evaluate(const pcgx_kdtree *base, const float *target, int64_t nt,
                                         float max_dist, float min_dist_sq, int32_t min_pairs,
                                         pcgx_icp_evaluated *out) {
  pcgx_icp_params p;
  memset(&p, 0, sizeof p);
  p.max_dist = max_dist;
  p.min_dist_sq = min_dist_sq;
  p.min_pairs = min_pairs;
  return pcgx_icp_evaluate_params(base, target, nt, &p, out);
}

extern "C" pcgx_status pcgx_icp_pairs(const pcgx_kdtree *base, const float *target, int64_t nt,
                                      float max_dist, float min_dist_sq, int64_t *base_id,
                                      int64_t *target_id, float *dist_sq, int64_t *npairs) {
  PCGX_API_CALL();
  if (!base || !npairs || nt < 0 || (nt > 0 && (!target || !base_id || !target_id || !dist_sq)))
    return fail(PCGX_E_INVALID, "pcgx_icp_pairs: bad argument");
  *npairs = 0;
  if (nt == 0) return PCGX_OK;
  std::vector<int64_t> ids((size_t)nt);
  std::vector<float> dsq((size_t)nt);
  PCGX_TRY(pcgx_kdtree_nearest_batch(base, target, nt, max_dist, min_dist_sq, ids.data(), dsq.data()));
  int64_t m = 0;
  for (int64_t i = 0; i < nt; i++) {  // order-preserving compaction (correspondence.go:25-36)
    if (ids[i] < 0) continue;
    base_id[m] = ids[i];
    target_id[m] = i;
    dist_sq[m] = dsq[i];
    m++;
  }
  *npairs = m;
  return PCGX_OK;
}

// Measurement aid: see include/pcgx.h.
extern "C" pcgx_status pcgx_debug_icp_grid_stats(pcgx_icp_session *s, void *stream, int64_t out[6]) {
  PCGX_API_LOCK();
  if (!s || !out) return fail(PCGX_E_INVALID, "pcgx_debug_icp_grid_stats: NULL argument");
  out[0] = s->nt;
  out[1] = out[2] = out[3] = out[4] = out[5] = 0;
  if (s->patched || !grid_enabled(s->base) || s->kp.min_dist_sq > 0.0f || s->nt == 0) return PCGX_OK;
  hipStream_t st = pick_stream(stream);
  s->touch(st);
  PCGX_TRY(settle(s, st));  // (steps enqueued without the leftover walk: enqueue_corr)
  PCGX_TRY(general_prepare(s, st));
  unsigned long long *d_trace = nullptr;
  PCGX_HIP_TRY(dev_cache_alloc((void **)&d_trace, 40 * sizeof(unsigned long long)));
  PCGX_HIP_TRY(hipMemsetAsync(d_trace, 0, 40 * sizeof(unsigned long long), st));
  const float *x = s->d_xyz, *y = s->d_xyz + s->nt, *z = s->d_xyz + 2 * s->nt;
  const unsigned gb = (unsigned)((s->nt + kIcpGridBlock - 1) / kIcpGridBlock);
  hipLaunchKernelGGL((icp_grid_kernel<false, true>), dim3(xcd_grid(gb)), dim3(kIcpGridBlock), 0, st, s->base->grid, x, y, z, s->nt,
                     s->d_state, s->kp, s->d_match, s->d_match_id, (const float4 *)s->d_normals, s->d_first_leaf,
                     s->d_walk_list, s->d_walk_count, (uint32_t)s->grid, s->d_partials, d_trace, (const uint32_t *)nullptr,
                     (float4 *)nullptr, s->base->grid.cert ? s->d_match_cert : nullptr);
  unsigned long long h[40];
  hipError_t e = hipMemcpyAsync(h, d_trace, sizeof h, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  dev_cache_free(d_trace);
  if (e != hipSuccess) return fail(PCGX_E_HIP, "pcgx_debug_icp_grid_stats: %s", hipGetErrorString(e));
  out[1] = (int64_t)h[0];
  out[2] = (int64_t)h[1];
  out[3] = (int64_t)h[2];
  out[4] = (int64_t)h[3];
  out[5] = (int64_t)h[36];
  if (getenv("PCGX_GRID_TRACE_PRINT"))
    for (int b = 0; b < 2; b++) {
      fprintf(stderr, "%d-segment scan per target: none %.4f, rounds of 4:", b ? 9 : 4, (double)h[4 + 16 * b] / (double)s->nt);
      for (int k = 1; k < 16; k++) fprintf(stderr, " %d:%.4f", k - 1, (double)h[4 + 16 * b + k] / (double)s->nt);
      fprintf(stderr, "\n");
    }
  return PCGX_OK;
}
