// strict_terms.h -- what the kernels of the strict sums share with the correspondence kernels (icp.hip):
// the work descriptor, the nine float32 terms of a pair formed exactly as evaluator.go:122-145 forms
// them, and the float64 tile sums the summary kernel's guesses start from.  Device code only.
#pragma once
#include "pcgx_internal.h"
#include "strict_sum.h"

namespace pcgx {
using namespace ss;

constexpr int kStrictRows = 9;  // Value, G0..G5, DistRMS, sum of weights (evaluator.go:132-144)


// (by DPP: a partner's value is a register move -- row_shr:n inside a row of 16 lanes, row_bcast:15 / :31 from a row's
// last lane to the rows behind it -- where __shfl is a trip through the LDS crossbar and a wait; lane 63 ends with the
// sum of all lanes, every lane reads it from there)
// (moves inside a row of 16 lanes, or by one lane over the wave: a lane without a partner gets 0.0 whatever `old` is --
// bound_ctrl -- and the move needs no `old` set up in front of it; the broadcasts leave the rows they do not write at `old`)
template <int kCtrl, int kRowMask>
__device__ __forceinline__ double dpp_f64(double old, double v) {
  constexpr bool kZeroFill = kRowMask == 0xf;  // (every caller's `old` is 0.0 there)
  const unsigned long long o = (unsigned long long)__double_as_longlong(old), x = (unsigned long long)__double_as_longlong(v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)o, (int)(uint32_t)x, kCtrl, kRowMask, 0xf, kZeroFill);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(o >> 32), (int)(uint32_t)(x >> 32), kCtrl, kRowMask, 0xf, kZeroFill);
  return __longlong_as_double((long long)((unsigned long long)hi << 32 | lo));
}
__device__ __forceinline__ double lane_f64(double v, int src) {  // lane `src`'s value (a wave-uniform lane number), in every lane
  const unsigned long long x = (unsigned long long)__double_as_longlong(v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, src), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), src);
  return __longlong_as_double((long long)((unsigned long long)hi << 32 | lo));
}
__device__ __forceinline__ double wave_allsum_f64(double v) {
  v += dpp_f64<0x111, 0xf>(0.0, v);
  v += dpp_f64<0x112, 0xf>(0.0, v);
  v += dpp_f64<0x114, 0xf>(0.0, v);
  v += dpp_f64<0x118, 0xf>(0.0, v);
  v += dpp_f64<0x142, 0xa>(0.0, v);
  v += dpp_f64<0x143, 0xc>(0.0, v);
  return lane_f64(v, 63);
}

struct LeafAux {  // per leaf of a tile with a level crossing: compositions of leaves 0..l and l..63
  Summary pre, suf;
};
static_assert(sizeof(LeafAux) == 96, "two summaries");

// per leaf of a tile WITHOUT a window (a sum hovering around zero): the leaf's own window and the
// composition of the leaves from the head of its run (neighbouring leaves of equal windows) up to it
struct LeafRec {
  int32_t key;  // -1: the leaf has no window (its 32 terms are added one by one)
  int32_t pad[3];
  Summary run;
};
static_assert(sizeof(LeafRec) == 64 && sizeof(LeafRec) <= sizeof(LeafAux), "a leaf record fits a LeafAux");

struct JobDesc {  // a (row, tile) that crosses a level or has no window: strict_sum_kernel -> strict_job_kernel
  int64_t tile;
  int32_t row, pad;
};

// Slots are handed out by kAuxShards counters a cache line apart (shard = (tile + 7 row) % kAuxShards, each with
// naux / kAuxShards slots): one counter for all was ~100 returning atomics on one address per launch, served one
// after the other -- the last job of a launch waited 30-40 us for its slot number.
constexpr int kAuxShards = 64;

// Candidate start states of a job tile (strict.hip, strict_job_kernel): candidate i stands cand_offset(i) floats above
// the tile's guessed start state in magnitude.  The table has room for every float within 384 of the guess; a level
// crossing fills the 256 in the middle (its guess is a few floats off), a tile without a window all of it (the guess
// of a tile behind other such tiles is 30-100 floats off: their rounding errors are only known approximately).
// (Measured: wider, sparser candidates with "a state between two candidates that end in the same float ends there
// too" -- x -> fl(x + t) is monotone -- caught one tile in ten of those the dense ones missed: around a guess the
// tile's map is one-to-one nearly everywhere.)
constexpr int kCand = 768;
#ifdef PCGX_CAND_INNER
constexpr int kCandInner = PCGX_CAND_INNER;
#else
constexpr int kCandInner = 256;        // what one wave carries: four candidates per lane
#endif
constexpr uint32_t kCandReach = 512u;  // > |cand_offset|: the guess must be that far from zero (and from infinity)
__host__ __device__ __forceinline__ int32_t cand_offset(int i) { return i - kCand / 2; }

struct StrictWork {
  const float *xyz_caller;      // [nt][3] the targets in the caller's order
  double *tile_sum;             // [9][ntiles] float64 sums of the tiles' terms
  double *tile_err;             // [9][ntiles] rounding error the float32 chain makes inside the tile (from its guess chains)
  double *tile_pub;             // [16][ntiles_pad] the same tile sums, written write-through by the tile's workgroup of
                                // strict_sum_kernel and read by the later tiles' workgroups of the SAME launch
  unsigned int *tile_arrived;   // arrival bits of that exchange, every word on a 128-byte line of its own: word g (at
                                // [32 g]) bit t % 32 = tile 32 g + t has published; behind them, at [32 (ngroups + G)],
                                // bit g % 32 of word G = g / 32: group g is complete (zeroed by the chain kernel)
  int64_t ntiles_pad;           // ntiles rounded up to a multiple of 128 (16-byte loads of two tiles per lane stay inside a row)
  uint32_t *tile_pairs;         // [ntiles] matched targets of the tiles
  TileRec *recs;                // [9][ntiles]
  LeafAux *aux;                 // [naux][64]: what the chain kernel needs to recompute a tile that owns a slot
  float4 *aux_terms;            // [naux][512]: that tile's terms (layout of tile_quad)
  struct JobDesc *jobs;         // [naux]: what strict_job_kernel needs to know about the slot's tile
  uint32_t *cand;               // [naux][kCand]: the tile carried out from kCand start states (cand_offset)
  unsigned int *aux_count;      // [kAuxShards] x 32 words: slots handed out this iteration, per shard (zeroed by the chain kernel)
  unsigned int *done_rows;      // rows of the chain kernel that have finished (ticket of the fused update)
  unsigned long long *dbg;      // [64] counters (measurement aid)
  unsigned long long *stamps;   // [ntiles][8] wall-clock stamps of the summary kernel's workgroups (measurement aid)
  int64_t nt, ntiles;
  int32_t naux;
  int32_t nrows;      // 9, or 8 with the default weight: the sum of the weights is then min(pairs, 2^24) exactly
  int32_t weight_fn;  // evaluator.go:130 (PCGX_WEIGHT_*)
  float weight_a;
  const float *raw_terms;  // testing (pcgx_debug_strict_sum_dev): [9][nt] float32 terms given as they are, no pairs
  int32_t exchange;   // strict_sum_kernel forms the tile sums itself and its workgroups exchange them inside the launch
  // A target spread over ranks (strict_enqueue_sharded): the sequential order is the ranks' tiles one after the other,
  // rank 0's first.  This rank's guesses start from the float64 sums (row_base) and chain rounding errors (err_base)
  // of the ranks before it, its walk from the states they ended in (start_bits), and leaves its own end states in
  // hop_out (the float's bits as a float64: exact under the all-reduce that hands them on).  All nullptr on one GPU.
  const double *row_base;     // [9]
  const double *err_base;     // [9]
  const uint32_t *start_bits; // [9]
  double *hop_out;            // [16]
  int32_t first_exact;        // this rank holds the first tile of the whole target (added up term by term from 0.0f)
  // The chain kernel runs one workgroup per (sum, chunk of 512 tiles): everything but the walk is independent between
  // chunks.  chunk_state[(row * nchunks + c) * 16] = the state row `row` has in front of chunk c, bits in the low word
  // and `epoch` in the high one: ONE 64-bit word written write-through by chunk c - 1's walker and polled by chunk
  // c's -- a word of an earlier launch carries an earlier epoch, nothing is ever reset.
  unsigned long long *chunk_state;
  int32_t nchunks;
  uint32_t epoch;             // this launch's (counted by the host, from 1)
  // A target spread over ranks, the RING form (strict_enqueue_ring): every rank owns an inbox of 64-bit words in
  // host-coherent memory that all GPUs of the node write and poll directly (RingLayout below); a word is a 32-bit
  // payload under a 32-bit tag -- the communicator's step count, `ring_epoch` -- so it says by itself whether it
  // is this step's: no flags, no fences, no resets.  ring == nullptr: one GPU, or the collective form above.
  // (a word's tag, `ring_epoch` for this step: {the communicator's Fit number, step + 1}, kRingTagStepBits for the step)
  unsigned long long *ring;   // the ranks' ABORT words: rank k's at ring + k * ring_words + RingLayout::abort(), in the
  int32_t ring_words;         // host-coherent block every host can write (comm.hip); != nullptr: the ring form
  // the DATA words of the inboxes: ring_tab[k] = rank k's inbox as THIS device addresses it -- the rank's own GPU's
  // memory, mapped by its peers (hipIpcOpenMemHandle between processes, peer access inside one: a store crosses xGMI,
  // the owner polls its own HBM), or, where that cannot be had, its part of the host-coherent block (a PCIe round
  // trip per poll).  ring_mine == ring_tab[rank].
  unsigned long long *const *ring_tab;
  unsigned long long *ring_mine;
  int32_t rank, world;
  uint32_t ring_epoch;
  long long ring_guess_ticks; // bound (100 MHz ticks) of the waits for what only guesses depend on (RingView::guess_ticks)
  double *ring_base;          // device: [0..8] row_base, [16..24] err_base of this step, as fetched from the inbox
  unsigned int *ring_flag;    // device: == ring_epoch once tile 0 of strict_sum_kernel has put row_base up
  int32_t spec_depth; // a walker with at least this many walks in front of it walks ahead of its wait (strict_chain_kernel<., kSpec>)
  int32_t selfcheck;  // bit 0: every step of the chain walk is re-derived term by term and compared (dbg[12..15]);
                      // 1: PCGX_STRICT_TRACE stamps; 2: no candidate tables; 3: wall-clock columns of the counters
};

// ---- the ring (a target spread over ranks, strict_enqueue_ring) -------------------------------------------------
// Inbox of one rank, 64-bit words {payload | tag << 32}; a float64 travels as two words (low, high half):
//   row_tot[k][row][2]  float64 total of rank k's terms of sum `row` (written by rank k < me: my guesses start there)
//   err_tot[k][row][2]  float64 total of the rounding errors rank k's chains make (the same, for the job tiles)
//   start[row], start_pairs[2]   the state the rank before me ended sum `row` in (my walk starts there), and the pairs so far
//   final[row], final_pairs[2]   the state the LAST rank ended in: the sums of the whole target, to every rank
//   abort                        {reason, epoch}: a rank could not go on in that step (or a wait ran out of time)
struct RingLayout {
  int world;
  __host__ __device__ int row_tot(int k, int row) const { return (k * kStrictRows + row) * 2; }
  __host__ __device__ int err_tot(int k, int row) const { return world * 2 * kStrictRows + (k * kStrictRows + row) * 2; }
  __host__ __device__ int start(int row) const { return world * 4 * kStrictRows + row; }
  __host__ __device__ int start_pairs() const { return world * 4 * kStrictRows + kStrictRows; }
  __host__ __device__ int final(int row) const { return world * 4 * kStrictRows + 16 + row; }
  __host__ __device__ int final_pairs() const { return world * 4 * kStrictRows + 16 + kStrictRows; }
  __host__ __device__ int abort() const { return world * 4 * kStrictRows + 32; }
  __host__ __device__ int words() const { return (world * 4 * kStrictRows + 33 + 15) & ~15; }
};
// Waits for what only guesses depend on -- the earlier ranks' float64 totals: a late word costs time, not bits.  A rank
// cannot END its step before the rank in front of it has ended its walk, which lies behind that rank's totals in stream
// order: on a GPU of its own a rank loses nothing by waiting for them as long as it takes (kRingGuessTicksApart: a rank
// that is late by milliseconds costs the others those milliseconds, where giving up after 2 ms cost them the slow paths
// of hundreds of tiles with guesses off: 37 ms a step, round 5's rehearsals).  Ranks that SHARE a device (tests and
// rehearsals on the one-GPU box) keep the short bound: there a waiting launch may hold the places the awaited one needs.
constexpr int kRingTagStepBits = 12;                    // a word's tag: {Fit number (20 bits), step + 1 (12 bits)}, comm.hip ring_tag
constexpr long long kRingGuessTicks = 200000;          // 2 ms
constexpr long long kRingGuessTicksApart = 20000000;   // 200 ms
constexpr long long kRingWalkTicks = 1000000000;  // 10 s: waits for the state a walk starts from (there is no going on without it)

#if defined(__HIPCC__)
__device__ __forceinline__ void ring_put(unsigned long long *w, uint32_t payload, uint32_t epoch) {
  __hip_atomic_store(w, (unsigned long long)epoch << 32 | payload, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void ring_put_f64(unsigned long long *w, double v, uint32_t epoch) {
  const unsigned long long x = (unsigned long long)__double_as_longlong(v);
  ring_put(w, (uint32_t)x, epoch);
  ring_put(w + 1, (uint32_t)(x >> 32), epoch);
}
__device__ __forceinline__ bool ring_peek(const unsigned long long *w, uint32_t epoch, uint32_t &payload) {
  const unsigned long long v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  payload = (uint32_t)v;
  return (uint32_t)(v >> 32) == epoch;
}
// the abort word of MY inbox: set in this step or an earlier one of THIS Fit (hosts run ahead of devices: a rank that
// fails while enqueuing step 12 tells the others' step 7).  Another Fit's number in the tag: what an earlier Fit on the
// communicator left, or a laggard of that Fit raised late -- not this Fit's business (comm.hip, comm_ring_new_fit).
__device__ __forceinline__ bool ring_aborted(const StrictWork &W) {
  const RingLayout RL{W.world};
  const unsigned long long v = __hip_atomic_load(W.ring + (size_t)W.rank * W.ring_words + RL.abort(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const uint32_t tag = (uint32_t)(v >> 32);
  return tag != 0u && tag >> kRingTagStepBits == W.ring_epoch >> kRingTagStepBits && tag <= W.ring_epoch;
}
// rank k's inbox (data words), as this device addresses it
__device__ __forceinline__ unsigned long long *ring_inbox(const StrictWork &W, int k) { return W.ring_tab[k]; }
// Wait for word `off` of MY inbox (every lane of the wave: the same address, one request).  0: here; 1: the ring was
// aborted; 2: out of time.
__device__ __forceinline__ int ring_wait(const StrictWork &W, int off, uint32_t &payload, long long max_ticks) {
  const unsigned long long *p = W.ring_mine + off;
  long long t_first = 0;
  for (int spins = 0;; spins++) {
    if (ring_peek(p, W.ring_epoch, payload)) return 0;
    if ((spins & 15) == 15) {  // (a clock read is a memory round trip, a look at the abort word one over PCIe: now and then)
      if (ring_aborted(W)) return 1;
      const long long now = (long long)wall_clock64();
      if (t_first == 0) t_first = now;
      if (now - t_first > max_ticks) return 2;
    }
    __builtin_amdgcn_s_sleep(8);
  }
}
// ... of a float64 (two words)
__device__ __forceinline__ int ring_wait_f64(const StrictWork &W, int off, double &v, long long max_ticks) {
  uint32_t lo = 0u, hi = 0u;
  int rc = ring_wait(W, off, lo, max_ticks);
  if (rc == 0) rc = ring_wait(W, off + 1, hi, max_ticks);
  v = __longlong_as_double((long long)((unsigned long long)hi << 32 | lo));
  return rc;
}
// a wait that ran out of time breaks the ring for everybody (they would wait as long otherwise)
__device__ __forceinline__ void ring_raise_abort(const StrictWork &W, uint32_t reason) {
  const RingLayout RL{W.world};
  for (int k = 0; k < W.world; k++) ring_put(W.ring + (size_t)k * W.ring_words + RL.abort(), reason, W.ring_epoch);
}
#endif

// ---- the terms ------------------------------------------------------------------------------------
// Nothing stores the nine float32 terms of a pair in HBM any more (round 2: 36 MB out of one kernel and
// into the next per iteration): they are formed where they are needed, from the target in the caller's
// order (12 B, coalesced) and its pair (16 B; the correspondence kernels leave every pair in the
// caller's order as well, match_caller; sessions on a patched tree gather through pos_of).
struct TermSrc {
  const float4 *match;
  const uint32_t *pos_of;  // nullptr: match[] is in the caller's order already
  const float *xyz;        // caller's order
  int64_t nt;
  float m[16];
  bool project;  // icp.go:27-30: the first Evaluate sees the raw target
  int32_t weight_fn;
  float weight_a;
  const float *raw;  // testing: the terms themselves, [9][nt] (StrictWork::raw_terms)
};

__device__ __forceinline__ TermSrc make_term_src(const float4 *match, const uint32_t *pos_of, const IcpState *state,
                                                 const StrictWork &W) {
  TermSrc S;
  S.match = match;
  S.pos_of = pos_of;
  S.xyz = W.xyz_caller;
  S.nt = W.nt;
#pragma unroll
  for (int k = 0; k < 16; k++) S.m[k] = state->trans[k];
  S.project = state->iter > 0;
  S.weight_fn = W.weight_fn;
  S.weight_a = W.weight_a;
  S.raw = W.raw_terms;
  return S;
}

// evaluator.go:122-145, every term in float32 as the reference forms it.  Unmatched targets and the
// padding behind nt carry -0.0f: x + (-0.0f) == x for EVERY float x (both zeros included).
__device__ __forceinline__ bool pair_terms(const TermSrc &S, float x0, float y0, float z0, const float4 &b, float *t /* [9] */) {
#pragma unroll
  for (int k = 0; k < kStrictRows; k++) t[k] = -0.0f;
  if (!(b.w >= 0.0f)) return false;  // correspondence.go:27-29
  if (S.raw) {  // testing: load_quad left the target's index in x0
    const int64_t i = (int64_t)__float_as_int(x0);
#pragma unroll
    for (int k = 0; k < kStrictRows; k++) t[k] = S.raw[(int64_t)k * S.nt + i];
    return true;
  }
  if (S.project) {  // icp.go:62-64
    float px, py, pz;
    mat4_transform(S.m, x0, y0, z0, px, py, pz);
    x0 = px; y0 = py; z0 = pz;
  }
  const float x1 = b.x, y1 = b.y, z1 = b.z;
  // The pair's DistSq is formed again here, from the two points, by the expression the searches use (knn_grid.h,
  // knn_walk.h: mat/vec3.go:18-20,38-40) -- the same bits as the search returned, since the operands are the same
  // floats.  b.w only says that there is a pair: a target that keeps its partner from one iteration to the next
  // (icp.hip, the certificate) then has nothing to write into the caller-order copy of the pairs, which was a
  // million scattered 4-byte stores per iteration.  (strict_check.hip adds the stored distances: the cross-check.)
  const float ddx = x1 - x0, ddy = y1 - y0, ddz = z1 - z0;
  const float dsq = (ddx * ddx + ddy * ddy) + ddz * ddz;
  const float w = eval_weight_fn(S.weight_fn, S.weight_a, dsq);  // evaluator.go:130
  t[0] = w * dsq;
  t[1] = w * (x0 - x1);
  t[2] = w * (y0 - y1);
  t[3] = w * (z0 - z1);
  t[4] = w * (z0 * y1 - y0 * z1);
  t[5] = w * (x0 * z1 - z0 * x1);
  t[6] = w * (y0 * x1 - x0 * y1);
  t[7] = w * norm_sq3(x0, y0, z0);
  t[8] = w;
  return true;
}

// the four consecutive targets i0 .. i0 + 3 (i0 a multiple of 4): pairs and coordinates
__device__ __forceinline__ void load_quad(const TermSrc &S, int64_t i0, float4 *bp, float *tx, float *ty, float *tz) {
  if (S.raw) {  // testing: every target below nt is a pair, pair_terms fetches its nine terms by index
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const int64_t i = i0 + c;
      bp[c] = make_float4(0.0f, 0.0f, 0.0f, i < S.nt ? 0.0f : -1.0f);
      tx[c] = __int_as_float((int)i);
      ty[c] = tz[c] = 0.0f;
    }
    return;
  }
  if (i0 + 3 < S.nt) {
    if (S.pos_of) {
      const uint4 p = *reinterpret_cast<const uint4 *>(S.pos_of + i0);
      bp[0] = S.match[p.x]; bp[1] = S.match[p.y]; bp[2] = S.match[p.z]; bp[3] = S.match[p.w];
    } else {
#pragma unroll
      for (int c = 0; c < 4; c++) bp[c] = S.match[i0 + c];
    }
    const float4 *x4 = reinterpret_cast<const float4 *>(S.xyz + 3 * i0);  // 48 B, 16-byte aligned
    const float4 a = x4[0], b = x4[1], d = x4[2];
    tx[0] = a.x; ty[0] = a.y; tz[0] = a.z;
    tx[1] = a.w; ty[1] = b.x; tz[1] = b.y;
    tx[2] = b.z; ty[2] = b.w; tz[2] = d.x;
    tx[3] = d.y; ty[3] = d.z; tz[3] = d.w;
    return;
  }
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const int64_t i = i0 + c;
    bp[c] = make_float4(0.0f, 0.0f, 0.0f, -1.0f);
    tx[c] = ty[c] = tz[c] = 0.0f;
    if (i < S.nt) {
      bp[c] = S.pos_of ? S.match[S.pos_of[i]] : S.match[i];
      tx[c] = S.xyz[3 * i];
      ty[c] = S.xyz[3 * i + 1];
      tz[c] = S.xyz[3 * i + 2];
    }
  }
}

// ---- tile sums ------------------------------------------------------------------------------------
// The float64 sums of the nine rows' terms over tile `tile` (2048 targets in the caller's order): what
// strict_sum_kernel's guesses start from (their prefix over the tiles).  Only guesses depend on them.
// A workgroup of kThreads (256 or 512) threads, consecutive threads consecutive quads; s_part: LDS,
// [kThreads / 64][kStrictRows] doubles.
template <int kThreads>
__device__ __forceinline__ void tile_sums_block(const TermSrc &S, const StrictWork &W, int64_t tile, double (*s_part)[kStrictRows]) {
  constexpr int kQuads = kTile / 4 / kThreads;
  static_assert(kQuads >= 1 && kQuads * kThreads * 4 == kTile, "the block covers the tile");
  double acc[kStrictRows];
#pragma unroll
  for (int k = 0; k < kStrictRows; k++) acc[k] = 0.0;
  float4 bp[kQuads][4];
  float tx[kQuads][4], ty[kQuads][4], tz[kQuads][4];
#pragma unroll
  for (int h = 0; h < kQuads; h++)  // every load of the tile is issued before the first use
    load_quad(S, tile * kTile + 4 * (int64_t)(h * kThreads + threadIdx.x), bp[h], tx[h], ty[h], tz[h]);
#pragma unroll
  for (int h = 0; h < kQuads; h++) {
    float t[4][kStrictRows];
#pragma unroll
    for (int c = 0; c < 4; c++) pair_terms(S, tx[h][c], ty[h][c], tz[h][c], bp[h][c], t[c]);
#pragma unroll
    for (int k = 0; k < kStrictRows; k++) acc[k] += (((double)t[0][k] + (double)t[1][k]) + (double)t[2][k]) + (double)t[3][k];
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < kStrictRows; k++) {
    const double v = wave_allsum_f64(acc[k]);
    if (lane == 0) s_part[wave][k] = v;
  }
  __syncthreads();
  if ((int)threadIdx.x < W.nrows) {
    double v = 0.0;
    for (int w = 0; w < kThreads / 64; w++) v += s_part[w][threadIdx.x];
    W.tile_sum[threadIdx.x * W.ntiles + tile] = v;
  }
  __syncthreads();
}

}  // namespace pcgx
