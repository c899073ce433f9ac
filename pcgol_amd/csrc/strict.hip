// strict.hip -- the evaluator's sums exactly as the reference forms them (sequential float32
// additions over the pairs in target order, pc/registration/icp/evaluator.go:122-145), computed by
// the whole GPU instead of one dependent chain.  Arithmetic and the proof sketch: strict_sum.h.
//
// Per iteration, after the correspondence kernels left match[] (icp.hip):
//   strict_terms_kernel  one thread per target in the CALLER's order: the nine float32 terms
//                        (rows of terms[]), float64 tile sums, level-1 bins (atomics), pair count
//   strict_sum_kernel    one wave per (row, tile): leaf guesses from the float64 prefix (refined once
//                        inside the tile by the rounding errors the chains make from them), class
//                        summaries of the 64 leaves, composed -> one 64-byte record per tile
//   strict_chain_kernel  one workgroup per row: records of equal windows merged into runs
//                        (segmented wave scan), one wave applies them in order to the exact state;
//                        a record that does not cover the state -> that tile is recomputed exactly
// Nothing here is approximate: a record is applied only when its interval proves the result.
#include "pcgx_internal.h"
#include "strict_sum.h"

namespace pcgx {
using namespace ss;

constexpr int kStrictRows = 9;  // Value, G0..G5, DistRMS, sum of weights (evaluator.go:132-144)

// ---- wave helpers ---------------------------------------------------------------------------------
__device__ __forceinline__ double wave_allsum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);  // a + b == b + a bit for bit: every lane ends with the same value
  return v;
}
__device__ __forceinline__ double wave_excl_scan_f64(double v, int lane) {
  double inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const double u = __shfl_up(inc, o);
    if (lane >= o) inc += u;
  }
  const double ex = __shfl_up(inc, 1);
  return lane == 0 ? 0.0 : ex;
}
__device__ __forceinline__ uint32_t wave_all_umin(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = umin(v, (uint32_t)__shfl_xor((int)v, o));
  return v;
}
__device__ __forceinline__ uint32_t wave_all_umax(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = umax(v, (uint32_t)__shfl_xor((int)v, o));
  return v;
}
__device__ __forceinline__ Summary shfl_summary(const Summary &S, int src) {
  Summary R;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    R.c[r] = __shfl(S.c[r], src);
    R.lo[r] = __shfl(S.lo[r], src);
    R.hi[r] = __shfl(S.hi[r], src);
  }
  return R;
}
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

struct LeafAux {  // per leaf of a tile with a level crossing: compositions of leaves 0..l and l..63
  Summary pre, suf;
};

struct StrictWork {
  float *terms;                 // [9][ntiles][kLeaf / 4][64 lanes][4]: lane l of a tile holds its leaf's terms 4 by 4
  const float *xyz_caller;      // [nt][3] the targets in the caller's order
  double *tile_sum;             // [9][ntiles] float64 sums of the tiles' terms
  double *bin_sum;              // [9][nbins] level-1 sums of kBinTiles tiles (atomics; zeroed by the chain kernel)
  TileRec *recs;                // [9][ntiles]
  LeafAux *aux;                 // [naux][64]
  unsigned int *aux_count;      // slots handed out this iteration (zeroed by the chain kernel)
  unsigned int *done_rows;      // rows of the chain kernel that have finished (ticket of the fused update)
  unsigned long long *pairs;    // matched targets of this iteration (atomic; zeroed by the chain kernel)
  unsigned long long *dbg;      // [16] counters (measurement aid)
  int64_t nt, nt_pad, ntiles, nbins;
  int32_t naux;
  int32_t weight_fn;  // evaluator.go:130 (PCGX_WEIGHT_*)
  float weight_a;
  int32_t selfcheck;  // debugging: every step of the chain walk is re-derived term by term and compared (dbg[12..15])
};

// term i (caller's order) of a row lives at ((tile * 8 + j / 4) * 64 + l) * 4 + j % 4 with tile = i /
// 2048, l = i % 2048 / 32, j = i % 32: leaves are interleaved 4 floats at a time so that the 64 lanes
// of a wave read (and the terms kernel writes) them with fully coalesced 16-byte accesses
__device__ __forceinline__ void load_leaf(const float *__restrict__ row, int64_t tile, int lane, float *t) {
  const float4 *q = reinterpret_cast<const float4 *>(row) + tile * (kLeaf / 4) * kLanes + lane;
#pragma unroll
  for (int v = 0; v < kLeaf / 4; v++) {
    const float4 a = q[v * kLanes];
    t[4 * v] = a.x; t[4 * v + 1] = a.y; t[4 * v + 2] = a.z; t[4 * v + 3] = a.w;
  }
}
__device__ __forceinline__ double leaf_sum_f64(const float *t) {
  double v = 0.0;
#pragma unroll
  for (int j = 0; j < kLeaf; j++) v += (double)t[j];
  return v;
}

// float64 prefix of `tile` from the level-1 bins and the tile sums inside its bin; every lane gets it
__device__ __forceinline__ double tile_prefix(const double *__restrict__ tile_v, const double *__restrict__ bin_v,
                                              int64_t ntiles, int64_t nbins, int row, int64_t tile, int lane) {
  const int64_t bin = tile / kBinTiles;
  double v = 0.0;
  for (int64_t b = lane; b < bin; b += 64) v += bin_v[row * nbins + b];
  const int64_t t = bin * kBinTiles + lane;
  if (t < tile) v += tile_v[row * ntiles + t];
  return wave_allsum_f64(v);
}

// ---- terms ---------------------------------------------------------------------------------------
// One workgroup per tile of 2048 targets in the caller's order: the target itself is read in that
// order (xyz_caller) and re-projected here exactly as the correspondence kernels do, only its pair
// is a gather (pos_of: where the session keeps target i).  Unmatched targets and the padding behind
// nt carry -0.0f: x + (-0.0f) == x for every x.
constexpr int kTermsBlock = 256;
// One workgroup per tile; a thread forms the terms of two QUADS (4 consecutive targets of one leaf):
// its eight gathers are issued together and each row's four terms leave as one 16-byte store, lanes
// side by side (the interleaved layout of terms[]).
__global__ __launch_bounds__(kTermsBlock) void strict_terms_kernel(const float4 *__restrict__ match,
                                                                   const uint32_t *__restrict__ pos_of,
                                                                   const IcpState *__restrict__ state, StrictWork W) {
  __shared__ double s_part[kTermsBlock / 64][kStrictRows];
  __shared__ int s_pairs[kTermsBlock / 64];
  if (state->done) return;
  float m[16];
#pragma unroll
  for (int k = 0; k < 16; k++) m[k] = state->trans[k];
  const bool project = state->iter > 0;  // icp.go:27-30: the first Evaluate sees the raw target
  const int64_t tile = blockIdx.x;
  double acc[kStrictRows];
#pragma unroll
  for (int k = 0; k < kStrictRows; k++) acc[k] = 0.0;
  int npairs = 0;
  constexpr int kQuads = kTile / 4 / kTermsBlock;  // 2
  float4 bp[kQuads][4];
  float tx[kQuads][4], ty[kQuads][4], tz[kQuads][4];
#pragma unroll
  for (int h = 0; h < kQuads; h++) {
    const int q = h * kTermsBlock + threadIdx.x, l = q / (kLeaf / 4), v = q % (kLeaf / 4);  // consecutive threads, consecutive targets
    const int64_t i0 = tile * kTile + l * kLeaf + 4 * v;
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const int64_t i = i0 + c;
      bp[h][c] = make_float4(0.0f, 0.0f, 0.0f, -1.0f);
      tx[h][c] = ty[h][c] = tz[h][c] = 0.0f;
      if (i < W.nt) {
        bp[h][c] = pos_of ? match[pos_of[i]] : match[i];  // (nullptr: match[] is in the caller's order already)
        tx[h][c] = W.xyz_caller[3 * i];
        ty[h][c] = W.xyz_caller[3 * i + 1];
        tz[h][c] = W.xyz_caller[3 * i + 2];
      }
    }
  }
#pragma unroll
  for (int h = 0; h < kQuads; h++) {
    const int q = h * kTermsBlock + threadIdx.x, l = q / (kLeaf / 4), v = q % (kLeaf / 4);  // consecutive threads, consecutive targets
    float t[kStrictRows][4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
#pragma unroll
      for (int k = 0; k < kStrictRows; k++) t[k][c] = -0.0f;
      const float4 b = bp[h][c];
      if (b.w >= 0.0f) {  // correspondence.go:27-29
        npairs++;
        float x0 = tx[h][c], y0 = ty[h][c], z0 = tz[h][c];
        if (project) {
          float px, py, pz;
          mat4_transform(m, x0, y0, z0, px, py, pz);
          x0 = px; y0 = py; z0 = pz;
        }
        const float x1 = b.x, y1 = b.y, z1 = b.z;
        const float w = eval_weight_fn(W.weight_fn, W.weight_a, b.w);  // evaluator.go:130
        t[0][c] = w * b.w;
        t[1][c] = w * (x0 - x1);
        t[2][c] = w * (y0 - y1);
        t[3][c] = w * (z0 - z1);
        t[4][c] = w * (z0 * y1 - y0 * z1);
        t[5][c] = w * (x0 * z1 - z0 * x1);
        t[6][c] = w * (y0 * x1 - x0 * y1);
        t[7][c] = w * norm_sq3(x0, y0, z0);
        t[8][c] = w;
      }
    }
    float4 *out = reinterpret_cast<float4 *>(W.terms) + (tile * (kLeaf / 4) + v) * kLanes + l;
#pragma unroll
    for (int k = 0; k < kStrictRows; k++) {
      out[(int64_t)k * (W.nt_pad / 4)] = make_float4(t[k][0], t[k][1], t[k][2], t[k][3]);
      acc[k] += (((double)t[k][0] + (double)t[k][1]) + (double)t[k][2]) + (double)t[k][3];
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < kStrictRows; k++) {
    const double v = wave_allsum_f64(acc[k]);
    if (lane == 0) s_part[wave][k] = v;
  }
  {
    int p = npairs;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) p += __shfl_xor(p, o);
    if (lane == 0) s_pairs[wave] = p;
  }
  __syncthreads();
  if (threadIdx.x < kStrictRows) {
    double v = 0.0;
    for (int w = 0; w < kTermsBlock / 64; w++) v += s_part[w][threadIdx.x];
    W.tile_sum[threadIdx.x * W.ntiles + tile] = v;
    unsafeAtomicAdd(&W.bin_sum[threadIdx.x * W.nbins + tile / kBinTiles], v);  // global_atomic_add_f64, no CAS loop
  } else if (threadIdx.x == 64) {
    int p = 0;
    for (int w = 0; w < kTermsBlock / 64; w++) p += s_pairs[w];
    if (p) atomicAdd(W.pairs, (unsigned long long)p);
  }
}

// the targets in the caller's order, from the session's Morton-ordered SoA copy (once per session)
__global__ __launch_bounds__(256) void strict_xyz_caller_kernel(const float *__restrict__ tx, const float *__restrict__ ty,
                                                                const float *__restrict__ tz,
                                                                const uint32_t *__restrict__ pos_of, int64_t nt,
                                                                float *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nt) return;
  const uint32_t pos = pos_of[i];
  out[3 * i] = tx[pos];
  out[3 * i + 1] = ty[pos];
  out[3 * i + 2] = tz[pos];
}

// ---- summaries ---------------------------------------------------------------------------------------
// leaf guesses of a tile whose first state is (about) base: float64 prefix of the leaf sums, then one
// refinement with the prefix of the rounding errors the chains make from those guesses
__device__ __forceinline__ void tile_guesses(const float *t, double base, int lane, uint32_t &g, ChainRange &cr) {
  const double lsum = leaf_sum_f64(t);
  const double pre = wave_excl_scan_f64(lsum, lane);
  g = f2u((float)(base + pre));
  cr = guess_chain(t, g);
  const double err = ((double)u2f(cr.end) - (double)u2f(g)) - lsum;
  const double epre = wave_excl_scan_f64(err, lane);
  const uint32_t g2 = f2u((float)(base + pre + epre));
  // (guesses a couple of ulps off are as good: the intervals are thousands wide except next to a level)
  const int32_t moved = (int32_t)g2 - (int32_t)g;
  if (__ballot(moved > 2 || moved < -2) != 0ull) {  // uniform
    g = g2;
    cr = guess_chain(t, g);
  }
}

// the 32 additions of leaf l, one after the other, on the state every lane holds
__device__ __forceinline__ uint32_t serial_leaf(uint32_t s, const float *t, int l) {
  float x = u2f(s);
#pragma unroll
  for (int j = 0; j < kLeaf; j++) x = x + u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(t[j]), l));
  return f2u(x);
}

// all 2048 additions of a tile, one after the other: the terms go to LDS in their order and every
// lane runs the same chain over them (broadcast reads; the next 32 terms are read while the 32
// additions on the current ones run)
// a tile's terms in their order in LDS (every lane its leaf)
__device__ __forceinline__ void stage_tile(const float *t, int lane, float *lds /* [kTile] of this wave */) {
  float4 *w4 = reinterpret_cast<float4 *>(lds) + lane * (kLeaf / 4);
#pragma unroll
  for (int v = 0; v < kLeaf / 4; v++) w4[v] = make_float4(t[4 * v], t[4 * v + 1], t[4 * v + 2], t[4 * v + 3]);
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's own LDS writes have landed
}

// the additions of staged terms [first, first + count) (multiples of 64), one after the other: every
// lane runs the same chain (broadcast reads; the next 32 terms are read while the 32 additions on
// the current ones run)
__device__ __forceinline__ uint32_t serial_span(uint32_t s, const float *lds, int first, int count) {
  const float4 *r4 = reinterpret_cast<const float4 *>(lds) + first / 4;
  const int n4 = count / 4;
  float x = u2f(s);
  float4 a[8], b[8];
#pragma unroll
  for (int u = 0; u < 8; u++) a[u] = r4[u];
  for (int k = 0; k < n4; k += 16) {
#pragma unroll
    for (int u = 0; u < 8; u++) b[u] = r4[k + 8 + u];
#pragma unroll
    for (int u = 0; u < 8; u++) x = (((x + a[u].x) + a[u].y) + a[u].z) + a[u].w;
    const int kn = k + 16 < n4 ? k + 16 : 0;
#pragma unroll
    for (int u = 0; u < 8; u++) a[u] = r4[kn + u];
#pragma unroll
    for (int u = 0; u < 8; u++) x = (((x + b[u].x) + b[u].y) + b[u].z) + b[u].w;
  }
  return f2u(x);
}

// all 2048 additions of a tile
__device__ __forceinline__ uint32_t serial_tile(uint32_t s, const float *t, int lane, float *lds /* [kTile] of this wave */) {
  stage_tile(t, lane, lds);
  const uint32_t r = serial_span(s, lds, 0, kTile);
  __builtin_amdgcn_wave_barrier();
  return r;
}

// tiles at the start of every sum that are added up term by term in the summary kernel (by the wave
// of tile 0, one after the other, beside the other waves' work).  A sum starts at 0.0f and runs
// through a new binade every few terms, so no window holds its first tile.  More than one such tile
// was measured (4: the summary kernel's slowest wave then outlasts the rest of it, 151.9 vs 147.7 us
// per iteration at C4; the tiles where a sum hovers around zero are not the first ones)
constexpr int kExactTiles = 1;

__global__ __launch_bounds__(256) void strict_sum_kernel(const IcpState *__restrict__ state, StrictWork W) {
  if (state->done) return;
  const int lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= kStrictRows * W.ntiles) return;  // whole wave
  // the nine waves of the first tiles take longest (they carry out kExactTiles x 2048 additions): they go first
  int row;
  int64_t tile;
  if (w < kStrictRows) {
    row = (int)w;
    tile = 0;
  } else {
    row = (int)((w - kStrictRows) / (W.ntiles - 1));
    tile = 1 + (w - kStrictRows) % (W.ntiles - 1);
  }
  __shared__ float s_tile[4][kTile];
  float t[kLeaf];
  load_leaf(W.terms + (int64_t)row * W.nt_pad, tile, lane, t);
  TileRec T;
  T.s = summary_identity();
  if (tile < kExactTiles) {
    if (tile != 0) return;  // done by the wave of tile 0
    // from the one state known in advance (0.0f, evaluator.go:122) the additions are simply carried
    // out, here, off the chain kernel's critical path -> point records
    uint32_t state = f2u(0.0f);
    for (int64_t k = 0; k < kExactTiles && k < W.ntiles; k++) {
      if (k > 0) load_leaf(W.terms + (int64_t)row * W.nt_pad, k, lane, t);
      const uint32_t next = serial_tile(state, t, lane, s_tile[threadIdx.x >> 6]);
      T.key = -1;
      T.in = state;
      T.out = next;
      T.cons = 1;
      if (lane == 0) W.recs[row * W.ntiles + k] = T;
      state = next;
    }
    return;
  }
  const double P0 = tile_prefix(W.tile_sum, W.bin_sum, W.ntiles, W.nbins, row, tile, lane);
  uint32_t g;
  ChainRange cr;
  tile_guesses(t, P0, lane, g, cr);
  // window of the tile
  const uint32_t mn = wave_all_umin(cr.mn), mx = wave_all_umax(cr.mx);
  const bool one_sign = __ballot(cr.sg_or != cr.sg_and) == 0ull &&
                        (__ballot(cr.sg_or != 0u) == 0ull || __ballot(cr.sg_or == 0u) == 0ull);
  const uint32_t g_first = (uint32_t)rfl((int)g);
  const int32_t key = one_sign ? choose_window(mn, mx, g_first >> 31, g_first & 0x7fffffffu) : -1;
  // point record: the guess chains join up exactly
  const uint32_t g_next = (uint32_t)__shfl_down((int)g, 1);
  const bool cons = __ballot(lane < 63 && g_next != cr.end) == 0ull;
  T.key = key;
  T.in = g_first;
  T.out = (uint32_t)__shfl((int)cr.end, 63);
  T.cons = cons ? 1 : 0;
  if (key >= 0) {  // uniform
    const bool one_binade = (cr.mn >> 23) == (cr.mx >> 23);
    Summary S;
    if (__ballot(!one_binade) != 0ull) {  // uniform: some leaf crosses the level
      if (one_binade) S = leaf_summary_binade(t, g, key);
      else S = leaf_summary_general(t, g, key);
      // this is where a record is most likely not to cover the true state (a landing next to the
      // level): keep the compositions of leaves 0..l and l..63, so that the chain kernel finds the
      // leaf in one parallel step and carries on behind it
      Summary P = S, Q = S;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const Summary X = shfl_summary(P, lane - o);
        if (lane >= o) P = compose(X, P);
        const Summary Y = shfl_summary(Q, lane + o);
        if (lane + o < 64) Q = compose(Q, Y);
      }
      T.s = shfl_summary(P, 63);
      unsigned slot = 0xffffffffu;
      if (lane == 0) slot = atomicAdd(W.aux_count, 1u);
      slot = (unsigned)rfl((int)slot);
      if (slot < (unsigned)W.naux) {
        LeafAux *a = W.aux + (size_t)slot * kLanes + lane;
        a->pre = P;
        a->suf = Q;
        T.cons |= (int32_t)(slot + 1u) << 8;
      }
    } else {
      S = leaf_summary_binade(t, g, key);
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {  // ordered reduction over the 64 leaves
        const Summary Y = shfl_summary(S, lane + o);
        if ((lane & (2 * o - 1)) == 0) S = compose(S, Y);
      }
      T.s = S;
    }
  } else {
    // no window holds the tile (its sum changes sign, or runs through three binades: sums that hover
    // around zero): four records of 16 leaves each, a window of its own for each that has one --
    // the chain kernel then carries out 512 additions per quarter without one instead of 2048
    const unsigned long long mixed = __ballot(cr.sg_or != cr.sg_and), neg = __ballot(cr.sg_or != 0u);
    const int q = lane >> 4;
    const uint32_t qmixed = (uint32_t)(mixed >> (16 * q)) & 0xffffu, qneg = (uint32_t)(neg >> (16 * q)) & 0xffffu;
    uint32_t mnq = cr.mn, mxq = cr.mx;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
      mnq = umin(mnq, (uint32_t)__shfl_xor((int)mnq, o));
      mxq = umax(mxq, (uint32_t)__shfl_xor((int)mxq, o));
    }
    const uint32_t gq = (uint32_t)__shfl((int)g, q * 16);
    int32_t kq = -1;
    if (qmixed == 0u && (qneg == 0u || qneg == 0xffffu)) kq = choose_window(mnq, mxq, qneg ? 1u : 0u, gq & 0x7fffffffu);
    Summary S = summary_identity();
    if (kq >= 0) {
      if ((cr.mn >> 23) == (cr.mx >> 23)) S = leaf_summary_binade(t, g, kq);
      else S = leaf_summary_general(t, g, kq);
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {  // ordered composition over the 16 leaves of a quarter
      const Summary Y = shfl_summary(S, lane + o);
      if ((lane & (2 * o - 1)) == 0) S = compose(S, Y);
    }
    unsigned slot = 0xffffffffu;
    if (lane == 0) slot = atomicAdd(W.aux_count, 1u);
    slot = (unsigned)rfl((int)slot);
    if (slot < (unsigned)W.naux) {
      if ((lane & 15) == 0) {
        TileRec Q;
        Q.key = kq;
        Q.in = Q.out = 0u;
        Q.cons = 0;
        Q.s = S;
        reinterpret_cast<TileRec *>(W.aux + (size_t)slot * kLanes)[q] = Q;
      }
      T.cons |= (int32_t)(slot + 1u) << 8;
    }
  }
  if (lane == 0) W.recs[row * W.ntiles + tile] = T;
}

// ---- chain -----------------------------------------------------------------------------------------
__device__ __forceinline__ bool apply_point(uint32_t &s, const TileRec &R) {
  if ((R.cons & 1) && R.in == s) {
    s = R.out;
    return true;
  }
  return false;
}

// One tile, exactly, from the known state s (every lane holds it; returns it in every lane).
// With aux (a tile with a level crossing whose record did not cover s): the first leaf whose prefix
// composition fails is found by all lanes at once, that leaf is added term by term, and the suffix
// composition behind it finishes the tile.  Without: the 2048 additions, one after the other.
__device__ __forceinline__ uint32_t resolve_tile(uint32_t s, const float *__restrict__ row_terms, int64_t tile, int32_t key,
                                             const LeafAux *__restrict__ aux, int lane, unsigned long long *dbg,
                                             float *lds) {
  const long long t_begin = wall_clock64();
  float t[kLeaf];
  load_leaf(row_terms, tile, lane, t);
  int serial = 0;
  int l = 0;
  if (aux && key < 0) {
    // a tile without a window: its four quarter records (strict_sum_kernel), 64 words, one per lane
    const int32_t w = reinterpret_cast<const int32_t *>(aux)[lane];
    stage_tile(t, lane, lds);
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int32_t kq = __builtin_amdgcn_readlane(w, 16 * q);
      Summary S;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        S.c[k] = __builtin_amdgcn_readlane(w, 16 * q + 4 + k);
        S.lo[k] = __builtin_amdgcn_readlane(w, 16 * q + 8 + k);
        S.hi[k] = __builtin_amdgcn_readlane(w, 16 * q + 12 + k);
      }
      if (kq >= 0 && apply(s, kq, S)) continue;
      s = serial_span(s, lds, q * (kTile / 4), kTile / 4);
      serial += kLanes / 4;
    }
    __builtin_amdgcn_wave_barrier();
  } else if (aux) {
    const LeafAux A = aux[lane];
    uint32_t mine = s;
    const bool ok = apply(mine, key, A.pre);
    const unsigned long long bad = __ballot(!ok);
    l = bad ? __builtin_ctzll(bad) : kLanes;  // leaves 0 .. l-1 are covered
    if (l > 0) s = (uint32_t)__builtin_amdgcn_readlane((int)mine, l - 1);
    while (l < kLanes) {
      s = serial_leaf(s, t, l);
      serial++;
      l++;
      if (l == kLanes) break;
      uint32_t rest = s;
      const bool ok2 = apply(rest, key, A.suf);  // lane l: leaves l..63
      const int okl = __builtin_amdgcn_readlane(ok2 ? 1 : 0, l);
      if (okl) {
        s = (uint32_t)__builtin_amdgcn_readlane((int)rest, l);
        break;
      }
    }
  } else {
    s = serial_tile(s, t, lane, lds);
    serial = kLanes;
  }
  if (lane == 0) {
    atomicAdd(&dbg[2], 1ull);
    atomicAdd(&dbg[3], (unsigned long long)serial);
    if (!aux) atomicAdd(&dbg[5], 1ull);
    atomicAdd(&dbg[aux ? 10 : 11], (unsigned long long)(wall_clock64() - t_begin));
  }
  return s;
}

// debugging aid (W.selfcheck): the state after tiles [a, b) from `before`, term by term; mismatches
// against what the walk produced are counted per path in dbg[12 + path]
__device__ __forceinline__ void selfcheck(const StrictWork &W, const float *row_terms, uint32_t before, uint32_t after,
                                          int64_t a, int64_t b, int path, int lane, float *lds) {
  if (!W.selfcheck) return;
  uint32_t x = before;
  for (int64_t k = a; k < b; k++) {
    float t[kLeaf];
    load_leaf(row_terms, k, lane, t);
    x = serial_tile(x, t, lane, lds);
    __builtin_amdgcn_wave_barrier();
  }
  if (x != after && lane == 0) {
    const unsigned long long k = atomicAdd(&W.dbg[12 + path], 1ull);
    if (0) {  // details of the first few: dbg[16 + 4k ..] = {a | b << 32, before | after << 32, expected, blockIdx}
      W.dbg[16 + 4 * k] = (unsigned long long)a | ((unsigned long long)b << 32);
      W.dbg[17 + 4 * k] = (unsigned long long)before | ((unsigned long long)after << 32);
      W.dbg[18 + 4 * k] = x;
      W.dbg[19 + 4 * k] = blockIdx.x;
    }
  }
}

constexpr int kChainBlock = 512;

// a record every lane of the walking wave reads from the same LDS address, as scalars
__device__ __forceinline__ TileRec load_rec_uniform(const TileRec *p) {
  const int4 *w4 = reinterpret_cast<const int4 *>(p);
  const int4 a = w4[0], b = w4[1], c = w4[2], d = w4[3];  // four 16-byte LDS reads in flight, one wait
  TileRec R;
  R.key = rfl(a.x);
  R.in = (uint32_t)rfl(a.y);
  R.out = (uint32_t)rfl(a.z);
  R.cons = rfl(a.w);
  R.s.c[0] = rfl(b.x); R.s.c[1] = rfl(b.y); R.s.c[2] = rfl(b.z); R.s.c[3] = rfl(b.w);
  R.s.lo[0] = rfl(c.x); R.s.lo[1] = rfl(c.y); R.s.lo[2] = rfl(c.z); R.s.lo[3] = rfl(c.w);
  R.s.hi[0] = rfl(d.x); R.s.hi[1] = rfl(d.y); R.s.hi[2] = rfl(d.z); R.s.hi[3] = rfl(d.w);
  return R;
}

__device__ __forceinline__ TileRec shfl_rec(const TileRec &R, int src) {
  TileRec X;
  X.key = __shfl(R.key, src);
  X.in = (uint32_t)__shfl((int)R.in, src);
  X.out = (uint32_t)__shfl((int)R.out, src);
  X.cons = __shfl(R.cons, src);
  X.s = shfl_summary(R.s, src);
  return X;
}
// X (earlier tiles) then Y, same window
__device__ __forceinline__ TileRec compose_rec(const TileRec &X, const TileRec &Y) {
  TileRec Z;
  Z.key = Y.key;
  Z.s = compose(X.s, Y.s);
  Z.cons = ((X.cons & 1) && (Y.cons & 1) && X.out == Y.in) ? 1 : 0;
  Z.in = X.in;
  Z.out = Y.out;
  return Z;
}

__global__ __launch_bounds__(kChainBlock) void strict_chain_kernel(IcpState *__restrict__ state, StrictWork W,
                                                                  double *__restrict__ sums10, IcpKernelParams kp,
                                                                  int fuse_update) {
  __shared__ TileRec s_rec[kChainBlock];  // the tiles' own records
  __shared__ TileRec s_pre[kChainBlock];  // composition from the tile's run head to the tile
  __shared__ TileRec s_suf[kChainBlock];  // composition from the tile to its run's tail
  __shared__ int16_t s_tail[kChainBlock];  // per wave: the tails of its runs, in order
  __shared__ int16_t s_head[kChainBlock];
  __shared__ int32_t s_count[kChainBlock / 64];
  __shared__ float s_tile[kTile];
  if (state->done) return;
  const int row = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float *row_terms = W.terms + (int64_t)row * W.nt_pad;
  uint32_t s = f2u(0.0f);  // walker state (wave 0), evaluator.go:122: the sums start at zero
  for (int64_t chunk = 0; chunk < W.ntiles; chunk += kChainBlock) {
    // ---- runs of equal windows: segmented scans inside each wave, forwards and backwards
    const long long t_a = wall_clock64();
    const int64_t tile = chunk + threadIdx.x;
    const bool valid = tile < W.ntiles;
    TileRec R;
    if (valid) {
      R = W.recs[row * W.ntiles + tile];
    } else {
      R.key = -2;
      R.in = R.out = 0u;
      R.cons = 0;
      R.s = summary_identity();
    }
    s_rec[threadIdx.x] = R;
    const int32_t key_prev = __shfl_up(R.key, 1), key_next = __shfl_down(R.key, 1);
    const bool head = lane == 0 || R.key < 0 || R.key != key_prev;
    const bool tail = lane == 63 || R.key < 0 || R.key != key_next;
    TileRec P = R, Q = R;
    int fp = head ? 1 : 0, fq = tail ? 1 : 0;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const TileRec X = shfl_rec(P, lane - o);
      const int xf = __shfl_up(fp, o);
      if (lane >= o && !fp) {
        P = compose_rec(X, P);
        fp = xf;
      }
      const TileRec Y = shfl_rec(Q, lane + o);
      const int yf = __shfl_down(fq, o);
      if (lane + o < 64 && !fq) {
        Q = compose_rec(Q, Y);
        fq = yf;
      }
    }
    s_pre[threadIdx.x] = P;
    s_suf[threadIdx.x] = Q;
    const unsigned long long tails = __ballot(tail && valid);
    if (tail && valid) {
      const unsigned long long below = tails & ((1ull << lane) - 1ull);  // the run starts behind the tail before this one
      const int idx = wave * 64 + __popcll(below);
      s_tail[idx] = (int16_t)threadIdx.x;
      s_head[idx] = (int16_t)(wave * 64 + (below ? 64 - __builtin_clzll(below) : 0));
    }
    if (lane == 0) s_count[wave] = __popcll(tails);
    __syncthreads();
    // ---- the walk: one wave, every lane with the same state
    if (wave == 0) {
      const long long t_b = wall_clock64();
      unsigned long long n_run = 0, n_runfail = 0, n_recfail = 0;
      for (int w = 0; w < kChainBlock / 64; w++) {
        const int cnt = s_count[w];
        for (int i = 0; i < cnt; i++) {
          const int e = s_tail[w * 64 + i], h = s_head[w * 64 + i];
          n_run++;
          {
            const TileRec Qr = load_rec_uniform(&s_pre[e]);  // the whole run
            const uint32_t s_in = s;
            if ((Qr.key >= 0 && apply(s, Qr.key, Qr.s)) || apply_point(s, Qr)) {
              selfcheck(W, row_terms, s_in, s, chunk + h, chunk + e + 1, 0, lane, s_tile);
              if (W.selfcheck) {  // the same run composed tile after tile from the tiles' own records
                TileRec acc = load_rec_uniform(&s_rec[h]);
                for (int q = h + 1; q <= e; q++) acc = compose_rec(acc, load_rec_uniform(&s_rec[q]));
                uint32_t y = s_in;
                const bool okc = acc.key >= 0 && apply(y, acc.key, acc.s);
                bool same = acc.key == Qr.key && acc.in == Qr.in && acc.out == Qr.out && (acc.cons & 1) == (Qr.cons & 1);
                for (int r = 0; r < 4; r++) same = same && acc.s.c[r] == Qr.s.c[r] && acc.s.lo[r] == Qr.s.lo[r] && acc.s.hi[r] == Qr.s.hi[r];
                if (lane == 0) {
                  if (!same) atomicAdd(&W.dbg[6], 1ull);          // the wave scan disagrees with the serial composition
                  if (okc && y != s) {
                    const unsigned long long k = atomicAdd(&W.dbg[7], 1ull);  // and gives another result
                    if (k < 2) {
                      uint32_t z = s_in;
                      const int32_t nn = state_to_n(s_in, Qr.key);
                      const bool okq = apply(z, Qr.key, Qr.s);
                      unsigned long long *d = W.dbg + 16 + 16 * k;
                      d[0] = s_in; d[1] = s; d[2] = y; d[3] = (unsigned long long)(uint32_t)Qr.key; d[4] = (unsigned long long)(uint32_t)nn;
                      d[5] = (uint32_t)Qr.s.c[nn & 3]; d[6] = (uint32_t)Qr.s.lo[nn & 3]; d[7] = (uint32_t)Qr.s.hi[nn & 3];
                      d[8] = (uint32_t)acc.s.c[nn & 3]; d[9] = okq; d[10] = z; d[11] = Qr.in; d[12] = Qr.out; d[13] = Qr.cons;
                    }
                  }
                }
              }
              continue;
            }
          }
          n_runfail++;
          // which tile?  lane j tries the composition h .. h + j
          int f = h;
          if (e > h) {
            uint32_t mine = s;
            bool ok = false;
            if (h + lane <= e) {
              const TileRec Pj = s_pre[h + lane];
              ok = Pj.key >= 0 && apply(mine, Pj.key, Pj.s);
            }
            const unsigned long long good = __ballot(ok);
            const int ngood = __builtin_ctzll(~good);  // tiles h .. h + ngood - 1 are covered
            const uint32_t s_in = s;
            if (ngood > 0) s = (uint32_t)__builtin_amdgcn_readlane((int)mine, ngood - 1);
            f = h + ngood;
            selfcheck(W, row_terms, s_in, s, chunk + h, chunk + f, 1, lane, s_tile);
          }
          while (f <= e) {
            const TileRec T = load_rec_uniform(&s_rec[f]);
            const uint32_t s_in = s;
            if (!(f > h && ((T.key >= 0 && apply(s, T.key, T.s)) || apply_point(s, T)))) {
              n_recfail++;
              const int slot = (T.cons >> 8) - 1;
              s = (uint32_t)rfl((int)resolve_tile(s, row_terms, chunk + f, T.key,
                                                  slot >= 0 ? W.aux + (size_t)slot * kLanes : nullptr, lane, W.dbg, s_tile));
            }
            selfcheck(W, row_terms, s_in, s, chunk + f, chunk + f + 1, 2, lane, s_tile);
            f++;
            if (f > e) break;
            const TileRec Sf = load_rec_uniform(&s_suf[f]);  // the rest of the run in one step
            const uint32_t s_in2 = s;
            if ((Sf.key >= 0 && apply(s, Sf.key, Sf.s)) || apply_point(s, Sf)) {
              selfcheck(W, row_terms, s_in2, s, chunk + f, chunk + e + 1, 3, lane, s_tile);
              break;
            }
          }
        }
      }
      if (lane == 0) {
        atomicAdd(&W.dbg[0], n_run);
        atomicAdd(&W.dbg[1], n_runfail);
        atomicAdd(&W.dbg[4], n_recfail);
        atomicAdd(&W.dbg[8], (unsigned long long)(t_b - t_a));
        atomicAdd(&W.dbg[9], (unsigned long long)(wall_clock64() - t_b));
        atomicMax(&W.dbg[14 + 32], (unsigned long long)(wall_clock64() - t_b));   // slowest row of the launch
      }
    }
    __syncthreads();
  }
  for (int64_t b = threadIdx.x; b < W.nbins; b += kChainBlock) W.bin_sum[row * W.nbins + b] = 0.0;
  // component order of sums10: Value, G0..G5, DistRMS, Weight, Pairs
  if (threadIdx.x == 0) {
    const int slot = row == 0 ? S_VALUE : (row <= 6 ? S_G0 + row - 1 : (row == 7 ? S_DIST_RMS : S_WEIGHT));
    __hip_atomic_store(&sums10[slot], (double)u2f(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // the row that finishes last has all nine sums: evaluate tail + pose update (evaluator.go:156-186,
    // updater.go:44-71) in the same launch
    __threadfence();
    const unsigned ticket = atomicAdd(W.done_rows, 1u);
    if (ticket == (unsigned)kStrictRows - 1u) {
      __threadfence();
      double sums[S_COUNT];
      for (int k = 0; k < S_COUNT; k++) sums[k] = __hip_atomic_load(&sums10[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      sums[S_PAIRS] = (double)__hip_atomic_load(W.pairs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      sums10[S_PAIRS] = sums[S_PAIRS];
      *W.pairs = 0ull;
      *W.aux_count = 0u;
      *W.done_rows = 0u;
      if (fuse_update) icp_update_step(state, sums, kp);
    }
  }
}

// ---- host side ---------------------------------------------------------------------------------------
struct StrictBuffers {
  StrictWork w;
  void *block = nullptr;
};

pcgx_status strict_create(int64_t nt, const float *tx, const float *ty, const float *tz, const uint32_t *pos_of,
                          StrictBuffers **out, hipStream_t st) {
  StrictBuffers *b = new StrictBuffers();
  StrictWork &W = b->w;
  W.nt = nt;
  W.ntiles = nt > 0 ? (nt + kTile - 1) / kTile : 1;
  W.nt_pad = W.ntiles * kTile;
  W.nbins = (W.ntiles + kBinTiles - 1) / kBinTiles;
  W.selfcheck = getenv("PCGX_STRICT_SELFCHECK") ? 1 : 0;
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t sz_terms = up((size_t)kStrictRows * W.nt_pad * sizeof(float));
  const size_t sz_tile = up((size_t)kStrictRows * W.ntiles * sizeof(double));
  const size_t sz_bin = up((size_t)kStrictRows * W.nbins * sizeof(double));
  const size_t sz_rec = up((size_t)kStrictRows * W.ntiles * sizeof(TileRec));
  W.naux = (int32_t)(kStrictRows * W.ntiles / 4 + 64);
  const size_t sz_aux = up((size_t)W.naux * kLanes * sizeof(LeafAux));
  const size_t sz_xyz = up((size_t)(nt ? nt : 1) * 12);
  const size_t total = sz_terms + sz_tile + sz_bin + sz_rec + sz_aux + sz_xyz + 256 + 512;
  hipError_t e = dev_cache_alloc(&b->block, total);
  if (e != hipSuccess) {
    delete b;
    return fail(PCGX_E_OOM, "strict sums: allocation of %zu bytes failed: %s", total, hipGetErrorString(e));
  }
  uint8_t *p = (uint8_t *)b->block;
  W.terms = (float *)p; p += sz_terms;
  W.tile_sum = (double *)p; p += sz_tile;
  W.bin_sum = (double *)p; p += sz_bin;
  W.recs = (TileRec *)p; p += sz_rec;
  W.aux = (LeafAux *)p; p += sz_aux;
  W.xyz_caller = (const float *)p; p += sz_xyz;
  W.pairs = (unsigned long long *)p;
  W.aux_count = (unsigned int *)(p + 8);
  W.done_rows = (unsigned int *)(p + 12); p += 256;
  W.dbg = (unsigned long long *)p;
  // bins, pair counter and debug counters start at zero (the chain kernel re-zeroes what it consumed)
  e = hipMemsetAsync(W.bin_sum, 0, sz_bin, st);
  if (e == hipSuccess) e = hipMemsetAsync(W.pairs, 0, 768, st);
  if (e == hipSuccess && nt > 0) {
    hipLaunchKernelGGL(strict_xyz_caller_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, st, tx, ty, tz, pos_of, nt,
                       const_cast<float *>(W.xyz_caller));
    e = hipGetLastError();
  }
  if (e != hipSuccess) {
    dev_cache_free(b->block);
    delete b;
    return fail(PCGX_E_HIP, "strict sums: hipMemsetAsync failed: %s", hipGetErrorString(e));
  }
  *out = b;
  return PCGX_OK;
}

void strict_destroy(StrictBuffers *b) {
  if (!b) return;
  dev_cache_free(b->block);
  delete b;
}

pcgx_status strict_enqueue(StrictBuffers *b, const float4 *match, const uint32_t *pos_of, IcpState *state,
                           double *sums10, const IcpKernelParams &kp, bool fuse_update, hipStream_t st) {
  b->w.weight_fn = kp.weight_fn;
  b->w.weight_a = kp.weight_a;
  const StrictWork &W = b->w;
  const unsigned waves = (unsigned)(kStrictRows * W.ntiles);
  {
    ProfScope prof(PCGX_PROF_STRICT_TERMS, st);
    hipLaunchKernelGGL(strict_terms_kernel, dim3((unsigned)W.ntiles), dim3(kTermsBlock), 0, st, match, pos_of, (const IcpState *)state, W);
  }
  {
    ProfScope prof(PCGX_PROF_STRICT_SUM, st);
    hipLaunchKernelGGL(strict_sum_kernel, dim3((waves + 3) / 4), dim3(256), 0, st, (const IcpState *)state, W);
  }
  {
    ProfScope prof(PCGX_PROF_STRICT_CHAIN, st);
    hipLaunchKernelGGL(strict_chain_kernel, dim3(kStrictRows), dim3(kChainBlock), 0, st, state, W, sums10, kp, fuse_update ? 1 : 0);
  }
  PCGX_HIP_TRY(hipGetLastError());
  return PCGX_OK;
}

pcgx_status strict_read_debug(StrictBuffers *b, unsigned long long out[48], hipStream_t st) {
  PCGX_HIP_TRY(hipMemcpyAsync(out, b->w.dbg, 48 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
  PCGX_HIP_TRY(hipStreamSynchronize(st));
  PCGX_HIP_TRY(hipMemsetAsync(b->w.dbg, 0, 48 * sizeof(unsigned long long), st));
  return PCGX_OK;
}

}  // namespace pcgx

// Host model of the same pipeline (no GPU involved): the CPU tests run it against a plain
// sequential float32 loop.  See include/pcgx.h.
extern "C" pcgx_status pcgx_debug_strict_sum_host(const float *terms, int64_t n, int32_t mode, float *out,
                                                  int64_t stats[8]) {
  if (n < 0 || (n > 0 && !terms) || !out || !stats) return pcgx::fail(PCGX_E_INVALID, "pcgx_debug_strict_sum_host: bad argument");
  *out = pcgx::ss::ss_host_model(terms, n, stats, mode);
  return PCGX_OK;
}
