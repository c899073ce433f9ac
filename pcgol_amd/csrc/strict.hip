// strict.hip -- the evaluator's sums exactly as the reference forms them (sequential float32
// additions over the pairs in target order, pc/registration/icp/evaluator.go:122-145), computed by
// the whole GPU instead of one dependent chain.  Arithmetic and the proof sketch: strict_sum.h.
//
// Per iteration, after the correspondence kernels left match[] (icp.hip):
//   strict_terms_kernel  one thread per target in the CALLER's order: the nine float32 terms
//                        (rows of terms[]), float64 tile sums, level-1 bins (atomics), pair count
//   strict_err_kernel    one wave per (row, tile): rounding error the chain makes in this tile when
//                        started from the float64 prefix  -> tile_err, bins
//   strict_sum_kernel    one wave per (row, tile): guesses from prefix + error prefix, class
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

struct StrictWork {
  float *terms;                 // [9][nt_pad]
  double *tile_sum, *tile_err;  // [9][ntiles]
  double *bin_sum, *bin_err;    // [9][nbins] level-1 sums of kBinTiles tiles (atomics; zeroed by the chain kernel)
  TileRec *recs;                // [9][ntiles]
  unsigned long long *pairs;    // matched targets of this iteration (atomic; zeroed by the chain kernel)
  unsigned long long *dbg;      // [16] counters (measurement aid), may be null
  int64_t nt, nt_pad, ntiles, nbins;
};

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
// One workgroup per tile of 2048 targets in the caller's order (pos_of: where the session keeps
// target i).  Unmatched targets and the padding behind nt carry -0.0f: x + (-0.0f) == x for every x.
__global__ __launch_bounds__(1024) void strict_terms_kernel(const float *__restrict__ tx, const float *__restrict__ ty,
                                                            const float *__restrict__ tz,
                                                            const float4 *__restrict__ match,
                                                            const uint32_t *__restrict__ pos_of,
                                                            const IcpState *__restrict__ state, StrictWork W) {
  __shared__ double s_part[16][kStrictRows];
  __shared__ int s_pairs[16];
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
#pragma unroll
  for (int h = 0; h < kTile / 1024; h++) {
    const int64_t i = tile * kTile + h * 1024 + threadIdx.x;
    float t[kStrictRows];
#pragma unroll
    for (int k = 0; k < kStrictRows; k++) t[k] = -0.0f;
    if (i < W.nt) {
      const uint32_t pos = pos_of[i];
      const float4 bp = match[pos];
      if (bp.w >= 0.0f) {  // correspondence.go:27-29
        npairs++;
        float x0 = tx[pos], y0 = ty[pos], z0 = tz[pos];
        if (project) {
          float px, py, pz;
          mat4_transform(m, x0, y0, z0, px, py, pz);
          x0 = px; y0 = py; z0 = pz;
        }
        const float x1 = bp.x, y1 = bp.y, z1 = bp.z, w = 1.0f;  // evaluator.go:21-23,130
        t[0] = w * bp.w;
        t[1] = w * (x0 - x1);
        t[2] = w * (y0 - y1);
        t[3] = w * (z0 - z1);
        t[4] = w * (z0 * y1 - y0 * z1);
        t[5] = w * (x0 * z1 - z0 * x1);
        t[6] = w * (y0 * x1 - x0 * y1);
        t[7] = w * norm_sq3(x0, y0, z0);
        t[8] = w;
      }
    }
#pragma unroll
    for (int k = 0; k < kStrictRows; k++) {
      W.terms[(int64_t)k * W.nt_pad + i] = t[k];
      acc[k] += (double)t[k];
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
    for (int w = 0; w < 16; w++) v += s_part[w][threadIdx.x];
    W.tile_sum[threadIdx.x * W.ntiles + tile] = v;
    atomicAdd(&W.bin_sum[threadIdx.x * W.nbins + tile / kBinTiles], v);
  } else if (threadIdx.x == 64) {
    int p = 0;
    for (int w = 0; w < 16; w++) p += s_pairs[w];
    if (p) atomicAdd(W.pairs, (unsigned long long)p);
  }
}

// ---- error pass ------------------------------------------------------------------------------------
__device__ __forceinline__ void load_leaf(const float *__restrict__ p, float *t) {
  const float4 *q = reinterpret_cast<const float4 *>(p);
#pragma unroll
  for (int v = 0; v < kLeaf / 4; v++) {
    const float4 a = q[v];
    t[4 * v] = a.x; t[4 * v + 1] = a.y; t[4 * v + 2] = a.z; t[4 * v + 3] = a.w;
  }
}
__device__ __forceinline__ double leaf_sum_f64(const float *t) {
  double v = 0.0;
#pragma unroll
  for (int j = 0; j < kLeaf; j++) v += (double)t[j];
  return v;
}

__global__ __launch_bounds__(256) void strict_err_kernel(const IcpState *__restrict__ state, StrictWork W) {
  if (state->done) return;
  const int lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= kStrictRows * W.ntiles) return;  // whole wave
  const int row = (int)(w / W.ntiles);
  const int64_t tile = w % W.ntiles;
  float t[kLeaf];
  load_leaf(W.terms + (int64_t)row * W.nt_pad + tile * kTile + lane * kLeaf, t);
  const double P0 = tile_prefix(W.tile_sum, W.bin_sum, W.ntiles, W.nbins, row, tile, lane);
  const double lsum = leaf_sum_f64(t);
  const double pre = wave_excl_scan_f64(lsum, lane);
  const float g = (float)(P0 + pre);
  float s = g;
#pragma unroll
  for (int j = 0; j < kLeaf; j++) s = s + t[j];
  // rounding error of this leaf's 32 additions (exact: the differences are exact in float64)
  const double err = wave_allsum_f64(((double)s - (double)g) - lsum);
  if (lane == 0) {
    W.tile_err[row * W.ntiles + tile] = err;
    atomicAdd(&W.bin_err[row * W.nbins + tile / kBinTiles], err);
  }
}

// ---- summaries ---------------------------------------------------------------------------------------
// leaf guesses of a tile whose first state is (about) base: float64 prefix of the leaf sums, then one
// refinement with the prefix of the rounding errors the chains make from those guesses
__device__ __forceinline__ void tile_guesses(const float *t, double base, bool exact_first, uint32_t first, int lane,
                                             uint32_t &g, ChainRange &cr) {
  const double lsum = leaf_sum_f64(t);
  const double pre = wave_excl_scan_f64(lsum, lane);
  g = (exact_first && lane == 0) ? first : f2u((float)(base + pre));
  cr = guess_chain(t, g);
  const double err = ((double)u2f(cr.end) - (double)u2f(g)) - lsum;
  const double epre = wave_excl_scan_f64(err, lane);
  const uint32_t g2 = (exact_first && lane == 0) ? first : f2u((float)(base + pre + epre));
  if (__ballot(g2 != g) != 0ull) {  // uniform
    g = g2;
    cr = guess_chain(t, g);
  }
}

__global__ __launch_bounds__(256) void strict_sum_kernel(const IcpState *__restrict__ state, StrictWork W) {
  if (state->done) return;
  const int lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= kStrictRows * W.ntiles) return;  // whole wave
  const int row = (int)(w / W.ntiles);
  const int64_t tile = w % W.ntiles;
  float t[kLeaf];
  load_leaf(W.terms + (int64_t)row * W.nt_pad + tile * kTile + lane * kLeaf, t);
  const double P0 = tile_prefix(W.tile_sum, W.bin_sum, W.ntiles, W.nbins, row, tile, lane);
  const double E0 = tile_prefix(W.tile_err, W.bin_err, W.ntiles, W.nbins, row, tile, lane);
  uint32_t g;
  ChainRange cr;
  tile_guesses(t, P0 + E0, false, 0u, lane, g, cr);
  // window of the tile
  const uint32_t mn = wave_all_umin(cr.mn), mx = wave_all_umax(cr.mx);
  const bool one_sign = __ballot(cr.sg_or != cr.sg_and) == 0ull &&
                        (__ballot(cr.sg_or != 0u) == 0ull || __ballot(cr.sg_or == 0u) == 0ull);
  const uint32_t g_first = (uint32_t)rfl((int)g);
  const int32_t key = one_sign ? choose_window(mn, mx, g_first >> 31, g_first & 0x7fffffffu) : -1;
  // point record: the guess chains join up exactly
  const uint32_t g_next = (uint32_t)__shfl_down((int)g, 1);
  const bool cons = __ballot(lane < 63 && g_next != cr.end) == 0ull;
  const uint32_t out = (uint32_t)__shfl((int)cr.end, 63);
  TileRec T;
  T.key = key;
  T.in = g_first;
  T.out = out;
  T.cons = cons ? 1 : 0;
  T.s = summary_identity();
  if (key >= 0) {  // uniform
    const bool one_binade = (cr.mn >> 23) == (cr.mx >> 23);
    Summary S;
    if (__ballot(!one_binade) != 0ull) {  // uniform: some leaf crosses the level
      if (one_binade) S = leaf_summary_binade(t, g, key);
      else S = leaf_summary_general(t, g, key);
    } else {
      S = leaf_summary_binade(t, g, key);
    }
    // ordered reduction over the 64 leaves
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const Summary Y = shfl_summary(S, lane + o);
      if ((lane & (2 * o - 1)) == 0) S = compose(S, Y);
    }
    T.s = S;
  }
  if (lane == 0) W.recs[row * W.ntiles + tile] = T;
}

// ---- chain -----------------------------------------------------------------------------------------
// Exact recomputation of one tile from the known state: leaf guesses from the state itself, every
// leaf summarised in a window of its own, then leaf after leaf: its summary if it covers the
// state, the 32 additions themselves if not.
__device__ __noinline__ uint32_t resolve_generic(uint32_t s, const float *__restrict__ tt, int lane,
                                                 unsigned long long *dbg) {
  float t[kLeaf];
  load_leaf(tt + lane * kLeaf, t);
  uint32_t g;
  ChainRange cr;
  tile_guesses(t, (double)u2f(s), true, s, lane, g, cr);
  const int32_t key = cr.sg_or == cr.sg_and ? choose_window(cr.mn, cr.mx, cr.sg_or, g & 0x7fffffffu) : -1;
  Summary S = summary_identity();
  if (key >= 0) {
    if ((cr.mn >> 23) == (cr.mx >> 23)) S = leaf_summary_binade(t, g, key);
    else S = leaf_summary_general(t, g, key);
  }
  int serial = 0;
  for (int l = 0; l < kLanes; l++) {
    const int32_t k = __builtin_amdgcn_readlane(key, l);
    bool done = false;
    const int32_t n = state_to_n(s, k);
    if (n >= 0) {
      const int r = n & 3;
      int32_t c, lo, hi;
      if (r == 0) { c = __builtin_amdgcn_readlane(S.c[0], l); lo = __builtin_amdgcn_readlane(S.lo[0], l); hi = __builtin_amdgcn_readlane(S.hi[0], l); }
      else if (r == 1) { c = __builtin_amdgcn_readlane(S.c[1], l); lo = __builtin_amdgcn_readlane(S.lo[1], l); hi = __builtin_amdgcn_readlane(S.hi[1], l); }
      else if (r == 2) { c = __builtin_amdgcn_readlane(S.c[2], l); lo = __builtin_amdgcn_readlane(S.lo[2], l); hi = __builtin_amdgcn_readlane(S.hi[2], l); }
      else { c = __builtin_amdgcn_readlane(S.c[3], l); lo = __builtin_amdgcn_readlane(S.lo[3], l); hi = __builtin_amdgcn_readlane(S.hi[3], l); }
      if (n >= lo && n <= hi) {
        s = n_to_state(n + c, k);
        done = true;
      }
    }
    if (!done) {
      float x = u2f(s);
#pragma unroll
      for (int j = 0; j < kLeaf; j++) x = x + u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(t[j]), l));
      s = f2u(x);
      serial++;
    }
  }
  if (dbg && lane == 0) {
    atomicAdd(&dbg[2], 1ull);
    atomicAdd(&dbg[3], (unsigned long long)serial);
  }
  return s;
}

struct RunRec {
  TileRec r;
  int32_t begin, end;  // tiles [begin, end)
};

constexpr int kChainBlock = 512;

__global__ __launch_bounds__(kChainBlock) void strict_chain_kernel(const IcpState *__restrict__ state, StrictWork W,
                                                                  double *__restrict__ sums10) {
  __shared__ TileRec s_run[kChainBlock];
  __shared__ int32_t s_begin[kChainBlock], s_end[kChainBlock];
  __shared__ int32_t s_count[kChainBlock / 64];
  if (state->done) return;
  const int row = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t s = f2u(0.0f);  // walker state (wave 0), evaluator.go:122: the sums start at zero
  for (int64_t chunk = 0; chunk < W.ntiles; chunk += kChainBlock) {
    // ---- runs of equal windows: segmented inclusive scan inside each wave
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
    const int32_t key_prev = __shfl_up(R.key, 1);
    const bool head = lane == 0 || R.key < 0 || R.key != key_prev;
    int32_t begin = (int32_t)tile;
    int flag = head ? 1 : 0;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const Summary X = shfl_summary(R.s, lane - o);
      const uint32_t xin = (uint32_t)__shfl_up((int)R.in, o), xout = (uint32_t)__shfl_up((int)R.out, o);
      const int xcons = __shfl_up(R.cons, o), xbegin = __shfl_up(begin, o), xflag = __shfl_up(flag, o);
      if (lane >= o && !flag) {
        R.s = compose(X, R.s);
        R.cons = xcons && R.cons && xout == R.in;
        R.in = xin;
        begin = xbegin;
        flag = xflag;
      }
    }
    const int head_next = __shfl_down(head ? 1 : 0, 1);
    const bool tail = valid && (lane == 63 || tile + 1 >= W.ntiles || head_next);
    const unsigned long long tails = __ballot(tail);
    if (tail) {
      const int idx = wave * 64 + __popcll(tails & ((1ull << lane) - 1ull));
      s_run[idx] = R;
      s_begin[idx] = begin;
      s_end[idx] = (int32_t)tile + 1;
    }
    if (lane == 0) s_count[wave] = __popcll(tails);
    __syncthreads();
    // ---- the walk: one wave, every lane with the same state
    if (wave == 0) {
      unsigned long long n_run = 0, n_runfail = 0, n_recfail = 0;
      for (int w = 0; w < kChainBlock / 64; w++) {
        const int cnt = s_count[w];
        for (int i = 0; i < cnt; i++) {
          const TileRec &Q = s_run[w * 64 + i];
          n_run++;
          if (Q.key >= 0 && apply(s, Q.key, Q.s)) continue;
          if (Q.cons && Q.in == s) {
            s = Q.out;
            continue;
          }
          n_runfail++;
          const int32_t b = s_begin[w * 64 + i], e = s_end[w * 64 + i];
          for (int32_t q = b; q < e; q++) {
            if (e - b > 1) {  // a run of several tiles: their own records first
              const TileRec T = W.recs[row * W.ntiles + q];
              if (T.key >= 0 && apply(s, T.key, T.s)) continue;
              if (T.cons && T.in == s) {
                s = T.out;
                continue;
              }
            }
            n_recfail++;
            s = (uint32_t)rfl((int)resolve_generic(s, W.terms + (int64_t)row * W.nt_pad + (int64_t)q * kTile, lane, W.dbg));
          }
        }
      }
      if (W.dbg && lane == 0) {
        atomicAdd(&W.dbg[0], n_run);
        atomicAdd(&W.dbg[1], n_runfail);
        atomicAdd(&W.dbg[4], n_recfail);
      }
    }
    __syncthreads();
  }
  // component order of sums10: Value, G0..G5, DistRMS, Weight, Pairs
  if (threadIdx.x == 0) {
    const int slot = row == 0 ? S_VALUE : (row <= 6 ? S_G0 + row - 1 : (row == 7 ? S_DIST_RMS : S_WEIGHT));
    sums10[slot] = (double)u2f(s);
    if (row == 0) {
      sums10[S_PAIRS] = (double)*W.pairs;
      *W.pairs = 0ull;
    }
  }
  for (int64_t b = threadIdx.x; b < W.nbins; b += kChainBlock) {
    W.bin_sum[row * W.nbins + b] = 0.0;
    W.bin_err[row * W.nbins + b] = 0.0;
  }
}

// ---- host side ---------------------------------------------------------------------------------------
struct StrictBuffers {
  StrictWork w;
  void *block = nullptr;
};

pcgx_status strict_create(int64_t nt, StrictBuffers **out, hipStream_t st) {
  StrictBuffers *b = new StrictBuffers();
  StrictWork &W = b->w;
  W.nt = nt;
  W.ntiles = nt > 0 ? (nt + kTile - 1) / kTile : 1;
  W.nt_pad = W.ntiles * kTile;
  W.nbins = (W.ntiles + kBinTiles - 1) / kBinTiles;
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t sz_terms = up((size_t)kStrictRows * W.nt_pad * sizeof(float));
  const size_t sz_tile = up((size_t)kStrictRows * W.ntiles * sizeof(double));
  const size_t sz_bin = up((size_t)kStrictRows * W.nbins * sizeof(double));
  const size_t sz_rec = up((size_t)kStrictRows * W.ntiles * sizeof(TileRec));
  const size_t total = sz_terms + 2 * sz_tile + 2 * sz_bin + sz_rec + 256 + 256;
  hipError_t e = dev_cache_alloc(&b->block, total);
  if (e != hipSuccess) {
    delete b;
    return fail(PCGX_E_OOM, "strict sums: allocation of %zu bytes failed: %s", total, hipGetErrorString(e));
  }
  uint8_t *p = (uint8_t *)b->block;
  W.terms = (float *)p; p += sz_terms;
  W.tile_sum = (double *)p; p += sz_tile;
  W.tile_err = (double *)p; p += sz_tile;
  W.bin_sum = (double *)p; p += sz_bin;
  W.bin_err = (double *)p; p += sz_bin;
  W.recs = (TileRec *)p; p += sz_rec;
  W.pairs = (unsigned long long *)p; p += 256;
  W.dbg = (unsigned long long *)p;
  // bins, pair counter and debug counters start at zero (the chain kernel re-zeroes what it consumed)
  e = hipMemsetAsync(W.bin_sum, 0, 2 * sz_bin, st);
  if (e == hipSuccess) e = hipMemsetAsync(W.pairs, 0, 512, st);
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

pcgx_status strict_enqueue(StrictBuffers *b, const float *tx, const float *ty, const float *tz, const float4 *match,
                           const uint32_t *pos_of, const IcpState *state, double *sums10, hipStream_t st) {
  const StrictWork &W = b->w;
  const unsigned waves = (unsigned)(kStrictRows * W.ntiles);
  hipLaunchKernelGGL(strict_terms_kernel, dim3((unsigned)W.ntiles), dim3(1024), 0, st, tx, ty, tz, match, pos_of, state, W);
  hipLaunchKernelGGL(strict_err_kernel, dim3((waves + 3) / 4), dim3(256), 0, st, state, W);
  hipLaunchKernelGGL(strict_sum_kernel, dim3((waves + 3) / 4), dim3(256), 0, st, state, W);
  hipLaunchKernelGGL(strict_chain_kernel, dim3(kStrictRows), dim3(kChainBlock), 0, st, state, W, sums10);
  PCGX_HIP_TRY(hipGetLastError());
  return PCGX_OK;
}

pcgx_status strict_read_debug(StrictBuffers *b, unsigned long long out[16], hipStream_t st) {
  PCGX_HIP_TRY(hipMemcpyAsync(out, b->w.dbg, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
  PCGX_HIP_TRY(hipStreamSynchronize(st));
  PCGX_HIP_TRY(hipMemsetAsync(b->w.dbg, 0, 16 * sizeof(unsigned long long), st));
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
