// strict.hip -- the evaluator's sums exactly as the reference forms them (sequential float32
// additions over the pairs in target order, pc/registration/icp/evaluator.go:122-145), computed by
// the whole GPU instead of one dependent chain.  Arithmetic and the proof sketch: strict_sum.h.
//
// Per iteration, after the correspondence kernels left every pair in the caller's target order
// (match_caller, icp.hip) -- the nine float32 terms of a pair are formed where they are needed
// (strict_terms.h), never stored as rows in HBM:
//   (tile sums)          float64 sums of every tile's terms, per sum: formed by the correspondence kernel's
//                        workgroups on their way out (strict_terms.h), or by strict_tilesum_kernel
//   strict_sum_kernel    one workgroup per tile of 2048 targets: the tile's terms into LDS (72 KB), then
//                        one wave per sum: leaf guesses from the float64 prefix (refined once inside the
//                        tile by the rounding errors the chains make from them), parity summaries of
//                        the 64 leaves, composed -> one 64-byte record per (sum, tile); tiles that cross
//                        a level or have no window are handed on as jobs
//   strict_job_kernel    one workgroup per job: four class chains per leaf, scans over the leaves ->
//                        the tile's record and its leaves' records (what the chain kernel needs to
//                        recompute the tile from a state the record does not cover)
//   strict_chain_kernel  one workgroup per sum: records of equal windows merged into runs
//                        (segmented wave scans), one wave applies them in order to the exact state;
//                        a record that does not cover the state -> a helper wave, which holds that
//                        tile's leaf records and terms in registers, recomputes the tile exactly
// Nothing here is approximate: a record is applied only when its interval proves the result.
#include <stddef.h>

#include "strict_terms.h"

namespace pcgx {

// ---- wave helpers ---------------------------------------------------------------------------------
// (scans and reductions over a wave by DPP: a partner's value is a register move -- row_shr:n inside a row of 16 lanes,
// row_bcast:15 / :31 from a row's last lane to the rows behind it, wave_shr:1 -- where __shfl is a trip through the LDS
// crossbar and a wait; lanes without a partner get `old`)
// the lane behind / in front (wave_shl:1 / wave_shr:1); the last / first lane keeps `edge`
__device__ __forceinline__ int lane_next(int v, int edge) { return __builtin_amdgcn_update_dpp(edge, v, 0x130, 0xf, 0xf, false); }
__device__ __forceinline__ int lane_prev(int v, int edge) { return __builtin_amdgcn_update_dpp(edge, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ double wave_excl_scan_f64(double v, int lane) {
  double inc = v;
  inc += dpp_f64<0x111, 0xf>(0.0, inc);
  inc += dpp_f64<0x112, 0xf>(0.0, inc);
  inc += dpp_f64<0x114, 0xf>(0.0, inc);
  inc += dpp_f64<0x118, 0xf>(0.0, inc);
  inc += dpp_f64<0x142, 0xa>(0.0, inc);  // rows 1, 3: the row before
  inc += dpp_f64<0x143, 0xc>(0.0, inc);  // rows 2, 3: rows 0 and 1
  (void)lane;
  return dpp_f64<0x138, 0xf>(0.0, inc);  // wave_shr:1 (lane 0: nothing before it)
}
// (a lane without a partner gets the operation's identity for `old`: the compiler then folds the move into the
// v_min_u32 / v_max_u32 itself, one instruction per step instead of three)
__device__ __forceinline__ uint32_t wave_all_umin(uint32_t v) {
#define PCGX_DPP_U32(CTRL, MASK) (uint32_t) __builtin_amdgcn_update_dpp(-1, (int)v, CTRL, MASK, 0xf, false)
  v = umin(v, PCGX_DPP_U32(0x111, 0xf));
  v = umin(v, PCGX_DPP_U32(0x112, 0xf));
  v = umin(v, PCGX_DPP_U32(0x114, 0xf));
  v = umin(v, PCGX_DPP_U32(0x118, 0xf));
  v = umin(v, PCGX_DPP_U32(0x142, 0xa));
  v = umin(v, PCGX_DPP_U32(0x143, 0xc));
#undef PCGX_DPP_U32
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);  // (lane 63 has seen every lane)
}
__device__ __forceinline__ uint32_t wave_all_umax(uint32_t v) {
#define PCGX_DPP_U32(CTRL, MASK) (uint32_t) __builtin_amdgcn_update_dpp(0, (int)v, CTRL, MASK, 0xf, false)
  v = umax(v, PCGX_DPP_U32(0x111, 0xf));
  v = umax(v, PCGX_DPP_U32(0x112, 0xf));
  v = umax(v, PCGX_DPP_U32(0x114, 0xf));
  v = umax(v, PCGX_DPP_U32(0x118, 0xf));
  v = umax(v, PCGX_DPP_U32(0x142, 0xa));
  v = umax(v, PCGX_DPP_U32(0x143, 0xc));
#undef PCGX_DPP_U32
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
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
// A word from the lane a DPP control names, for scans whose steps look at it only in the lanes that HAVE such a partner:
// the others' result is not defined (no `old` operand to set up in front of every move: that was one v_mov per
// word and step, a quarter of a scan's instructions).
template <int kCtrl, int kRowMask>
__device__ __forceinline__ int dpp_partner(int v) {
  return __builtin_amdgcn_mov_dpp(v, kCtrl, kRowMask, 0xf, kRowMask == 0xf);
}
// a summary from the lane a DPP control names (see wave_excl_scan_f64)
template <int kCtrl, int kRowMask>
__device__ __forceinline__ Summary dpp_summary(const Summary &S) {
  Summary R;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    R.c[r] = dpp_partner<kCtrl, kRowMask>(S.c[r]);
    R.lo[r] = dpp_partner<kCtrl, kRowMask>(S.lo[r]);
    R.hi[r] = dpp_partner<kCtrl, kRowMask>(S.hi[r]);
  }
  return R;
}

// Wall-clock reads for the debug counters' tick columns (tools/strict_probe.py) and the PCGX_STRICT_TRACE stamps.
// Off unless asked for (PCGX_STRICT_CLOCKS / PCGX_STRICT_TRACE): an s_memrealtime is a round trip of its own,
// and the chain kernel's walker would make two per chunk and two per tile it has resolved.
__device__ __forceinline__ long long stat_clock(const StrictWork &W) { return (W.selfcheck & 8) ? (long long)wall_clock64() : 0ll; }
// (PCGX_STRICT_TRACE alone: a stamp when a workgroup comes and when it goes, and where it says so; nothing inside the walk)
__device__ __forceinline__ long long trace_clock(const StrictWork &W) { return (W.selfcheck & 10) ? (long long)wall_clock64() : 0ll; }
// a stamp that cannot be taken before `dep` is known (the clock read has no inputs of its own: the compiler moves it
// up past the arithmetic it is meant to time)
__device__ __forceinline__ unsigned long long clock_after(uint32_t dep) {
  unsigned long long t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "v"(dep) : "memory");
  return t;
}

// A tile's terms of one row, 2048 floats as 512 quads: quad v (terms 4v .. 4v + 3) of leaf l sits at
// float4 index v * 64 + (l ^ v).  In LDS the 64 lanes of a wave read their leaves' quad v without bank
// conflicts, and the 8 threads that write the quads of one leaf hit different banks (the xor); in HBM
// (aux_terms) a wave reads and writes 64 consecutive quads per instruction.
__device__ __forceinline__ int tile_quad(int l, int v) { return v * kLanes + (l ^ v); }

struct LdsQuads {  // the leaf of `lane` in a tile staged in LDS, quad by quad (strict_sum.h chains)
  const float4 *R;
  int lane;
  __device__ __forceinline__ float4 operator()(int v) const { return R[tile_quad(lane, v)]; }
};

__device__ __forceinline__ void load_leaf_quads(const float4 *R, int lane, float *t) {
#pragma unroll
  for (int v = 0; v < kLeaf / 4; v++) {
    const float4 a = R[tile_quad(lane, v)];
    t[4 * v] = a.x; t[4 * v + 1] = a.y; t[4 * v + 2] = a.z; t[4 * v + 3] = a.w;
  }
}
__device__ __forceinline__ void store_leaf_quads(float4 *R, int lane, const float *t) {
#pragma unroll
  for (int v = 0; v < kLeaf / 4; v++) R[tile_quad(lane, v)] = make_float4(t[4 * v], t[4 * v + 1], t[4 * v + 2], t[4 * v + 3]);
}
__device__ __forceinline__ void lds_fence_wave() {
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's own LDS accesses have landed
}

// the additions of leaves [l0, l1) of a tile staged in LDS, one after the other: every lane runs the
// same chain (broadcast reads; the next leaf's eight quads are read while the 32 additions run)
__device__ __forceinline__ uint32_t serial_leaves(uint32_t s, const float4 *R, int l0, int l1) {
  float x = u2f(s);
  if (l0 >= l1) return s;
  float4 a[8], b[8];
#pragma unroll
  for (int v = 0; v < 8; v++) a[v] = R[tile_quad(l0, v)];
  for (int l = l0; l < l1; l += 2) {
    const int ln = l + 1 < l1 ? l + 1 : l0;
#pragma unroll
    for (int v = 0; v < 8; v++) b[v] = R[tile_quad(ln, v)];
#pragma unroll
    for (int v = 0; v < 8; v++) x = (((x + a[v].x) + a[v].y) + a[v].z) + a[v].w;
    if (l + 1 >= l1) break;
    const int ln2 = l + 2 < l1 ? l + 2 : l0;
#pragma unroll
    for (int v = 0; v < 8; v++) a[v] = R[tile_quad(ln2, v)];
#pragma unroll
    for (int v = 0; v < 8; v++) x = (((x + b[v].x) + b[v].y) + b[v].z) + b[v].w;
  }
  return f2u(x);
}

// The chain of leaves l0 .. l1 - 1 from kN start states per lane (every lane the same terms), out of a LINEAR copy of
// the tile's terms (L[8 l + v] = quad v of leaf l): eight reads at one base register plus immediate offsets per
// leaf, a leaf ahead of the additions, and 32 x kN plain v_add_f32.  Out of the swizzled layout (serial_leaves
// above) the compiler sets up 28 addresses and reads per leaf in a block in front of the 32 dependent additions:
// 0.18 us per leaf where the additions alone are 0.08 (a dependent v_add_f32 issues every 6 cycles,
// tools/micro/dep_add.cpp); terms as v_readlane scalars, or taken from a lane of the own quad by DPP: the same 0.18.
// kN > 1: independent additions per term fill that pipeline -- four start states cost 1.9x one.
template <int kN>
__device__ __forceinline__ void serial_leaves_lin(float (&x)[kN], const float4 *L, int l0, int l1) {
  if (l0 >= l1) return;
  float4 a[kLeaf / 4];
#pragma unroll
  for (int v = 0; v < kLeaf / 4; v++) a[v] = L[8 * l0 + v];
  for (int l = l0; l < l1; l++) {  // uniform
    const int ln = l + 1 < l1 ? l + 1 : l;
    float4 b[kLeaf / 4];
#pragma unroll
    for (int v = 0; v < kLeaf / 4; v++) b[v] = L[8 * ln + v];
#pragma unroll
    for (int v = 0; v < kLeaf / 4; v++) {
#pragma unroll
      for (int c = 0; c < kN; c++) x[c] = x[c] + a[v].x;
#pragma unroll
      for (int c = 0; c < kN; c++) x[c] = x[c] + a[v].y;
#pragma unroll
      for (int c = 0; c < kN; c++) x[c] = x[c] + a[v].z;
#pragma unroll
      for (int c = 0; c < kN; c++) x[c] = x[c] + a[v].w;
    }
#pragma unroll
    for (int v = 0; v < kLeaf / 4; v++) a[v] = b[v];
  }
}
constexpr int kCandPerLane = kCandInner / kLanes;

// Row `row` of `tile` formed again from the pairs and staged in lds (layout of tile_quad): the chain
// kernel's way to a tile that owns no slot (rare).  One wave: eight rounds of 64 consecutive quads
// (coalesced loads), two rounds in flight.
__device__ __forceinline__ void recompute_tile_to_lds(const TermSrc &S, int row, int64_t tile, int lane, float4 *lds) {
#pragma unroll 1
  for (int r = 0; r < kTile / 4 / kLanes; r += 2) {
    float4 bp[2][4];
    float tx[2][4], ty[2][4], tz[2][4];
#pragma unroll
    for (int h = 0; h < 2; h++) load_quad(S, tile * kTile + 4 * (int64_t)((r + h) * kLanes + lane), bp[h], tx[h], ty[h], tz[h]);
#pragma unroll
    for (int h = 0; h < 2; h++) {
      float o[4];
#pragma unroll
      for (int c = 0; c < 4; c++) {
        float q[kStrictRows];
        pair_terms(S, tx[h][c], ty[h][c], tz[h][c], bp[h][c], q);
        o[c] = q[0];
#pragma unroll
        for (int k = 1; k < kStrictRows; k++) o[c] = row == k ? q[k] : o[c];
      }
      const int quad = (r + h) * kLanes + lane;  // quad `quad & 7` of leaf `quad >> 3`
      lds[tile_quad(quad >> 3, quad & 7)] = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
  lds_fence_wave();
}

__device__ __forceinline__ double leaf_sum_f64(const float *t) {
  double v = 0.0;
#pragma unroll
  for (int j = 0; j < kLeaf; j++) v += (double)t[j];
  return v;
}

// float64 prefix of `tile`: the sums of all tiles before it, every lane gets it.  Eight independent
// loads per lane and round (489 tiles at C4: one round).  No level-1 bins: round 2 filled them with
// 4401 atomic adds on five cache lines, 12 of the 24 us of its terms kernel.
// (Measured and rejected in round 3: no kernel for the tile sums at all -- every tile's workgroup
// publishes its sums with agent-scope stores / exchanges, the last tile of every 32 their total, and
// polls the ones before it.  Correct, and never faster than 80 us for the summary kernel: agent-scope
// loads are served by the polling XCD's own L2, which keeps the "not yet" it saw first until the line
// happens to be evicted, tens of microseconds later, whatever the poll interval; acquire fences before
// every poll: slower still.)
__device__ __forceinline__ double tile_prefix(const double *__restrict__ tile_v, int64_t ntiles, int row, int64_t tile,
                                              int lane) {
  const double *p = tile_v + (int64_t)row * ntiles;
  double v = 0.0;
  for (int64_t base = 0; base < tile; base += 8 * 64) {
    double x[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int64_t k = base + u * 64 + lane;
      x[u] = k < tile ? p[k] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; u++) v += x[u];
  }
  return wave_allsum_f64(v);
}

// The same prefix out of what the tiles' workgroups of THIS launch have published (StrictWork::tile_pub,
// [row][ntiles_pad] doubles): 16-byte loads that bypass the vector L1 (sc1), lane l of a round the tiles 2 l and
// 2 l + 1 -- a wave instruction reads 1 KB in one piece -- four rounds in flight, behind the arrival poll of
// strict_sum_kernel and a workgroup barrier.  The producers' stores are write-through (sc1) and drained (s_waitcnt
// vmcnt(0)) before their arrival bit is set (MI355X guide, inter-workgroup visibility: sc1 stores + drained flag +
// sc1 loads).  Only guesses depend on the values: a sum read too early would cost time (records that do not cover
// the state), never a wrong result.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double tile_prefix_pub(const double *__restrict__ pub, int64_t ntiles_pad, int row, int64_t tile,
                                                  int lane) {
  const char *p = reinterpret_cast<const char *>(pub + (int64_t)row * ntiles_pad);
  double v = 0.0;
  for (int64_t base = 0; base < tile; base += 4 * 128) {  // (ntiles_pad is a multiple of 128: every load lies inside the row)
    u32x4 x[4];
    const char *a[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int64_t k = base + u * 128 + 2 * lane;
      a[u] = p + (size_t)(k < tile ? k : 0) * 8;
    }
    asm volatile(
        "global_load_dwordx4 %0, %4, off sc1\n\t"
        "global_load_dwordx4 %1, %5, off sc1\n\t"
        "global_load_dwordx4 %2, %6, off sc1\n\t"
        "global_load_dwordx4 %3, %7, off sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3])
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3])
        : "memory");
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int64_t k = base + u * 128 + 2 * lane;
      const double d0 = __longlong_as_double((long long)((unsigned long long)x[u].y << 32 | x[u].x));
      const double d1 = __longlong_as_double((long long)((unsigned long long)x[u].w << 32 | x[u].z));
      v += k < tile ? d0 : 0.0;
      v += k + 1 < tile ? d1 : 0.0;
    }
  }
  return wave_allsum_f64(v);
}

// the prefixes of two arrays at once (strict_job_kernel: tile sums + tile errors), their loads in flight together
__device__ __forceinline__ double tile_prefix2(const double *__restrict__ a, const double *__restrict__ b, int64_t ntiles, int row,
                                               int64_t tile, int lane) {
  const double *pa = a + (int64_t)row * ntiles, *pb = b + (int64_t)row * ntiles;
  double v = 0.0, w = 0.0;
  for (int64_t base = 0; base < tile; base += 8 * 64) {
    double x[8], y[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int64_t k = base + u * 64 + lane;
      x[u] = k < tile ? pa[k] : 0.0;
      y[u] = k < tile ? pb[k] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      v += x[u];
      w += y[u];
    }
  }
  return wave_allsum_f64(v) + wave_allsum_f64(w);  // (each the sum tile_prefix() returns, bit for bit)
}

// ---- tile sums (strict_terms.h) as a kernel of their own: sessions whose correspondence kernels do not form them
constexpr int kTileSumBlock = 256;
__global__ __launch_bounds__(kTileSumBlock) void strict_tilesum_kernel(const float4 *__restrict__ match,
                                                                       const uint32_t *__restrict__ pos_of,
                                                                       const IcpState *__restrict__ state, StrictWork W) {
  __shared__ double s_part[kTileSumBlock / 64][kStrictRows];
  if (state->done) return;
  const TermSrc S = make_term_src(match, pos_of, state, W);
  tile_sums_block<kTileSumBlock>(S, W, blockIdx.x, s_part);
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

// the 32 additions of leaf l, one after the other, on the state every lane holds; the leaf's terms are in
// lane l's registers
__device__ __forceinline__ uint32_t serial_leaf_regs(uint32_t s, const float *t, int l) {
  float x = u2f(s);
#pragma unroll
  for (int j = 0; j < kLeaf; j++) x = x + u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(t[j]), l));
  return f2u(x);
}

// ---- summaries ---------------------------------------------------------------------------------------
// leaf guesses of a tile whose first state is (about) base: float64 prefix of the leaf sums (pre), then
// one refinement with the prefix of the rounding errors the chains make from those guesses (the first
// pass needs the chains' ends only; the chain that is kept, with its extremes, runs once, from the
// final guesses)
__device__ __forceinline__ void tile_guesses(const LdsQuads &q, double base, double lsum, double pre, int lane, uint32_t &g,
                                             ChainRange &cr, double &tile_err) {
  g = f2u((float)(base + pre));
  const float e = u2f(plain_chain_q(q, g));
  const double err = ((double)e - (double)u2f(g)) - lsum;
  const double epre = wave_excl_scan_f64(err, lane);
  tile_err = lane_f64(epre, 63) + lane_f64(err, 63);  // what the float32 chain is off the exact sum by, over this tile
  const uint32_t g2 = f2u((float)(base + pre + epre));
  // (guesses a couple of ulps off are as good: the intervals are thousands wide except next to a level)
  const int32_t moved = (int32_t)g2 - (int32_t)g;
  if (__ballot(moved > 2 || moved < -2) != 0ull) g = g2;  // uniform
  cr = guess_chain_q(q, g);
}
// the same, with the chain of the other parity (from g ^ 1) run next to the guess chain: the summary kernel's
// plain tiles need both
__device__ __forceinline__ void tile_guesses_pair(const LdsQuads &q, double base, double lsum, double pre, int lane, uint32_t &g,
                                                  ChainRange &cr, ChainRange &crb, double &tile_err) {
  g = f2u((float)(base + pre));
  const float e = u2f(plain_chain_q(q, g));
  const double err = ((double)e - (double)u2f(g)) - lsum;
  const double epre = wave_excl_scan_f64(err, lane);
  tile_err = lane_f64(epre, 63) + lane_f64(err, 63);
  const uint32_t g2 = f2u((float)(base + pre + epre));
  const int32_t moved = (int32_t)g2 - (int32_t)g;
  if (__ballot(moved > 2 || moved < -2) != 0ull) g = g2;  // uniform
  guess_chain_pair_q(q, g, cr, crb);
}

// tiles at the start of every sum that are added up term by term in the summary kernel.  A sum starts
// at 0.0f and runs through a new binade every few terms, so no window holds its first tile.  More than
// one such tile was measured (4: the summary kernel's slowest wave then outlasts the rest of it; the
// tiles where a sum hovers around zero are not the first ones)
constexpr int kExactTiles = 1;
constexpr uint32_t kPlainMargin = 512u;
constexpr uint32_t kMaybeMargin = 1u << 17;  // 1.6 % of a binade at either end: ~3 % of the plain tiles

// One workgroup per tile, one wave per row.  The workgroup forms the tile's terms ONCE, into LDS (72 KB:
// the CDNA4-sized LDS is what lets nine rows of 2048 terms sit next to each other, two workgroups per
// CU), then every wave summarises its row: leaf guesses, guess chains, window, class summaries, ordered
// composition -> one 64-byte record.  A row whose tile crosses a level, or has no window at all (a sum
// hovering around zero), needs four class chains per leaf and two scans over the leaves -- five times
// the work of a plain row, and what the launch used to wait for: it becomes a JOB that the whole
// workgroup shares (four waves one class each, then two waves one scan each).
constexpr int kSumBlock = 512;
constexpr int kSumWaves = kSumBlock / 64;
enum { JOB_CROSSING = 1, JOB_NOWINDOW = 2 };

// kExchange: nobody formed the tile sums before this launch.  Every workgroup forms its own tile's nine sums from the
// terms it has just staged (phase 1), publishes them (one 128-byte line, write-through), sets its arrival bit and
// waits for the bits of all tiles BEFORE its own; then its waves read their prefixes out of the published lines.
// A workgroup waits for lower block indices only, and each XCD's dispatcher hands out its blocks in ascending
// order, so the lowest unfinished tile is always resident or next in line on an XCD with room: the wait ends
// whatever part of the grid is resident (C5's 3907 tiles on 512 places as well).  The poll is bounded all the same
// (kExchangeTicks): a workgroup that gives up goes on with the lines as they are -- its guesses are then off, its
// records do not cover the states, the chain kernel adds its tile term by term: slow and still exact (dbg[63]
// counts; the tests require 0).  What this buys: the 28 MB pass over pairs and targets that formed the tile sums
// in front of this kernel (12-14 us of the C4 step), for a wait of 2-4 us behind the slowest phase 1.
// (The bound is wall-clock time, 2 ms: two such launches running side by side -- two Fits in two of the library's call
// contexts -- could in principle hold each other's next-in-line workgroups out of the XCDs they need, each side's
// residents waiting; giving up frees the places.)
// kSharded (a target spread over ranks, the ring form: strict_enqueue_ring; with kExchange): the guesses of rank r > 0
// start from the float64 totals of the ranks before it.  A rank's totals are known to its LAST tile as soon as its
// exchange is through (the prefix over all earlier tiles + its own sums): that tile's waves put them into the inboxes
// of the ranks behind this one.  Tile 0's first wave fetches the totals of the ranks before this one from the rank's
// inbox (host-coherent memory: one workgroup polls over PCIe, not five hundred), adds them up in rank order and puts
// the result up in device memory like a tile's sums (W.ring_base, write-through, then W.ring_flag); every workgroup
// waits for that flag behind its wait for the earlier tiles.  All of it bounded like the exchange itself, and only
// guesses depend on it.  No rank waits for anything that depends on ANOTHER rank's wait: the totals leave a rank
// without row_base in them, so the ranks' launches do not chain up.
constexpr long long kExchangeTicks = 200000;  // s_memrealtime runs at 100 MHz
template <bool kExchange, bool kSharded = false>
__global__ __launch_bounds__(kSumBlock) __attribute__((amdgpu_waves_per_eu(4, 4))) void strict_sum_kernel(
    const float4 *__restrict__ match, const uint32_t *__restrict__ pos_of, const IcpState *__restrict__ state, StrictWork W) {
  __shared__ float4 s_terms[kStrictRows][kTile / 4];
  __shared__ int s_np[kSumWaves];
  __shared__ double s_tot[16];
  static_assert(!kSharded || kExchange, "the ring form rides on the exchange");
  const int done = state->done;  // (looked at behind phase 1, whose loads it would only hold up: nothing is written before)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t tile = blockIdx.x;
  const int NR = W.nrows;
  const long long t_0 = trace_clock(W);
  long long t_x0 = 0;
  // the float64 prefix of this wave's row (strict_tilesum_kernel's sums): its loads fly while phase 1 runs
  double P0 = (!kExchange && wave < NR) ? tile_prefix(W.tile_sum, W.ntiles, wave, tile, lane) : 0.0;
  // ---- phase 1: the tile's terms, every thread one quad
  {
    const TermSrc S = make_term_src(match, pos_of, state, W);
    const int l = threadIdx.x >> 3, v = threadIdx.x & 7;
    float4 bp[4];
    float tx[4], ty[4], tz[4];
    load_quad(S, tile * kTile + (int64_t)l * kLeaf + 4 * v, bp, tx, ty, tz);
    float t[4][kStrictRows];
    int np = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) np += pair_terms(S, tx[c], ty[c], tz[c], bp[c], t[c]) ? 1 : 0;
#pragma unroll
    for (int k = 0; k < kStrictRows; k++) s_terms[k][tile_quad(l, v)] = make_float4(t[0][k], t[1][k], t[2][k], t[3][k]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) np += __shfl_xor(np, o);
    if (lane == 0) s_np[wave] = np;
  }
  __syncthreads();
  if (done) return;  // uniform
  if (threadIdx.x == 0) {
    int np = 0;
    for (int w = 0; w < kSumWaves; w++) np += s_np[w];
    W.tile_pairs[tile] = (uint32_t)np;
  }
  // (a wave's rows are the same before and behind the exchange: its leaves' float64 sums and their scan are kept)
  static_assert(kStrictRows <= 2 * kSumWaves, "a wave has at most two rows");
  double lsum_kept[2] = {0.0, 0.0}, pre_kept[2] = {0.0, 0.0};
  if (kExchange) {
    // ---- the tile's own sums out, the earlier tiles' sums in
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
      const int row = wave + pass * kSumWaves;
      if (row >= NR) break;  // uniform
      const LdsQuads q{s_terms[row], lane};
      double lsum = 0.0;
#pragma unroll
      for (int v = 0; v < kLeaf / 4; v++) {
        const float4 a = q(v);
        lsum += (double)a.x; lsum += (double)a.y; lsum += (double)a.z; lsum += (double)a.w;
      }
      const double pre = wave_excl_scan_f64(lsum, lane);  // (the order phase 2 adds them up in: the same double)
      lsum_kept[pass] = lsum;
      pre_kept[pass] = pre;
      if (lane == 63) s_tot[row] = pre + lsum;
    }
    __syncthreads();
    t_x0 = trace_clock(W);
    if (wave == 0) {
      if (lane < NR) {
        const double v = s_tot[lane];
        __hip_atomic_store(&W.tile_pub[(int64_t)lane * W.ntiles_pad + tile], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        W.tile_sum[(int64_t)lane * W.ntiles + tile] = v;  // (for strict_job_kernel, a launch later)
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // arrival: my bit in my group's word; whoever completes a group of 32 tiles says so one level up
      const int64_t g_mine = tile >> 5, n_groups = (W.ntiles + 31) >> 5;
      if (lane == 0) {
        const uint32_t bit = 1u << (tile & 31);
        const uint32_t full = (g_mine == n_groups - 1 && (W.ntiles & 31)) ? (1u << (W.ntiles & 31)) - 1u : 0xffffffffu;
        const uint32_t old = __hip_atomic_fetch_or(&W.tile_arrived[32 * g_mine], bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((old | bit) == full)
          __hip_atomic_fetch_or(&W.tile_arrived[32 * (n_groups + (g_mine >> 5))], 1u << (g_mine & 31), __ATOMIC_RELAXED,
                                __HIP_MEMORY_SCOPE_AGENT);
      }
      // wait for every tile before mine: lane 0 for the earlier tiles of my group, lanes 1.. for the groups before it
      bool gave_up = false;
      const int64_t G_mine = g_mine >> 5;
      long long t_first = 0;
      for (int64_t G0 = 0; G0 <= G_mine && !gave_up; G0 += kLanes - 1) {  // uniform (one round up to 64512 tiles)
        const int64_t G = G0 + lane - 1;
        uint32_t want = 0u;
        const unsigned int *word = W.tile_arrived;
        if (lane == 0) {
          want = G0 == 0 ? (1u << (tile & 31)) - 1u : 0u;
          word = W.tile_arrived + 32 * g_mine;
        } else if (G <= G_mine) {
          want = G < G_mine ? 0xffffffffu : (1u << (g_mine & 31)) - 1u;
          word = W.tile_arrived + 32 * (n_groups + G);
        }
        for (int spins = 0;; spins++) {
          uint32_t have = 0u;
          if (want) have = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (__ballot((have & want) != want) == 0ull) break;
          if ((spins & 63) == 63) {  // (a clock read is a memory round trip: now and then)
            const long long now = (long long)wall_clock64();
            if (t_first == 0) t_first = now;
            if (now - t_first > kExchangeTicks) {
              gave_up = true;
              break;
            }
          }
          __builtin_amdgcn_s_sleep(4);
        }
      }
      if (gave_up && lane == 0) atomicAdd(&W.dbg[63], 1ull);
      if (kSharded && W.rank > 0) {  // uniform
        if (tile == 0) {
          // the totals of the ranks before this one: lane (k, row) fetches rank k's, lanes 0 .. 8 add them up in rank order
          const RingLayout RL{W.world};
          double acc = 0.0;
          for (int k0 = 0; k0 < W.rank; k0 += 7) {  // uniform (seven ranks a round)
            const int k = k0 + lane / kStrictRows, row = lane % kStrictRows;
            double v = 0.0;
            if (lane < 7 * kStrictRows && k < W.rank && row < NR) (void)ring_wait_f64(W, RL.row_tot(k, row), v, W.ring_guess_ticks);
#pragma unroll
            for (int j = 0; j < 7; j++) {
              const double vj = __shfl(v, j * kStrictRows + (lane < kStrictRows ? lane : 0));
              acc += vj;  // (rank order; a rank beyond `rank`, or a word that never came: 0.0)
            }
          }
          if (lane < NR) __hip_atomic_store(&W.ring_base[lane], acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (lane == 0) __hip_atomic_store(W.ring_flag, W.ring_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          long long t_first = 0;
          for (int spins = 0;; spins++) {  // uniform
            if (__hip_atomic_load(W.ring_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == W.ring_epoch) break;
            if ((spins & 63) == 63) {
              const long long now = (long long)wall_clock64();
              if (t_first == 0) t_first = now;
              if (now - t_first > 2 * W.ring_guess_ticks) break;  // (guesses without the ranks before this one: slow, still exact)
            }
            __builtin_amdgcn_s_sleep(4);
          }
        }
      }
    }
    __syncthreads();
  }
  const long long t_1 = trace_clock(W);
  // ---- phase 2: every wave its row (one pass unless a weight function adds the ninth row)
  for (int row = wave; row < NR; row += kSumWaves) {
    if (kExchange) P0 = tile_prefix_pub(W.tile_pub, W.ntiles_pad, row, tile, lane);
    else if (row >= kSumWaves) P0 = tile_prefix(W.tile_sum, W.ntiles, row, tile, lane);
    double P0r = P0 + ((!kSharded && W.row_base) ? W.row_base[row] : 0.0);  // (the ranks before this one, strict_enqueue_sharded)
    if (kSharded) {
      // (... in the ring form: what tile 0 has put up, read past the caches like the tiles' sums)
      if (W.rank > 0) P0r = P0 + __hip_atomic_load(&W.ring_base[row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // this rank's own total, to the ranks behind it: the last tile has it here
      if (tile == W.ntiles - 1 && lane < W.world - 1 - W.rank) {
        const RingLayout RL{W.world};
        ring_put_f64(ring_inbox(W, W.rank + 1 + lane) + RL.row_tot(W.rank, row), P0 + s_tot[row], W.ring_epoch);
      }
    }
    const LdsQuads q{s_terms[row], lane};
    TileRec T;
    T.s = summary_identity();
    T.key = -1;
    T.in = T.out = 0u;
    T.cons = 0;
    // The first tile of a row is added up term by term from the one state known in advance (0.0f,
    // evaluator.go:122) -> a point record.  2048 dependent additions take 5-15 us (one every 6 cycles at best,
    // and the SIMD is shared): here they kept one workgroup busy 6 us longer than any other.  The tile goes to
    // strict_job_kernel instead (a job like the tiles below), which has that time to spare; what this kernel
    // keeps is the estimate of the tile's rounding error, from the guess chains like everywhere else.
    const bool first = tile < kExactTiles && W.first_exact;
    double lsum = 0.0, pre;
    if (kExchange) {
      lsum = row < kSumWaves ? lsum_kept[0] : lsum_kept[1];
      pre = row < kSumWaves ? pre_kept[0] : pre_kept[1];
    } else {
#pragma unroll
      for (int v = 0; v < kLeaf / 4; v++) {
        const float4 a = q(v);
        lsum += (double)a.x; lsum += (double)a.y; lsum += (double)a.z; lsum += (double)a.w;
      }
      pre = wave_excl_scan_f64(lsum, lane);
    }
    uint32_t g;
    ChainRange cr, crb;
    double terr;
    tile_guesses_pair(q, P0r, lsum, pre, lane, g, cr, crb, terr);
    // (plus what the stored tile sum is off the sum of the terms by: it was formed while a few of the tile's
    // pairs were still being walked for, icp.hip -- the job kernel's guesses then start from prefixes that add
    // up to the terms as they are)
    const double tile_total = lane_f64(pre, 63) + lane_f64(lsum, 63);
    if (lane == 0)
      W.tile_err[(int64_t)row * W.ntiles + tile] = terr + (tile_total - (kExchange ? s_tot[row] : W.tile_sum[(int64_t)row * W.ntiles + tile]));
    // window of the tile
    const uint32_t mn = wave_all_umin(cr.mn), mx = wave_all_umax(cr.mx);
    const bool one_sign = __ballot(cr.sg_or != cr.sg_and) == 0ull &&
                          (__ballot(cr.sg_or != 0u) == 0ull || __ballot(cr.sg_or == 0u) == 0ull);
    const uint32_t g_first = (uint32_t)rfl((int)g);
    const int32_t key = one_sign ? choose_window(mn, mx, g_first >> 31, g_first & 0x7fffffffu) : -1;
    // point record: the guess chains join up exactly
    const uint32_t g_next = (uint32_t)lane_next((int)g, (int)g);  // (lane 63's is not looked at)
    const bool cons = __ballot(lane < 63 && g_next != cr.end) == 0ull;
    T.key = key;
    T.in = g_first;
    T.out = (uint32_t)__builtin_amdgcn_readlane((int)cr.end, 63);
    T.cons = cons ? 1 : 0;
    // a plain tile: all of it in one binade, and not within kPlainMargin floats of the binade's ends -- the record
    // of a plain tile covers exactly the states that stay inside the binade, the true state is some floats off
    // the guess (the float32 chain's own drift since the row began is not in a guess made here: strict_job_kernel),
    // and a plain tile that does not cover it owns no slot: the chain kernel forms its terms again from the pairs
    // and adds all 2048 (~25 us; seen a few times in twenty iterations of C4).  Next to an end the tile goes
    // the way of a level crossing instead (its window holds the binade beyond that end).  What is left: a row
    // that drifts by 10^4 floats (C4's first iteration, where every term of a gradient sum has the same sign)
    // still loses the one or two tiles in which it passes a binade end.
    const uint32_t m_lo = mn & 0x7fffffu, m_hi = mx & 0x7fffffu;
    int32_t maybe_E = -1;  // >= 0: a plain tile (of that binade) that is a job as well
    if (!first && key >= 0 && (mn >> 23) == (mx >> 23) && m_lo >= kPlainMargin && m_hi <= 0x7fffffu - kPlainMargin) {  // uniform
      const uint32_t E = mn >> 23;
      Par S = leaf_parity_summary_pair(g, cr, crb, E, g_first >> 31);
      // ordered reduction over the 64 leaves (the partner's summary by DPP inside a row of 16 lanes -- row_shl:o, a move
      // per word -- and through the LDS crossbar for the two steps across rows)
      auto fold = [&](const Par &Y, int o) {
        if ((lane & (2 * o - 1)) == 0) S = par_compose(S, Y);
      };
#define PCGX_PAR_DPP(CTRL, O)                                                                         \
  do {                                                                                                \
    Par Y;                                                                                            \
    _Pragma("unroll") for (int p = 0; p < 2; p++) {                                                   \
      Y.c[p] = dpp_partner<CTRL, 0xf>(S.c[p]);                    \
      Y.lo[p] = dpp_partner<CTRL, 0xf>(S.lo[p]);                 \
      Y.hi[p] = dpp_partner<CTRL, 0xf>(S.hi[p]);                 \
    }                                                                                                 \
    fold(Y, O);                                                                                       \
  } while (0)
      PCGX_PAR_DPP(0x101, 1);
      PCGX_PAR_DPP(0x102, 2);
      PCGX_PAR_DPP(0x104, 4);
      PCGX_PAR_DPP(0x108, 8);
#undef PCGX_PAR_DPP
#pragma unroll
      for (int o = 16; o < 64; o <<= 1) {
        Par Y;
#pragma unroll
        for (int p = 0; p < 2; p++) {
          Y.c[p] = __shfl(S.c[p], lane + o);
          Y.lo[p] = __shfl(S.lo[p], lane + o);
          Y.hi[p] = __shfl(S.hi[p], lane + o);
        }
        fold(Y, o);
      }
      T.s = par_expand(S, E, key);
      if (lane == 0) W.recs[row * W.ntiles + tile] = T;
      // Within kMaybeMargin of an end the tile is ALSO handed to strict_job_kernel, which knows how far the chain has
      // drifted from the float64 sums by this tile (the prefix of the tiles' rounding errors: 10^4 floats in C4's
      // first iteration, where every term of a gradient sum has the same sign) and replaces this record by a level
      // crossing's if the corrected guesses say the tile leaves the binade, or lies in the other one.
      maybe_E = (m_lo < kMaybeMargin || m_hi > 0x7fffffu - kMaybeMargin) ? (int32_t)E : -1;
      if (maybe_E < 0) continue;
    }
    // The tile crosses a level -- this is where a record is most likely not to cover the true state (a
    // landing next to the level) -- or no window holds it (its sum changes sign, or runs through three
    // binades: a sum hovering around zero).  Such a tile needs four class chains per leaf and scans over
    // the leaves, five times the work of a plain one and what the launch used to wait for: it becomes a
    // JOB of strict_job_kernel (a workgroup of its own), with its guesses, windows and terms.
    unsigned slot = 0xffffffffu;
    // (the sums' jobs of one tile in different shards: the first tile is a job of every sum)
    const unsigned shard = (unsigned)((tile + 7 * row) % kAuxShards), per_shard = (unsigned)W.naux / kAuxShards;
    if (lane == 0) {
      const unsigned k = atomicAdd(&W.aux_count[shard * 32], 1u);
      slot = k < per_shard ? shard * per_shard + k : 0xffffffffu;
    }
    slot = (unsigned)rfl((int)slot);
    if (slot == 0xffffffffu) {  // no slot left (never seen): the chain kernel recomputes the tile from the pairs
      if (maybe_E < 0) {
        T.key = -1;    // (the first tile as well: its walk starts at 0.0f)
        T.cons = 0;
        if (lane == 0) W.recs[row * W.ntiles + tile] = T;
      }
      continue;
    }
    if (lane == 0) {
      JobDesc *J = W.jobs + slot;
      J->row = row;
      J->tile = tile;
      J->pad = maybe_E >= 0 ? (1 | (maybe_E << 8)) : 0;
    }
    float4 *dst = W.aux_terms + (size_t)slot * (kTile / 4);
    int lane_here = lane;  // (opaque: the eight store offsets are formed here, not kept in registers -- one of them in
    asm volatile("" : "+v"(lane_here));  // scratch -- through the whole row loop)
#pragma unroll
    for (int v = 0; v < kLeaf / 4; v++) dst[v * kLanes + lane_here] = s_terms[row][v * kLanes + lane_here];
  }
  if (threadIdx.x == 0 && (W.selfcheck & 2)) {  // measurement aid (PCGX_STRICT_TRACE), plain stores only
    W.stamps[tile * 16 + 0] = (unsigned long long)t_0;
    W.stamps[tile * 16 + 1] = (unsigned long long)t_1;
    W.stamps[tile * 16 + 2] = (unsigned long long)t_x0;
    W.stamps[tile * 16 + 5] = (unsigned long long)trace_clock(W);
  }
}

// One workgroup per slot that strict_sum_kernel handed out (the others leave at once): four waves, one
// class of every leaf's summary each, through LDS; then wave 0 composes the leaves forwards and writes
// the tile's record, wave 1 backwards (level crossing: the chain kernel finds the leaf that does not
// cover the state in one parallel step and carries on behind it), or wave 0 alone composes the runs of
// leaves under equal windows (no window: the chain kernel applies a run's last record).
constexpr int kJobBlock = 256;  // four waves: a class each, then the scans (role 0); candidate chains (roles 1, 2)
constexpr int kJobRoles = 3;
template <bool kRegs>
__device__ __forceinline__ uint32_t resolve_staged(uint32_t s, int kind, int32_t key, const LeafAux &A, int lane,
                                               const float4 *lds, const float *t, int &serial_out, int &tried_out,
                                               int &applied_out);
__global__ __launch_bounds__(kJobBlock) void strict_job_kernel(const IcpState *__restrict__ state, StrictWork W) {
  __shared__ float4 s_t[kTile / 4];    // the tile's terms, a leaf per lane (tile_quad)
  __shared__ float4 s_lin[kTile / 4];  // ... and leaf after leaf, for the chains that add them up one after the other
  __shared__ int32_t s_S[12][kLanes];  // class pieces c[4] | lo[4] | hi[4] per leaf
  __shared__ uint32_t s_g[kLanes];     // the leaves' guesses
  __shared__ int32_t s_lk[kLanes];     // and the windows they are summarised under
  __shared__ int32_t s_hdr[5];         // the tile's window, guess at its start, end of its last guess chain, "the chains join up", "the plain record stands"
  __shared__ uint32_t s_ctab[4][kCandInner];  // waves 4..7: the ends of their quarter's candidates
  __shared__ int s_cdone, s_pieces;
  const unsigned per_shard = (unsigned)W.naux / kAuxShards;
  const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
  // THREE workgroups of four waves per slot that could be handed out (most leave at once): role 0 for the tile's
  // record and leaf records, role 1 for its candidate table (a level crossing's four quarters; four of the six waves
  // of a tile without a window), role 2 for the other two waves of a tile without a window.  As one workgroup of
  // eight waves the candidate chains finished 8.7-11 us after the terms were in LDS and the launch waited for them
  // (7.0 now), and a CU held two jobs; it holds four workgroups now, so every job of a launch starts at once -- with
  // two per CU the fifth job of a shard waited for a place, and ended at 15 us.  Every role forms the guesses
  // (wave 0) for itself.
  // (A loop over the slots of a shard, tried twice: the loop alone takes the kernel from 67 to 146 VGPRs, and with
  // fewer workgroups per CU the jobs queue.)
  // (blocks 3 b .. 3 b + 2 look after slot b / kAuxShards of shard b % kAuxShards: the slots a launch hands out -- the
  // first few of every shard -- are the blocks dispatched first, whatever the number of blocks that leave at once
  // behind them)
  const int role = (int)(blockIdx.x % kJobRoles);
  const unsigned jb = blockIdx.x / kJobRoles;
  const unsigned shard = jb % kAuxShards, slot = shard * per_shard + jb / kAuxShards;
  // ("done", the shard's count, and -- for the first slots of a shard, where the jobs of a launch are -- the job's
  // description and terms in ONE round trip: the guesses behind them are two more dependent ones as it is)
  const JobDesc *J = W.jobs + slot;
  const float4 *src4 = W.aux_terms + (size_t)slot * (kTile / 4);
  static_assert(kTile / 4 == 2 * kJobBlock, "two quads of the tile per thread");
  const bool early = jb / kAuxShards < 6;  // uniform
  float4 q0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), q1 = q0;
  int row = 0, pad = 0;
  int64_t tile = 0;
  if (early) {
    q0 = src4[threadIdx.x];
    q1 = src4[threadIdx.x + kJobBlock];
    row = J->row;
    tile = J->tile;
    pad = J->pad;
  }
  const int done = state->done;
  const unsigned handed_out = W.aux_count[shard * 32];
  if (done || slot % per_shard >= min(handed_out, per_shard)) return;  // uniform
  if (threadIdx.x == 0) s_cdone = s_pieces = 0;
  if (!early) {
    q0 = src4[threadIdx.x];
    q1 = src4[threadIdx.x + kJobBlock];
    row = J->row;
    tile = J->tile;
    pad = J->pad;
  }
  {
    int i = threadIdx.x;  // quad v = i / 64 of leaf (i % 64) ^ v (tile_quad)
    s_t[i] = q0;
    s_lin[8 * ((i & 63) ^ (i >> 6)) + (i >> 6)] = q0;
    i += kJobBlock;
    s_t[i] = q1;
    s_lin[8 * ((i & 63) ^ (i >> 6)) + (i >> 6)] = q1;
  }
  const long long tj_0 = trace_clock(W);
  __syncthreads();
  const long long tj_1 = trace_clock(W);
  auto stamp_end = [&](int what) {
    if ((W.selfcheck & 2) && lane == 0) {
      W.stamps[tile * 16 + 6] = (unsigned long long)tj_0;
      W.stamps[tile * 16 + 7] = (unsigned long long)tj_1;
      atomicMax(&W.stamps[tile * 16 + 8 + (part >= 4 ? 1 : 0)], (unsigned long long)trace_clock(W));
      W.stamps[tile * 16 + 10] = (unsigned long long)(what | (row << 8));
    }
  };
  // The first tile of a row starts from 0.0f, and exactly so (evaluator.go:122): its job carries the tile out from
  // there.  Its sum runs through a binade every few terms at first, so no window holds the tile -- it is summarised
  // leaf by leaf like any tile without one, and wave 0 then walks the leaves' runs itself: the few leaves at the
  // start whose windows do not hold the state are added term by term (C4: 1 to 7 of the 64), the others are one
  // apply() per run.  (Round 3 added all 2048 terms one after the other: 10 us of dependent additions.)
  const bool first = tile < kExactTiles && W.first_exact;  // uniform
  if (first && role != 0) return;  // uniform (no table: the state the tile starts from is known)
  // (PCGX_STRICT_TRACE: wave 0's way through the job, words 11 .. 15 of the tile's line -- the rows' own stamps are in
  // the first lines, strict_chain_kernel)
  const bool traced = (W.selfcheck & 2) && part == 0 && !first && tile > kStrictRows;
  auto stage = [&](int word, uint32_t dep) {
    if (traced) {
      const unsigned long long tt = clock_after(dep);
      if (lane == 0) W.stamps[tile * 16 + word] = tt;
    }
  };
  // ---- wave 0: the leaves' guesses once more, now from a start state that includes the rounding errors
  // the float32 chain has made in all tiles before this one (strict_sum_kernel's tile_err, known only
  // after that launch): where a sum has come back towards zero the float64 prefix alone is off by 10^5
  // ulps of the small state (the chain's roundings were made at larger magnitudes), and leaf records
  // made from such guesses cover the true state half of the time; with the errors added the guess is
  // within a few ulps.  Then the tile's window (or its leaves' own) as strict_sum_kernel chose them.
  if (part == 0) {
    const LdsQuads q{s_t, lane};
    double lsum = 0.0;
#pragma unroll
    for (int v = 0; v < kLeaf / 4; v++) {
      const float4 a = q(v);
      lsum += (double)a.x; lsum += (double)a.y; lsum += (double)a.z; lsum += (double)a.w;
    }
    const double pre = wave_excl_scan_f64(lsum, lane);
    const double base = tile_prefix2(W.tile_sum, W.tile_err, W.ntiles, row, tile, lane) +
                        (W.row_base ? W.row_base[row] + W.err_base[row] : 0.0);  // (+ the ranks before this one)
    uint32_t g;
    ChainRange cr;
    double terr;
    tile_guesses(q, base, lsum, pre, lane, g, cr, terr);
    const uint32_t mn = wave_all_umin(cr.mn), mx = wave_all_umax(cr.mx);
    const bool one_sign = __ballot(cr.sg_or != cr.sg_and) == 0ull &&
                          (__ballot(cr.sg_or != 0u) == 0ull || __ballot(cr.sg_or == 0u) == 0ull);
    const uint32_t g_first = (uint32_t)rfl((int)g);
    const int32_t key = (one_sign && !first) ? choose_window(mn, mx, g_first >> 31, g_first & 0x7fffffffu) : -1;
    const uint32_t g_next = (uint32_t)lane_next((int)g, (int)g);  // (lane 63's is not looked at)
    const bool cons = __ballot(lane < 63 && g_next != cr.end) == 0ull;
    s_g[lane] = g;
    s_lk[lane] = key >= 0 ? key : leaf_key(cr, g);
    const uint32_t out_last = (uint32_t)__builtin_amdgcn_readlane((int)cr.end, 63);
    // a plain tile handed over because it lies near an end of its binade (strict_sum_kernel): with the drift added
    // the guesses say whether its record stands -- still one binade, the same one, clear of the ends
    const uint32_t m_lo = mn & 0x7fffffu, m_hi = mx & 0x7fffffu;
    const bool stands = (pad & 1) && key >= 0 && (mn >> 23) == (mx >> 23) && (int32_t)(mn >> 23) == (pad >> 8) &&
                        m_lo >= kPlainMargin && m_hi <= 0x7fffffu - kPlainMargin;
    stage(11, g ^ (uint32_t)key);
    if (lane == 0) {
      s_hdr[0] = key;
      s_hdr[1] = (int32_t)g_first;
      s_hdr[2] = (int32_t)out_last;
      s_hdr[3] = cons ? 1 : 0;
      s_hdr[4] = stands ? 1 : 0;
    }
  }
  __syncthreads();
  if (s_hdr[4]) { if (part == 0 && role == 0) stamp_end(4); return; }  // uniform
  const uint32_t g = s_g[lane];
  const int32_t lk = s_lk[lane];
  const int32_t tkey = s_hdr[0];
  const int kind = tkey >= 0 ? JOB_CROSSING : JOB_NOWINDOW;
  // (no workgroup barrier from here on: waves 4..7 start their candidate chains at once, the waves that scan wait for
  // the four class pieces through an LDS counter)
  if (part < 4 && role == 0) {
    int32_t c, lo, hi;
    leaf_class_piece_q(LdsQuads{s_t, lane}, g, lk, part, c, lo, hi);
    s_S[part][lane] = c;
    s_S[4 + part][lane] = lo;
    s_S[8 + part][lane] = hi;
    stage(12, (uint32_t)(c ^ lo ^ hi));
    lds_fence_wave();
    if (lane == 0) __hip_atomic_fetch_add(&s_pieces, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  const bool scans = role == 0 && (part == 0 || (part == 1 && kind == JOB_CROSSING));
  if (scans)
    while (__hip_atomic_load(&s_pieces, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < 4) __builtin_amdgcn_s_sleep(1);
  // ---- the tile's additions carried out from candidate start states (strict_terms.h), by the waves with nothing
  // else to do.  A tile without a window costs the chain kernel 6-8 us however its leaves are summarised (a dozen
  // leaf runs to apply, a dozen leaves to add term by term), a level crossing whose record does not cover the state
  // 1-2 us; if the walker's state is one of the candidates the tile is a look-up.  Four candidates per lane: four
  // independent chains cost 1.9x one (tools/micro/dep_add.cpp).
  if (first && part >= 1) return;  // (no candidates: the state the tile starts from is known)
  if (kind == JOB_NOWINDOW && (role != 0 || part >= 1)) {
    if (role == 0) return;  // (waves 1 .. 3 of the records' workgroup: their class pieces are in)
    // no window: waves 1..6, the whole tile from 768 candidates around the tile's guess, two per lane (two
    // independent chains cost 1.3x one, four 1.9x, tools/micro/dep_add.cpp: 7-8 us where three waves with four
    // each took 10-11 -- the launch is that much longer in the few iterations that have such tiles; cut into
    // quarters like a level crossing below, a third of these tiles were lost where one quarter's end fell outside
    // the next quarter's candidates)
    constexpr int kPer = 2, kWavesNw = kCand / (kPer * kLanes);
    static_assert(kWavesNw * kPer * kLanes == kCand && kWavesNw <= (kJobRoles - 1) * (kJobBlock / 64), "roles 1, 2 carry the table");
    const int wave_nw = (role - 1) * (kJobBlock / 64) + part;  // 0 .. kWavesNw - 1: this wave's part of the table
    if (wave_nw >= kWavesNw) return;
    const uint32_t g0 = (uint32_t)s_hdr[1];
    const uint32_t mag = g0 & 0x7fffffffu;
    uint32_t out[kPer];
#pragma unroll
    for (int c = 0; c < kPer; c++) out[c] = 0x7fc00000u;  // "no table"
    const int i0 = wave_nw * kPer * kLanes + lane;
    if (mag > kCandReach && mag < 0x7f800000u - kCandReach) {  // uniform
      float x[kPer];
#pragma unroll
      for (int c = 0; c < kPer; c++) x[c] = u2f(g0 + (uint32_t)cand_offset(i0 + c * kLanes));
      serial_leaves_lin(x, s_lin, 0, kLanes);
#pragma unroll
      for (int c = 0; c < kPer; c++) out[c] = f2u(x[c]);
    }
#pragma unroll
    for (int c = 0; c < kPer; c++) W.cand[(size_t)slot * kCand + i0 + c * kLanes] = out[c];
    stamp_end(2);
    return;
  }
  if (role == 2) return;  // uniform (a level crossing has no second half of a table)
  if (role == 1) {
    // level crossing: 2048 dependent additions are 6-10 us -- longer than anything else this launch does -- so
    // the four waves take a QUARTER of the tile each: 16 leaves from the 256 candidates around that quarter's first
    // leaf's guess, and wave 0 strings the quarters' tables together: a candidate's end in one quarter is looked up
    // among the candidates of the next (the guesses of such a tile agree with one another to a few floats: they
    // were refined by the chains' own rounding errors)
    const int k = part;
    const uint32_t gk = s_g[k * (kLanes / 4)];
    const uint32_t mag = gk & 0x7fffffffu;
    constexpr int kMid = (kCand - kCandInner) / 2;  // the table's entries [kMid, kMid + kCandInner)
    uint32_t out[kCandPerLane];
#pragma unroll
    for (int c = 0; c < kCandPerLane; c++) out[c] = 0x7fc00000u;  // "no table"
    if (mag > kCandReach && mag < 0x7f800000u - kCandReach) {  // uniform
      float x[kCandPerLane];
#pragma unroll
      for (int c = 0; c < kCandPerLane; c++) x[c] = u2f(gk + (uint32_t)cand_offset(kMid + c * kLanes + lane));
      serial_leaves_lin(x, s_lin, k * (kLanes / 4), (k + 1) * (kLanes / 4));
#pragma unroll
      for (int c = 0; c < kCandPerLane; c++) out[c] = f2u(x[c]);
    }
#pragma unroll
    for (int c = 0; c < kCandPerLane; c++) s_ctab[k][c * kLanes + lane] = out[c];
    if (k == 1 || k == 2) {  // the table's entries below / above the middle: none
      static_assert(kMid % kLanes == 0, "whole waves of entries on either side of the middle");
#pragma unroll
      for (int c = 0; c < kMid / kLanes; c++)
        W.cand[(size_t)slot * kCand + (k == 1 ? 0 : kMid + kCandInner) + c * kLanes + lane] = 0x7fc00000u;
    }
    lds_fence_wave();
    if (lane == 0) __hip_atomic_fetch_add(&s_cdone, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (k != 0) return;
    while (__hip_atomic_load(&s_cdone, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < 4) __builtin_amdgcn_s_sleep(1);
#pragma unroll
    for (int c = 0; c < kCandPerLane; c++) {
      uint32_t v = out[c];
#pragma unroll
      for (int q = 1; q < 4; q++) {
        const uint32_t gq = s_g[q * (kLanes / 4)];
        const int32_t idx = (int32_t)((v & 0x7fffffffu) - (gq & 0x7fffffffu)) + kCandInner / 2;
        const bool ok = (v & 0x7f800000u) != 0x7f800000u && ((v ^ gq) >> 31) == 0u && idx >= 0 && idx < kCandInner;
        const uint32_t next = s_ctab[q][ok ? idx : 0];
        v = ok ? next : 0x7fc00000u;  // (a quarter without a table holds NaNs: they carry through)
      }
      W.cand[(size_t)slot * kCand + kMid + c * kLanes + lane] = v;
    }
    stamp_end(1);
    return;
  }
  if (part >= 2) return;
  Summary S;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    S.c[r] = s_S[r][lane];
    S.lo[r] = s_S[4 + r][lane];
    S.hi[r] = s_S[8 + r][lane];
  }
  stage(13, (uint32_t)(S.c[0] ^ S.c[3] ^ S.hi[3]));
  TileRec R;
  R.s = summary_identity();
  R.key = tkey;
  R.in = (uint32_t)s_hdr[1];
  R.out = (uint32_t)s_hdr[2];
  R.cons = s_hdr[3] | (int32_t)((slot + 1u) << 8);
  if (kind == JOB_CROSSING) {
    if (part == 0) {
      Summary P = S;
      // (the leaves before, by DPP: inside the rows of 16 lanes by 1, 2, 4, 8, then the rows before through their last lanes)
      // (the partner's summary is fetched by ALL lanes, in front of the condition: a DPP read of a lane that is switched
      // off returns the reader's own value)
      auto before = [&](const Summary &X, bool takes) {
        if (takes) P = compose(X, P);
      };
      before(dpp_summary<0x111, 0xf>(P), (lane & 15) >= 1);
      before(dpp_summary<0x112, 0xf>(P), (lane & 15) >= 2);
      before(dpp_summary<0x114, 0xf>(P), (lane & 15) >= 4);
      before(dpp_summary<0x118, 0xf>(P), (lane & 15) >= 8);
      before(dpp_summary<0x142, 0xa>(P), (lane & 16) != 0);
      before(dpp_summary<0x143, 0xc>(P), lane >= 32);
      stage(14, (uint32_t)(P.c[0] ^ P.hi[3]));
      W.aux[(size_t)slot * kLanes + lane].pre = P;
      R.s = shfl_summary(P, 63);
      if (lane == 0) W.recs[row * W.ntiles + tile] = R;
    } else {
      Summary Q = S;
      // (the leaves behind: inside the rows of 16 lanes by DPP, row_shl; then rows 0 and 2 take what the first lane of the
      // row behind them has, rows 0 and 1 what lane 32 has)
      auto behind = [&](const Summary &Y, bool takes) {  // (fetched by all lanes, in front of the condition)
        if (takes) Q = compose(Q, Y);
      };
      behind(dpp_summary<0x101, 0xf>(Q), (lane & 15) + 1 < 16);
      behind(dpp_summary<0x102, 0xf>(Q), (lane & 15) + 2 < 16);
      behind(dpp_summary<0x104, 0xf>(Q), (lane & 15) + 4 < 16);
      behind(dpp_summary<0x108, 0xf>(Q), (lane & 15) + 8 < 16);
      {
        const Summary Y = shfl_summary(Q, (lane & ~15) + 16);
        if (!(lane & 16)) Q = compose(Q, Y);
      }
      {
        const Summary Y = shfl_summary(Q, 32);
        if (lane < 32) Q = compose(Q, Y);
      }
      W.aux[(size_t)slot * kLanes + lane].suf = Q;
    }
  } else if (part == 0) {
    const int32_t k_prev = lane_prev(lk, -1);
    Summary P = S;
    int fp = (lane == 0 || lk < 0 || lk != k_prev) ? 1 : 0;
    auto step = [&](const Summary &X, int xf, bool takes) {
      if (takes && !fp) {
        P = compose(X, P);
        fp = xf;
      }
    };
#define PCGX_FP(CTRL, MASK) dpp_partner<CTRL, MASK>(fp)
    step(dpp_summary<0x111, 0xf>(P), PCGX_FP(0x111, 0xf), (lane & 15) >= 1);
    step(dpp_summary<0x112, 0xf>(P), PCGX_FP(0x112, 0xf), (lane & 15) >= 2);
    step(dpp_summary<0x114, 0xf>(P), PCGX_FP(0x114, 0xf), (lane & 15) >= 4);
    step(dpp_summary<0x118, 0xf>(P), PCGX_FP(0x118, 0xf), (lane & 15) >= 8);
    step(dpp_summary<0x142, 0xa>(P), PCGX_FP(0x142, 0xa), (lane & 16) != 0);
    step(dpp_summary<0x143, 0xc>(P), PCGX_FP(0x143, 0xc), lane >= 32);
#undef PCGX_FP
    LeafRec L;
    L.key = lk;
    L.pad[0] = L.pad[1] = L.pad[2] = 0;
    L.run = P;
    stage(14, (uint32_t)(P.c[0] ^ P.hi[3]));
    if (first) {
      LeafAux A;
      __builtin_memset(&A, 0, sizeof A);
      __builtin_memcpy(&A, &L, sizeof L);  // (resolve_staged reads a LeafRec out of a LeafAux)
      int serial, tried, applied;
      R.key = -1;
      R.in = f2u(0.0f);
      R.out = resolve_staged<false>(f2u(0.0f), JOB_NOWINDOW, -1, A, lane, s_t, nullptr, serial, tried, applied);
      R.cons = 1;  // (a point record that owns nothing the chain kernel would fetch)
      if (lane == 0) W.recs[row * W.ntiles + tile] = R;
      stamp_end(3);
      return;
    }
    *reinterpret_cast<LeafRec *>(&W.aux[(size_t)slot * kLanes + lane]) = L;
    R.key = -1;
    if (lane == 0) W.recs[row * W.ntiles + tile] = R;
  }
}

// ---- repair (the first Evaluate of a Fit) ---------------------------------------------------------------------------
// A plain tile's record covers the states that stay inside the tile's binade; strict_sum_kernel hands a tile that lies
// within kMaybeMargin floats of a binade's end to strict_job_kernel as well, which knows how far the float32 chain has
// drifted from the float64 sums by then.  In the FIRST Evaluate of a Fit (the raw target: every term of a gradient sum
// has the same sign) the drift outgrows that margin -- 10^4 floats by the end of C4's rows, 10^5 and more at C5's 8M
// targets -- and the tiles in which a row passes a binade's end within that drift of the guess own no slot: the
// walker forms their terms again from the pairs and adds all 2048, 25 us each, on the launch's critical path (one or
// two tiles at C4; 55 of them, 1.3 ms, at C5's share).  This kernel runs between the summaries and the jobs, in that
// iteration only: one workgroup per sum predicts every plain tile's start state as the job kernel would (float64
// prefix of the tiles' sums + of their chains' rounding errors) and tests the tile's record on it, kRepairMargin
// floats either side; a tile that does not cover them becomes a job like any level crossing -- slot, description,
// terms formed again from the pairs, by a whole wave, off the critical path.
constexpr int kRepairBlock = 512;  // a workgroup per (sum, 512 tiles), a tile per thread
constexpr uint32_t kRepairMargin = 64u;
// (rows of fewer tiles are left alone: at C4's 489 tiles the pass costs 36 us in the first iteration and saves nothing
// measurable -- the one or two tiles it catches are not what that launch waits for; at C5's 3907 it takes the first
// iteration's chain kernel from 1.3 ms to 0.26)
constexpr int64_t kRepairMinTiles = 1024;
__global__ __launch_bounds__(kRepairBlock) void strict_repair_kernel(const float4 *__restrict__ match, const uint32_t *__restrict__ pos_of,
                                                                     const IcpState *__restrict__ state, StrictWork W) {
  __shared__ double s_before[kRepairBlock / 64], s_wtot[kRepairBlock / 64];
  __shared__ int s_list[kRepairBlock];
  __shared__ int s_n;
  const int row = (int)(blockIdx.x % (unsigned)W.nrows), part = (int)(blockIdx.x / (unsigned)W.nrows);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (state->done || state->iter != 0) return;  // uniform
  if (threadIdx.x == 0) s_n = 0;
  const double *ts = W.tile_sum + (int64_t)row * W.ntiles, *te = W.tile_err + (int64_t)row * W.ntiles;
  const int64_t first = (int64_t)part * kRepairBlock, t = first + threadIdx.x;
  // what the parts before this one add up to (every workgroup for itself: a few thousand doubles), then the scan inside
  double before = 0.0;
  for (int64_t k = threadIdx.x; k < first; k += kRepairBlock) before += ts[k] + te[k];
  before = wave_allsum_f64(before);
  const double mine = t < W.ntiles ? ts[t] + te[t] : 0.0;
  const double pre = wave_excl_scan_f64(mine, lane);
  if (lane == 63) {
    s_before[wave] = before;
    s_wtot[wave] = pre + mine;
  }
  __syncthreads();
  double base = pre + (W.row_base ? W.row_base[row] + W.err_base[row] : 0.0);  // (+ the ranks before this one)
  for (int w = 0; w < kRepairBlock / 64; w++) base += s_before[w] + (w < wave ? s_wtot[w] : 0.0);
  if (t < W.ntiles) {
    const TileRec T = W.recs[(int64_t)row * W.ntiles + t];
    const bool plain = T.key >= 0 && (T.cons >> 8) == 0 && !(t < kExactTiles && W.first_exact);
    if (plain) {
      const uint32_t g = f2u((float)base), mag = g & 0x7fffffffu;
      bool ok = mag > 2u * kRepairMargin && mag < 0x7f800000u - 2u * kRepairMargin;
#pragma unroll
      for (int j = 0; j < 4; j++) {  // (every parity class at either end of the margin)
        uint32_t a = g - kRepairMargin + (uint32_t)j, b = g + kRepairMargin - (uint32_t)j;
        ok = ok && apply(a, T.key, T.s) && apply(b, T.key, T.s);
      }
      if (!ok) s_list[atomicAdd(&s_n, 1)] = (int)threadIdx.x;
    }
  }
  __syncthreads();
  const int n = s_n;
  if (n == 0) return;  // uniform
  const TermSrc S = make_term_src(match, pos_of, state, W);
  const unsigned per_shard = (unsigned)W.naux / kAuxShards;
  for (int i = wave; i < n; i += kRepairBlock / 64) {  // uniform per wave
    const int64_t tile = first + s_list[i];
    unsigned slot = 0xffffffffu;
    if (lane == 0) {
      const unsigned shard = (unsigned)((tile + 7 * row) % kAuxShards);
      const unsigned k = atomicAdd(&W.aux_count[shard * 32], 1u);
      slot = k < per_shard ? shard * per_shard + k : 0xffffffffu;
    }
    slot = (unsigned)rfl((int)slot);
    if (slot == 0xffffffffu) continue;  // (no slot left: the walker's own way, as before)
    if (lane == 0) {
      JobDesc *J = W.jobs + slot;
      J->row = row;
      J->tile = tile;
      J->pad = 0;
      atomicAdd(&W.dbg[58], 1ull);
    }
    recompute_tile_to_lds(S, row, tile, lane, W.aux_terms + (size_t)slot * (kTile / 4));  // (the same layout: tile_quad)
  }
}

// ---- chain -----------------------------------------------------------------------------------------
__device__ __forceinline__ bool apply_point(uint32_t &s, const TileRec &R) {
  if ((R.cons & 1) && R.in == s) {
    s = R.out;
    return true;
  }
  return false;
}

// One tile, exactly, from the known state s (every lane holds it; returns it in every lane).  The
// tile's terms are staged in `lds` (layout of tile_quad).
//  kind JOB_CROSSING, aux = the leaf's LeafAux: the first leaf whose prefix composition does not cover
//    s is found by all lanes at once, that leaf is added term by term, and the suffix composition
//    behind it finishes the tile;
//  kind JOB_NOWINDOW, aux reread as the leaf's LeafRec: the runs of leaves under equal windows are
//    applied one after the other; a run that does not cover s, and a leaf without a window, are added
//    term by term;
//  kind 0 (a tile that owns no slot): all 2048 terms one after the other.

// (helpers: the leaf of every lane in its registers, t; the walker's tile without a slot: staged in lds, t unused)
template <bool kRegs>
__device__ __forceinline__ uint32_t resolve_staged(uint32_t s, int kind, int32_t key, const LeafAux &A, int lane,
                                               const float4 *lds, const float *t, int &serial_out, int &tried_out,
                                               int &applied_out) {
  auto add_leaves = [&](uint32_t x, int l0, int l1) {
    if (!kRegs) return serial_leaves(x, lds, l0, l1);
    for (int l = l0; l < l1; l++) x = serial_leaf_regs(x, t, l);  // uniform
    return x;
  };
  int serial = 0, n_try = 0, n_ok = 0;
  if (kind == JOB_NOWINDOW) {
    const int32_t lkey = A.pre.c[0];  // LeafRec{key, pad[3], run} laid over LeafAux{pre, suf}
    Summary run;
    run.c[0] = A.pre.lo[0]; run.c[1] = A.pre.lo[1]; run.c[2] = A.pre.lo[2]; run.c[3] = A.pre.lo[3];
    run.lo[0] = A.pre.hi[0]; run.lo[1] = A.pre.hi[1]; run.lo[2] = A.pre.hi[2]; run.lo[3] = A.pre.hi[3];
    run.hi[0] = A.suf.c[0]; run.hi[1] = A.suf.c[1]; run.hi[2] = A.suf.c[2]; run.hi[3] = A.suf.c[3];
    const int32_t k_next = __shfl_down(lkey, 1);
    const unsigned long long tails = __ballot(lane == 63 || lkey < 0 || lkey != k_next);
    int l = 0;
    while (l < kLanes) {  // uniform
      n_try++;
      const int e = l + __builtin_ctzll(tails >> l);
      const int32_t rk = __builtin_amdgcn_readlane(lkey, e);
      if (rk >= 0) {
        Summary R;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          R.c[r] = __builtin_amdgcn_readlane(run.c[r], e);
          R.lo[r] = __builtin_amdgcn_readlane(run.lo[r], e);
          R.hi[r] = __builtin_amdgcn_readlane(run.hi[r], e);
        }
        if (apply(s, rk, R)) {  // (the record of lane e composes its run from the run's head, where the walk stands)
          n_ok++;
          l = e + 1;
          continue;
        }
      }
      s = add_leaves(s, l, e + 1);
      serial += e + 1 - l;
      l = e + 1;
    }
  } else if (kind == JOB_CROSSING) {
    uint32_t mine = s;
    const bool ok = apply(mine, key, A.pre);
    const unsigned long long bad = __ballot(!ok);
    int l = bad ? __builtin_ctzll(bad) : kLanes;  // leaves 0 .. l-1 are covered
    if (l > 0) s = (uint32_t)__builtin_amdgcn_readlane((int)mine, l - 1);
    while (l < kLanes) {
      s = add_leaves(s, l, l + 1);
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
    s = add_leaves(s, 0, kLanes);
    serial = kLanes;
  }
  serial_out = serial;
  tried_out = n_try;
  applied_out = n_ok;
  return (uint32_t)rfl((int)s);
}
static_assert(sizeof(LeafRec) == 64 && offsetof(LeafRec, run) == 16, "resolve_staged reads a LeafRec out of a LeafAux");

__device__ __forceinline__ void resolve_stats(const StrictWork &W, int kind, int serial, int tried, int applied, long long ticks) {
  atomicAdd(&W.dbg[2], 1ull);
  atomicAdd(&W.dbg[3], (unsigned long long)serial);
  if (kind == 0) atomicAdd(&W.dbg[5], 1ull);
  atomicAdd(&W.dbg[kind ? 10 : 11], (unsigned long long)ticks);
  if (kind == JOB_NOWINDOW) {  // tiles without a window: leaf runs tried / applied, ticks
    atomicAdd(&W.dbg[16], 1ull);
    atomicAdd(&W.dbg[17], (unsigned long long)tried);
    atomicAdd(&W.dbg[18], (unsigned long long)applied);
    atomicAdd(&W.dbg[19], (unsigned long long)ticks);
    atomicAdd(&W.dbg[22], (unsigned long long)serial);
  } else if (kind == JOB_CROSSING) {
    atomicAdd(&W.dbg[20], 1ull);
    atomicAdd(&W.dbg[21], (unsigned long long)ticks);
    atomicAdd(&W.dbg[23], (unsigned long long)serial);
  }
}

// debugging aid (W.selfcheck): the state after tiles [a, b) from `before`, term by term; mismatches
// against what the walk produced are counted per path in dbg[12 + path]
// (kCheck: the chain kernel is built twice -- with these loops inlined at five places of the walk it is 130 KB of
// code, and the walker of every launch started on a cold instruction cache, jumping over them)
template <bool kCheck, class MakeSrc>
__device__ __forceinline__ void selfcheck(const StrictWork &W, MakeSrc make_src, int row, uint32_t before, uint32_t after,
                                          int64_t a, int64_t b, int path, int lane, float4 *lds) {
  if (!kCheck) return;
  if (!(W.selfcheck & 1)) return;
  const TermSrc src = make_src();
  uint32_t x = before;
  for (int64_t k = a; k < b; k++) {
    __builtin_amdgcn_wave_barrier();
    recompute_tile_to_lds(src, row, k, lane, lds);
    x = serial_leaves(x, lds, 0, kLanes);
    __builtin_amdgcn_wave_barrier();
  }
  if (x != after && lane == 0) atomicAdd(&W.dbg[12 + path], 1ull);
}

// One workgroup per sum: a WALKER wave that applies the runs of equal windows to the exact state, one
// after the other, and eight HELPER waves.  The helpers first compose the tiles' records into runs
// (segmented scans over 64 tiles each, forwards -- then the walk starts -- and backwards).  Then each
// takes the tiles that own a slot (level crossings, no window: where a record is most likely not to
// cover the state) round robin, fetches such a tile's leaf records into registers and its terms into
// LDS AHEAD of the walk, and waits: when the walker finds the tile's record does not cover its state
// it hands the state over (an LDS mailbox) and gets the state behind the tile back; when the walker
// passes the tile without trouble the helper goes on to its next one.  Round 2's walker fetched a
// tile's 14 KB itself, after the failure: ~2.5 of the ~3.5 us a recomputed tile cost were that fetch,
// on the critical path of a launch that waits for its slowest sum.
#ifndef PCGX_SPEC_PER
#define PCGX_SPEC_PER 4
#endif
constexpr int kChainSegs = 8;                 // waves = segments of 64 tiles per chunk
constexpr int kChainTiles = kChainSegs * 64;  // tiles per chunk
constexpr int kChainBlock = kChainTiles;
constexpr int kWalker = kChainSegs - 1;       // the wave that walks (after the forward scan of its own segment)
constexpr int kTabSlots = 2 * (kChainSegs - 1);  // candidate tables in LDS: two per helper
constexpr int kHelpers = kChainSegs - 1;      // the others (eight waves, two per SIMD: 256 registers each, no scratch)
constexpr long long kChunkWaitTicks = 5000000;  // 50 ms (s_memrealtime runs at 100 MHz): a later chunk's wait for its start state

// The chunk's records in LDS, one array per word ([word][tile]): lane-indexed reads and writes of whole records
// then touch 64 consecutive words per instruction.  As an array of 64-byte structs every such access was a
// 16-way bank conflict (lanes l and l + 4 on the same bank) -- the scans' stores and the walker's gathers.
typedef int32_t RecArray[16][kChainTiles];
__device__ __forceinline__ void rec_put(RecArray &a, int i, const TileRec &R) {
  a[0][i] = R.key; a[1][i] = (int32_t)R.in; a[2][i] = (int32_t)R.out; a[3][i] = R.cons;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    a[4 + q][i] = R.s.c[q];
    a[8 + q][i] = R.s.lo[q];
    a[12 + q][i] = R.s.hi[q];
  }
}
__device__ __forceinline__ TileRec rec_get(const RecArray &a, int i) {
  TileRec R;
  R.key = a[0][i]; R.in = (uint32_t)a[1][i]; R.out = (uint32_t)a[2][i]; R.cons = a[3][i];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    R.s.c[q] = a[4 + q][i];
    R.s.lo[q] = a[8 + q][i];
    R.s.hi[q] = a[12 + q][i];
  }
  return R;
}
// a record every lane of the walking wave reads from the same LDS address, as scalars
__device__ __forceinline__ TileRec load_rec_uniform(const RecArray &a, int i) {
  const TileRec V = rec_get(a, i);  // sixteen LDS reads in flight, one wait
  TileRec R;
  R.key = rfl(V.key); R.in = (uint32_t)rfl((int)V.in); R.out = (uint32_t)rfl((int)V.out); R.cons = rfl(V.cons);
#pragma unroll
  for (int q = 0; q < 4; q++) {
    R.s.c[q] = rfl(V.s.c[q]);
    R.s.lo[q] = rfl(V.s.lo[q]);
    R.s.hi[q] = rfl(V.s.hi[q]);
  }
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
// A record from the lane a DPP control names (row_shr:n inside the rows of 16 lanes, row_bcast:15 / :31 across them): a
// register move per word, no trip through the LDS crossbar and no wait for it (ds_bpermute: shfl_rec) -- the forward scan
// of the chain kernel's prologue is six such steps on the walk's critical path.
template <int kCtrl, int kRowMask>
__device__ __forceinline__ int dpp_int(int v) {
  return dpp_partner<kCtrl, kRowMask>(v);
}
template <int kCtrl, int kRowMask>
__device__ __forceinline__ TileRec dpp_rec(const TileRec &R) {
  TileRec X;
  X.key = dpp_int<kCtrl, kRowMask>(R.key);
  X.in = (uint32_t)dpp_int<kCtrl, kRowMask>((int)R.in);
  X.out = (uint32_t)dpp_int<kCtrl, kRowMask>((int)R.out);
  X.cons = dpp_int<kCtrl, kRowMask>(R.cons);
#pragma unroll
  for (int q = 0; q < 4; q++) {
    X.s.c[q] = dpp_int<kCtrl, kRowMask>(R.s.c[q]);
    X.s.lo[q] = dpp_int<kCtrl, kRowMask>(R.s.lo[q]);
    X.s.hi[q] = dpp_int<kCtrl, kRowMask>(R.s.hi[q]);
  }
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

// A job tile with a window whose record covers few states (a level crossing: its classes' intervals end at the level)
// is a run of its own: inside a run of its neighbours it is what makes the run fail -- search for the tile, its record,
// its table, the rest of the run: 1 us -- while a run of one job tile is looked up in its candidate table at once.
// (The narrowest class interval below 64 mantissa steps: +3 runs a row at 0.1 us each, a third of the failed runs gone,
// 0.0838 -> 0.0830 ms; at 16 nothing changes, at 128 / 256 / 1024 / 16384 the extra runs -- +5 to +15 a row -- cost what
// the failures saved or more.  0: off.)
#ifndef PCGX_NARROW_RECORD
#define PCGX_NARROW_RECORD 64
#endif
constexpr int32_t kNarrowRecord = PCGX_NARROW_RECORD;
__device__ __forceinline__ bool stands_alone(const TileRec &R) {
  if (kNarrowRecord <= 0 || R.key < 0 || (R.cons >> 8) == 0) return false;
  int32_t w = kBig;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int32_t d = R.s.hi[r] - R.s.lo[r];
    w = d >= 0 ? imin(w, d) : w;
  }
  return w < kNarrowRecord || w == kBig;
}

// words shared between the waves of the chain kernel (LDS), read and written with workgroup-scope atomics
__device__ __forceinline__ int lds_get(const int *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_put(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }

struct ChainMail {  // walker <-> one helper
  int req;          // ordinal + 1 of the tile the walker wants recomputed
  int ack;          // ordinal + 1 of the tile the helper has recomputed
  uint32_t s_in, s_out;
};

// kSpec (rows of several chunks, a target spread over ranks: the launches whose walkers WAIT for a start state): a
// walker that has to wait walks ahead of the wait -- from the 256 states around the float64 guess of its start state, four
// per lane, through the runs' compositions, point records and candidate tables -- and when the real state arrives and is one
// of them, the chunk's end state is that candidate's: a hand-over then costs a hop (0.5 us), not a hop and a walk (6 us),
// and the walks of all chunks and all ranks run side by side instead of one after the other (spec_walk below).
template <bool kCheck, bool kSpec = false>
__global__ __launch_bounds__(kChainBlock) void strict_chain_kernel(const float4 *__restrict__ match,
                                                                  const uint32_t *__restrict__ pos_of,
                                                                  IcpState *__restrict__ state, StrictWork W,
                                                                  double *__restrict__ sums10, IcpKernelParams kp,
                                                                  int fuse_update) {
  __shared__ RecArray s_rec;   // the tiles' own records
  __shared__ RecArray s_pre;   // composition from the tile's run head to the tile
  __shared__ RecArray s_suf;   // composition from the tile to its run's tail
  __shared__ int16_t s_tail[kChainTiles];  // per segment: the tails of its runs, in order
  __shared__ int16_t s_head[kChainTiles];
  __shared__ int16_t s_auxlist[kChainTiles];  // the tiles that own a slot, in order
  __shared__ int16_t s_auxord[kChainTiles];   // a tile's place in that list, -1: owns no slot
  __shared__ int32_t s_count[kChainSegs], s_auxcnt[kChainSegs];
  __shared__ uint32_t s_auxmask[kChainSegs][2];  // which tiles of a segment own a slot
  __shared__ int s_sufok[kChainSegs];
  __shared__ int s_pre_cnt[2];             // forward scans done, of segments 0 .. 3 | 4 .. 7
  __shared__ int s_progress;               // walker: the tile it stands at (tiles below are done)
  __shared__ ChainMail s_mail[kChainSegs];
  // candidate tables (strict_job_kernel) of the job tiles the walk gets to next: helper h puts up its tiles' -- ordinals
  // h, h + 7, ... -- two ahead (slot = ordinal % kTabSlots; with one slot each the table of a helper's next tile went
  // up when the walker had passed the one before, ~1 us of loads: of fifty tiles a Fit handed to a helper through the
  // mailbox, forty-eight were in tables that were not up yet)
  __shared__ uint32_t s_tab[kTabSlots][kCand];
  __shared__ int s_tab_ord[kTabSlots];         // the ordinal + 1 of the tile whose table a slot holds (0: none yet)
  __shared__ float4 s_tile[kTile / 4];     // walker: the terms of a tile without a slot, formed again from the pairs
  __shared__ unsigned long long s_np;
  __shared__ int s_np_ok;
  __shared__ uint32_t s_stat[8];
  __shared__ unsigned long long s_wk[8];  // PCGX_STRICT_CLOCKS: where the walker's ticks go
  __shared__ unsigned long long s_stat_ticks[2];
  const int done = state->done;  // (looked at behind the first chunk's loads, which it would only hold up)
  // (rfl: "which wave" is the same in all lanes, and the compiler has to know -- or the branch between walker and
  // helpers counts as divergent, the walker's state becomes a vector register, and every apply() of the walk runs
  // on the vector unit at its dependent-instruction latency instead of on the scalar unit)
  // (blocks chunk-major: the walkers wait for LOWER block indices only, and an XCD hands out a launch's blocks in
  // ascending order -- the lowest unfinished chunk is resident or next in line whatever part of the grid is)
  const int row = (int)(blockIdx.x % (unsigned)W.nrows), my_chunk = (int)(blockIdx.x / (unsigned)W.nrows);
  const int lane = threadIdx.x & 63, wave = rfl((int)(threadIdx.x >> 6));
  const bool walker = wave == kWalker;
  const bool last_chunk = my_chunk == W.nchunks - 1;  // uniform: this workgroup's walker ends the row
  // (the pairs' source is put together where a tile is formed again from the pairs -- rare -- and not kept: its
  // eighteen words would sit in scalar registers through the whole walk, which runs on the scalar unit and had
  // 188 of its registers spilled into vector lanes)
  auto term_src = [&]() { return make_term_src(match, pos_of, state, W); };
  // walker state, evaluator.go:122: the sums start at zero -- or where the ranks before this one ended (a target
  // spread over ranks, strict_enqueue_sharded)
  // (a later chunk's walker: the state the chunk before it ended in, below; the ring form: what the rank before this
  // one sends, row_start() below)
  uint32_t s = W.start_bits ? (uint32_t)rfl((int)W.start_bits[row]) : f2u(0.0f);
  // the state the row starts in on this rank: 0.0f (evaluator.go:122), or where the rank before this one ended it
  int ring_rc = 0;  // walker: 1 = the ring was aborted, 2 = a wait ran out of time (the Fit ends here, on every rank)
  auto row_start = [&]() -> uint32_t {
    if (!W.ring || W.rank == 0) return W.start_bits ? (uint32_t)rfl((int)W.start_bits[row]) : f2u(0.0f);
    const RingLayout RL{W.world};
    uint32_t v = 0u;
    ring_rc = ring_wait(W, RL.start(row), v, kRingWalkTicks);
    if (ring_rc == 2 && lane == 0) ring_raise_abort(W, 0x100u | (uint32_t)row);  // (nobody else needs to wait that long)
    return (uint32_t)rfl((int)v);
  };
  // walker: counters of the whole row, written once behind the last chunk (an atomic in flight holds up the
  // next release store of its wave, and the walk is a chain of those).  Vector registers (opaque to the compiler):
  // as scalars they were written to and read from spill lanes around every run.
  uint32_t n_run = 0, n_runfail = 0, n_recfail = 0, n_tab_nw = 0, n_tab_cross = 0;
  uint32_t n_mail = 0;  // why a tile went to its helper, 8 bits each: table not up | state outside the table | no such candidate
  asm volatile("" : "+v"(n_run), "+v"(n_runfail), "+v"(n_recfail), "+v"(n_tab_nw), "+v"(n_tab_cross), "+v"(n_mail));
  unsigned long long ticks_scan = 0, ticks_walk = 0;
  const long long t_enter = trace_clock(W);  // (PCGX_STRICT_TRACE: the row's stamps)
  const unsigned long long c_enter = (W.selfcheck & 2) ? (unsigned long long)__builtin_readcyclecounter() : 0ull;
  {
    const int64_t chunk = (int64_t)my_chunk * kChainTiles;
    const long long t_a = stat_clock(W);
    if (threadIdx.x < kChainSegs) {
      s_sufok[threadIdx.x] = 0;
      s_mail[threadIdx.x].req = 0;
      s_mail[threadIdx.x].ack = 0;
    }
    if (threadIdx.x < kTabSlots) s_tab_ord[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
      s_progress = 0;
      s_pre_cnt[0] = s_pre_cnt[1] = 0;
      s_np_ok = 0;
    }
    if (threadIdx.x < 8) s_wk[threadIdx.x] = 0ull;
    // ---- helpers: runs of equal windows, segmented scan forwards inside each wave
    TileRec R;
    R.key = -2;
    R.in = R.out = 0u;
    R.cons = 0;
    R.s = summary_identity();
    bool head = true, tail = true, valid = false;
    {
      const int64_t tile = chunk + threadIdx.x;
      valid = tile < W.ntiles;
      if (valid) R = W.recs[row * W.ntiles + tile];
      if ((W.selfcheck & 2) && last_chunk && walker) {
        const unsigned long long tt = clock_after((uint32_t)R.key);
        if (lane == 0) s_wk[4] = tt;  // (PCGX_STRICT_TRACE: the row's records have arrived)
      }
      rec_put(s_rec, threadIdx.x, R);
      const int32_t key_prev = lane_prev(R.key, R.key), key_next = lane_next(R.key, R.key);  // (lane 0's / 63's: not looked at)
      const int alone = stands_alone(R) ? 1 : 0, alone_prev = lane_prev(alone, 0), alone_next = lane_next(alone, 0);
      head = lane == 0 || R.key < 0 || R.key != key_prev || alone || alone_prev;
      tail = lane == 63 || R.key < 0 || R.key != key_next || alone || alone_next;
      // (the runs' ends are known from the windows alone: the lists below do not wait for the compositions)
      const unsigned long long tails = __ballot(tail && valid);
      if (tail && valid) {
        const unsigned long long below = tails & ((1ull << lane) - 1ull);  // the run starts behind the tail before this one
        const int idx = wave * 64 + __popcll(below);
        s_tail[idx] = (int16_t)threadIdx.x;
        s_head[idx] = (int16_t)(wave * 64 + (below ? 64 - __builtin_clzll(below) : 0));
      }
      if (lane == 0) s_count[wave] = __popcll(tails);
      const unsigned long long aux = __ballot(valid && (R.cons >> 8) != 0);
      if (lane == 0) {
        s_auxcnt[wave] = __popcll(aux);
        s_auxmask[wave][0] = (uint32_t)aux;
        s_auxmask[wave][1] = (uint32_t)(aux >> 32);
      }
    }
    if (done) return;  // uniform (nothing but LDS has been written)
    __syncthreads();
    int naux = 0;
    {  // the list of tiles that own a slot
      int base = 0;
      for (int w = 0; w < kChainSegs; w++) {
        const int c = s_auxcnt[w];
        base += w < wave ? c : 0;
        naux += c;
      }
      const bool has = valid && (R.cons >> 8) != 0;
      const unsigned long long aux = __ballot(has);
      const int ord = base + __popcll(aux & ((1ull << lane) - 1ull));
      s_auxord[threadIdx.x] = has ? (int16_t)ord : (int16_t)-1;
      if (has) s_auxlist[ord] = (int16_t)threadIdx.x;
    }
    __syncthreads();
    if (!walker) {
      // ---- helper: compose the runs forwards, segments 0 .. 3 first.  The walk starts at tile 0 and a forward scan is
      // 840 vector instructions: eight of them on four SIMDs are 2.9 us of issue slots however they are spread, so the
      // walker starts with the first four segments' runs as soon as those are composed.  (With s_setprio 3 on those
      // four waves' scans the step was 0.5 us LONGER: the other four segments' scans, which the walk needs a few
      // microseconds later, were starved.)  The walker does not scan: its
      // segment -- the last the walk gets to -- is the second scan of the wave that shares its SIMD.
      // (inside the rows of 16 lanes by 1, 2, 4, 8; then every lane of rows 1 and 3 takes what lane 15 of the row before
      // has -- its run's composition up to there --, then rows 2 and 3 what lane 31 has: the same compositions as
      // doubling over the whole wave, the operands fetched by DPP)
      auto scan_forwards = [&](int seg, TileRec P, bool is_head) {
        int fp = is_head ? 1 : 0;
        auto step = [&](const TileRec &X, int xf, bool takes) {
          if (takes && !fp) {
            P = compose_rec(X, P);
            fp = xf;
          }
        };
        step(dpp_rec<0x111, 0xf>(P), dpp_int<0x111, 0xf>(fp), (lane & 15) >= 1);
        step(dpp_rec<0x112, 0xf>(P), dpp_int<0x112, 0xf>(fp), (lane & 15) >= 2);
        step(dpp_rec<0x114, 0xf>(P), dpp_int<0x114, 0xf>(fp), (lane & 15) >= 4);
        step(dpp_rec<0x118, 0xf>(P), dpp_int<0x118, 0xf>(fp), (lane & 15) >= 8);
        step(dpp_rec<0x142, 0xa>(P), dpp_int<0x142, 0xa>(fp), (lane & 16) != 0);
        step(dpp_rec<0x143, 0xc>(P), dpp_int<0x143, 0xc>(fp), lane >= 32);
        rec_put(s_pre, seg * 64 + lane, P);
        lds_fence_wave();
        if (lane == 0) __hip_atomic_fetch_add(&s_pre_cnt[seg >> 2], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      };
      // (this helper's first tile with a candidate table: the table is asked for in front of the scan and put up right
      // behind it -- the walker, on its way two scans earlier than the helpers get to their tiles, would find none up
      // at its first failures and go through the mailbox)
      const bool have_cand = !(W.selfcheck & 4);  // uniform
      uint32_t cand_out[kCand / kLanes];
#pragma unroll
      for (int j = 0; j < kCand / kLanes; j++) cand_out[j] = 0x7fc00000u;
      if (wave < naux && have_cand) {  // uniform
        const int slot = (s_rec[3][s_auxlist[wave]] >> 8) - 1;
        const uint32_t *tab = W.cand + (size_t)slot * kCand + lane;
#pragma unroll
        for (int j = 0; j < kCand / kLanes; j++) cand_out[j] = tab[j * kLanes];
      }
      scan_forwards(wave, R, head);
      if (wave < naux) {
#pragma unroll
        for (int j = 0; j < kCand / kLanes; j++) s_tab[wave][j * kLanes + lane] = cand_out[j];
        lds_fence_wave();
        if (lane == 0) lds_put(&s_tab_ord[wave], wave + 1);
      }
      // (... and its second tile's, asked for here and put up behind the backward scans: until the helpers get to their
      // tiles -- two to four scans from now -- these fourteen tables are all the walker finds)
      const bool second_early = wave + kHelpers < naux;  // uniform
      if (second_early) {
#pragma unroll
        for (int j = 0; j < kCand / kLanes; j++) cand_out[j] = 0x7fc00000u;
        if (have_cand) {
          const int slot = (s_rec[3][s_auxlist[wave + kHelpers]] >> 8) - 1;
          const uint32_t *tab = W.cand + (size_t)slot * kCand + lane;
#pragma unroll
          for (int j = 0; j < kCand / kLanes; j++) cand_out[j] = tab[j * kLanes];
        }
      }
      if (wave == kWalker - 4) {
        const TileRec Ro = rec_get(s_rec, kWalker * 64 + lane);
        const int32_t kp = lane_prev(Ro.key, Ro.key);
        const int ao = stands_alone(Ro) ? 1 : 0, ao_prev = lane_prev(ao, 0);  // (fetched by all lanes, in front of the condition)
        scan_forwards(kWalker, Ro, lane == 0 || Ro.key < 0 || Ro.key != kp || ao || ao_prev);
      }
      // ---- compose the runs backwards (wave 0: the walker's segment as well), then serve the walker
      auto scan_backwards = [&](int seg, TileRec Q, bool is_tail) {
        int fq = is_tail ? 1 : 0;
        // (inside the rows of 16 lanes by DPP, row_shl; then rows 0 and 2 take what the first lane of the row behind
        // them has, rows 0 and 1 what lane 32 has -- every partner fetched by all lanes, in front of the condition)
        auto behind = [&](const TileRec &Y, int yf, bool takes) {
          if (takes && !fq) {
            Q = compose_rec(Q, Y);
            fq = yf;
          }
        };
        behind(dpp_rec<0x101, 0xf>(Q), dpp_int<0x101, 0xf>(fq), (lane & 15) + 1 < 16);
        behind(dpp_rec<0x102, 0xf>(Q), dpp_int<0x102, 0xf>(fq), (lane & 15) + 2 < 16);
        behind(dpp_rec<0x104, 0xf>(Q), dpp_int<0x104, 0xf>(fq), (lane & 15) + 4 < 16);
        behind(dpp_rec<0x108, 0xf>(Q), dpp_int<0x108, 0xf>(fq), (lane & 15) + 8 < 16);
        {
          const int src = (lane & ~15) + 16;
          const TileRec Y = shfl_rec(Q, src);
          const int yf = __shfl(fq, src);
          behind(Y, yf, !(lane & 16));
        }
        {
          const TileRec Y = shfl_rec(Q, 32);
          const int yf = __shfl(fq, 32);
          behind(Y, yf, lane < 32);
        }
        rec_put(s_suf, seg * 64 + lane, Q);
        lds_fence_wave();
        if (lane == 0) lds_put(&s_sufok[seg], 1);
      };
      // (the helper whose wave shares the walker's SIMD -- waves go to the four SIMDs in turn -- leaves its segment
      // to another wave: the walk's first microsecond ran at half its rate next to that scan's 700 instructions)
      constexpr int kQuiet = kWalker - 4;
      if (wave != kQuiet) scan_backwards(wave, R, tail);
      if (second_early) {
#pragma unroll
        for (int j = 0; j < kCand / kLanes; j++) s_tab[wave + kHelpers][j * kLanes + lane] = cand_out[j];
        lds_fence_wave();
        if (lane == 0) lds_put(&s_tab_ord[wave + kHelpers], wave + kHelpers + 1);
      }
      if (wave == 0 || wave == 2) {
        const int seg = wave == 0 ? kWalker : kQuiet;
        const TileRec Ro = rec_get(s_rec, seg * 64 + lane);
        const int32_t kn = lane_next(Ro.key, Ro.key);
        const int ao = stands_alone(Ro) ? 1 : 0, ao_next = lane_next(ao, 0);
        scan_backwards(seg, Ro, lane == 63 || Ro.key < 0 || Ro.key != kn || ao || ao_next);
      }
      // the pair count of the iteration: a helper of row 0 adds up the tiles' counts while the walk runs (integers: no
      // order to keep); the walker stores it with its sum
      if (row == 0 && wave == 1 && last_chunk) {
        unsigned long long v = 0ull;
        for (int64_t t = lane; t < W.ntiles; t += kLanes) v += W.tile_pairs[t];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) {
          s_np = v;
          lds_put(&s_np_ok, 1);
        }
      }
      int k = wave;  // my tiles: ordinals wave, wave + 7, ...
      while (k < naux) {  // uniform
        // the tile's leaf records and terms (lane l: leaf l), in registers
        const int cur_tile = s_auxlist[k];
        const int32_t cur_key = s_rec[0][cur_tile];
        const int slot = (s_rec[3][cur_tile] >> 8) - 1;
        const int cur_kind = cur_key >= 0 ? JOB_CROSSING : JOB_NOWINDOW;
        const LeafAux A = W.aux[(size_t)slot * kLanes + lane];
        float t[kLeaf];
        load_leaf_quads(W.aux_terms + (size_t)slot * (kTile / 4), lane, t);
        // A tile without a window (a sum hovering around zero): the job kernel has carried out its 2048
        // additions from kCand start states around the guess (strict_terms.h, cand_offset); lane c holds the
        // ends of candidates c, 64 + c, 128 + c, ...
        const uint32_t g0 = (uint32_t)s_rec[1][cur_tile];
        // the table goes to LDS, where the walker looks its state up by itself (a hand-over through the mailbox is
        // 1.5-2.5 us: the helper's turn-around and two polling latencies); the mailbox is for states not in it
        if (k != wave && k + kHelpers < naux) {  // uniform: my next tile's table goes up now (this tile's went up a turn ago; my first two: above)
          const int next_slot = (s_rec[3][s_auxlist[k + kHelpers]] >> 8) - 1;
#pragma unroll
          for (int j = 0; j < kCand / kLanes; j++) cand_out[j] = 0x7fc00000u;
          if (have_cand) {  // uniform
            const uint32_t *tab = W.cand + (size_t)next_slot * kCand + lane;
#pragma unroll
            for (int j = 0; j < kCand / kLanes; j++) cand_out[j] = tab[j * kLanes];
          }
#pragma unroll
          for (int j = 0; j < kCand / kLanes; j++) s_tab[(k + kHelpers) % kTabSlots][j * kLanes + lane] = cand_out[j];
          lds_fence_wave();
          if (lane == 0) lds_put(&s_tab_ord[(k + kHelpers) % kTabSlots], k + kHelpers + 1);
        }
        // (this tile's table, for the states the walker does not find in it: the candidates either side of the state)
#pragma unroll
        for (int j = 0; j < kCand / kLanes; j++) cand_out[j] = s_tab[k % kTabSlots][j * kLanes + lane];
        bool serve = false;
        while (true) {
          if (lds_get(&s_mail[wave].req) == k + 1) {
            serve = true;
            break;
          }
          const int prog = lds_get(&s_progress);
          if (prog > cur_tile) {  // the walker got past my tile without me (or is done with the chunk)
            if (prog == 1 << 30) k = naux;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
        if (serve) {
          const long long t_begin = stat_clock(W);
          const uint32_t s_in = (uint32_t)rfl((int)__hip_atomic_load(&s_mail[wave].s_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
          int serial = 0, tried = 0, applied = 0;
          uint32_t s_out;
          // candidates are in ascending magnitude: n_le of them are <= |s_in|, so candidate n_le - 1 is the last
          // one not above the state, and (unless it IS the state) candidate n_le the first one above it; a state
          // between two candidates with the same end has that end too (x -> fl(x + t) is monotone)
          bool hit = false;
          int hit_kind = 0;
          if (have_cand && ((s_in ^ g0) >> 31) == 0u) {  // uniform
            const uint32_t m = s_in & 0x7fffffffu;
            int n_le = 0, n_lt = 0;
#pragma unroll
            for (int j = 0; j < kCand / kLanes; j++) {
              const uint32_t cm = (g0 & 0x7fffffffu) + (uint32_t)cand_offset(j * kLanes + lane);
              n_le += __popcll(__ballot(cm <= m));
              n_lt += __popcll(__ballot(cm < m));
            }
            const int below = n_le - 1, above = n_le > n_lt ? below : n_le;
            if (below >= 0 && above < kCand) {
              uint32_t a = 0u, b = 0u;
#pragma unroll
              for (int j = 0; j < kCand / kLanes; j++) {
                const uint32_t va = (uint32_t)__builtin_amdgcn_readlane((int)cand_out[j], below & 63);
                const uint32_t vb = (uint32_t)__builtin_amdgcn_readlane((int)cand_out[j], above & 63);
                a = (below >> 6) == j ? va : a;
                b = (above >> 6) == j ? vb : b;
              }
              if (a == b && (a & 0x7f800000u) != 0x7f800000u) {
                hit = true;
                s_out = a;
                hit_kind = above == below ? 25 : 26;
              }
            }
          }
          if (!hit) s_out = resolve_staged<true>(s_in, cur_kind, cur_key, A, lane, nullptr, t, serial, tried, applied);
          if (lane == 0) {
            __hip_atomic_store(&s_mail[wave].s_out, s_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            lds_put(&s_mail[wave].ack, k + 1);
            // (counters only behind the release: it waits for every memory operation of this wave before it)
            if (cur_kind == JOB_NOWINDOW) atomicAdd(&W.dbg[24], 1ull);
            if (hit) atomicAdd(&W.dbg[cur_kind == JOB_NOWINDOW ? hit_kind : 45], 1ull);
            if (!hit && have_cand) {  // measurement aid: how far off the guess was (log2 of the distance in floats)
              const uint32_t m = s_in & 0x7fffffffu, gm = g0 & 0x7fffffffu;
              const uint32_t dist = m > gm ? m - gm : gm - m;
              // (dbg[48 .. 57]; [58 .. 63] belong to the repair pass, the chunks' hand-over and the summary kernel's exchange)
              const int bucket = ((s_in ^ g0) >> 31) ? 9 : (dist == 0u ? 0 : (32 - __clz((int)dist) > 8 ? 8 : 32 - __clz((int)dist)));
              atomicAdd(&W.dbg[48 + bucket], 1ull);
            }
            resolve_stats(W, cur_kind, serial, tried, applied, stat_clock(W) - t_begin);
          }
        }
        k += kHelpers;
      }
    } else {
      // ---- the walk: one wave, every lane with the same state
      const long long t_b = stat_clock(W);
      // ---- ahead of the wait (kSpec): the walk from the states around the guess
      constexpr int kSpecPer = PCGX_SPEC_PER, kSpecStates = kSpecPer * kLanes;
      uint32_t spec_s[kSpecPer] = {};  // lane l, k: where the walk from candidate 4 l + k ends
      uint32_t spec_alive = 0u;                      // bit k: that candidate got through
      uint32_t spec_g = 0u;                          // the guess: candidate kSpecStates / 2
      bool spec_done = false;                        // uniform
      // (... where it can pay: the walk ahead takes about three real walks' time -- four candidates a lane on the vector
      // unit -- so a walker with up to three walks in front of it does better to wait for them: StrictWork::spec_depth, PCGX_STRICT_SPEC_DEPTH)
      const int walks_before = (W.ring ? W.rank : 0) * W.nchunks + my_chunk;
      if (kSpec && !kCheck && walks_before >= W.spec_depth) {  // uniform
        bool here = false;  // the start state is there already: nothing to be ahead of
        if (my_chunk > 0) {
          const unsigned long long v = __hip_atomic_load(W.chunk_state + ((size_t)row * W.nchunks + my_chunk) * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          here = (uint32_t)rfl((int)(uint32_t)(v >> 32)) == W.epoch;
        } else {
          const RingLayout RL{W.world};
          uint32_t v;
          here = ring_peek(W.ring_mine + RL.start(row), W.ring_epoch, v);
          here = rfl(here ? 1 : 0) != 0;
        }
        if (!here) {
          // the state the row has in front of this chunk, as strict_job_kernel guesses a tile's: the float64 prefix of
          // the tiles' sums and of their chains' rounding errors (+ the ranks before this one)
          const double base = tile_prefix2(W.tile_sum, W.tile_err, W.ntiles, row, chunk, lane) +
                              (W.row_base ? W.row_base[row] + W.err_base[row] : 0.0);
          spec_g = (uint32_t)rfl((int)f2u((float)base));
          if (W.selfcheck & 16) spec_g += 100000u;  // (tests, PCGX_TEST_SPEC_MISS: the state that comes is none of the candidates)
          const uint32_t gm = spec_g & 0x7fffffffu;
          if (gm > 4096u && gm < 0x7f000000u) {  // uniform (the candidates keep the guess's sign and stay finite)
            spec_alive = (1u << kSpecPer) - 1u;
#pragma unroll
            for (int k = 0; k < kSpecPer; k++)
              spec_s[k] = (spec_g & 0x80000000u) | (gm + (uint32_t)(kSpecPer * lane + k) - (uint32_t)(kSpecStates / 2));
            int seg_base[kChainSegs + 1];
            seg_base[0] = 0;
#pragma unroll
            for (int w = 0; w < kChainSegs; w++) seg_base[w + 1] = seg_base[w] + s_count[w];
            const int n_runs = rfl(seg_base[kChainSegs]), n_runs_lo = rfl(seg_base[kChainSegs / 2]);
            constexpr int kBatch = kLanes / 4;
            const bool have_cand = !(W.selfcheck & 4);
            // A job tile's candidate table: out of LDS where a helper has put it up (the first fourteen of the chunk are,
            // before anything walks), else out of global memory, a load per lane -- this walk leaves the helpers where
            // they are (they follow the REAL walk's progress word, and a real walk may still have to come).
            auto from_table = [&](int ord, int tile_in_chunk, uint32_t g0, uint32_t &st) -> bool {  // per lane
              if (!have_cand) return false;
              const int32_t idx = (int32_t)((st & 0x7fffffffu) - (g0 & 0x7fffffffu)) + kCand / 2;
              if (((st ^ g0) >> 31) != 0u || idx < 0 || idx >= kCand) return false;
              uint32_t v;
              if (lds_get(&s_tab_ord[ord % kTabSlots]) == ord + 1) {  // uniform
                v = s_tab[ord % kTabSlots][idx];
              } else {
                const int slot = (s_rec[3][tile_in_chunk] >> 8) - 1;
                v = W.cand[(size_t)slot * kCand + (size_t)idx];
              }
              if ((v & 0x7f800000u) == 0x7f800000u) return false;  // (NaN: no such candidate)
              st = v;
              return true;
            };
            bool overtaken = false;  // uniform: the state came while this walk was on its way -- the real walk is the shorter way now
            for (int r0 = 0, r1 = 0; r0 < n_runs && __ballot(spec_alive != 0u) != 0ull; r0 = r1) {  // uniform
              if (r0 > 0) {
                bool arrived;
                if (my_chunk > 0) {
                  const unsigned long long v = __hip_atomic_load(W.chunk_state + ((size_t)row * W.nchunks + my_chunk) * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  arrived = (uint32_t)rfl((int)(uint32_t)(v >> 32)) == W.epoch;
                } else {
                  const RingLayout RL{W.world};
                  uint32_t v;
                  arrived = rfl(ring_peek(W.ring_mine + RL.start(row), W.ring_epoch, v) ? 1 : 0) != 0;
                }
                if (arrived && n_runs - r0 > kBatch) {  // (with a batch or less to go the walk ahead is through sooner)
                  overtaken = true;
                  break;
                }
              }
              const int half = r0 < n_runs_lo ? 0 : 1;
              const int bound = half == 0 ? n_runs_lo : n_runs;
              r1 = r0 + kBatch < bound ? r0 + kBatch : bound;
              while (lds_get(&s_pre_cnt[half]) < kChainSegs / 2) __builtin_amdgcn_s_sleep(1);
              const int my_run = r0 + (lane >> 2), my_class = lane & 3;
              int e_l = 0, h_l = 0, j_l = -1;
              int32_t key_l = -2, cons_l = 0, c_l = 0, lo_l = kBig, hi_l = -kBig, ord_l = -1;
              uint32_t in_l = 0u, out_l = 0u;
              if (my_run < r1) {
                int idx = 0;
#pragma unroll
                for (int w = 0; w < kChainSegs; w++)
                  if (my_run >= seg_base[w] && my_run < seg_base[w + 1]) idx = w * 64 + (my_run - seg_base[w]);
                e_l = s_tail[idx];
                h_l = s_head[idx];
                key_l = s_pre[0][e_l];
                in_l = (uint32_t)s_pre[1][e_l];
                out_l = (uint32_t)s_pre[2][e_l];
                cons_l = s_pre[3][e_l];
                c_l = s_pre[4 + my_class][e_l];
                lo_l = s_pre[8 + my_class][e_l];
                hi_l = s_pre[12 + my_class][e_l];
                if (h_l == e_l) ord_l = s_auxord[e_l];
                const unsigned long long owners = ((unsigned long long)s_auxmask[h_l >> 6][1] << 32 | s_auxmask[h_l >> 6][0]) >> (h_l & 63);
                const int len = e_l - h_l + 1;
                const unsigned long long inside = len < 64 ? owners & ((1ull << len) - 1ull) : owners;
                if (inside != 0ull) j_l = h_l + __builtin_ctzll(inside);
              }
              const int n_here = rfl(r1 - r0);
              for (int j = 0; j < n_here && __ballot(spec_alive != 0u) != 0ull; j++) {  // uniform
                const int l0 = 4 * j;
                TileRec R;  // the run's composed record, the same in every lane
                R.key = __builtin_amdgcn_readlane(key_l, l0);
                R.in = (uint32_t)__builtin_amdgcn_readlane((int)in_l, l0);
                R.out = (uint32_t)__builtin_amdgcn_readlane((int)out_l, l0);
                R.cons = __builtin_amdgcn_readlane(cons_l, l0);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                  R.s.c[q] = __builtin_amdgcn_readlane(c_l, l0 + q);
                  R.s.lo[q] = __builtin_amdgcn_readlane(lo_l, l0 + q);
                  R.s.hi[q] = __builtin_amdgcn_readlane(hi_l, l0 + q);
                }
                const int e = __builtin_amdgcn_readlane(e_l, l0), h = __builtin_amdgcn_readlane(h_l, l0);
                uint32_t failed = 0u;
#pragma unroll
                for (int k = 0; k < kSpecPer; k++) {
                  if (!((spec_alive >> k) & 1u)) continue;
                  uint32_t st = spec_s[k];
                  const bool ok = (R.key >= 0 && apply(st, R.key, R.s)) || apply_point(st, R);
                  spec_s[k] = st;
                  failed |= ok ? 0u : 1u << k;
                }
                if (__ballot(failed != 0u) != 0ull) {  // uniform: the slow ways, for the candidates that need them
                  const int ord = __builtin_amdgcn_readlane(ord_l, l0), jt = __builtin_amdgcn_readlane(j_l, l0);
                  if (ord >= 0) {  // a job tile on its own: its candidate table
                    const uint32_t g0 = R.in;
#pragma unroll
                    for (int k = 0; k < kSpecPer; k++)
                      if ((failed >> k) & 1u) {
                        uint32_t st = spec_s[k];
                        if (from_table(ord, e, g0, st)) {
                          spec_s[k] = st;
                          failed &= ~(1u << k);
                        }
                      }
                  } else if (e > h && jt >= 0) {  // a run that fails at its first job tile: up to it, its table, the rest of the run
                    const int ordj = rfl((int)s_auxord[jt]);
                    TileRec P;
                    P.key = -1;
                    P.s = summary_identity();
                    if (jt > h) P = load_rec_uniform(s_pre, jt - 1);
                    const uint32_t g0 = (uint32_t)rfl(s_rec[1][jt]);
                    TileRec Sf;
                    Sf.key = -1;
                    Sf.cons = 0;
                    Sf.in = Sf.out = 0u;
                    Sf.s = summary_identity();
                    if (jt < e) {
                      while (lds_get(&s_sufok[(jt + 1) >> 6]) == 0) __builtin_amdgcn_s_sleep(1);
                      Sf = load_rec_uniform(s_suf, jt + 1);
                    }
#pragma unroll
                    for (int k = 0; k < kSpecPer; k++)
                      if ((failed >> k) & 1u) {
                        uint32_t st = spec_s[k];
                        bool ok = jt == h || (P.key >= 0 && apply(st, P.key, P.s));
                        if (ok) ok = from_table(ordj, jt, g0, st);
                        if (ok && jt < e) ok = (Sf.key >= 0 && apply(st, Sf.key, Sf.s)) || apply_point(st, Sf);
                        if (ok) {
                          spec_s[k] = st;
                          failed &= ~(1u << k);
                        }
                      }
                  }
                  spec_alive &= ~failed;  // (whatever else a run may need -- its helper, the pairs -- is the real walk's)
                }
              }
            }
            spec_done = !overtaken;
          }
        }
      }
      if (my_chunk > 0) {  // uniform
        // The state the chunk before this one ended in: ONE 64-bit word, bits | epoch, stored write-through by that
        // chunk's walker and read past the caches here (a word of an earlier launch carries an earlier epoch).  The wait
        // is for a lower block index of this launch, which is resident or next in line (above); it is bounded by
        // wall-clock time all the same (launches running side by side could hold each other's next workgroups out):
        // a walker that gives up walks the row's earlier tiles ALONE -- their records are all in memory since the
        // launch before this one -- applying what covers its state and adding the other tiles term by term from the
        // pairs: slow, and the same state (dbg[62] counts; the tests require 0).
        const unsigned long long *src = W.chunk_state + ((size_t)row * W.nchunks + my_chunk) * 16;
        bool have = false;
        long long t_first = 0;
        for (int spins = 0;; spins++) {  // uniform
          const unsigned long long v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((uint32_t)rfl((int)(uint32_t)(v >> 32)) == W.epoch) {
            s = (uint32_t)rfl((int)(uint32_t)v);
            have = true;
            break;
          }
          if ((spins & 31) == 31) {  // (a clock read is a memory round trip: now and then)
            const long long now = (long long)wall_clock64();
            if (t_first == 0) t_first = now;
            // (kCheck, the debugging build: every run of a chunk is re-derived term by term, a chunk takes 10+ ms and its
            // successors would give up one after the other -- and a row that ends by walking alone lets the launch
            // end, pose update and all, under the feet of the chunks it no longer waits for: no bound there)
            if (!kCheck && now - t_first > kChunkWaitTicks) break;
          }
          __builtin_amdgcn_s_sleep(2);
        }
        if (!have) {
          uint32_t x = row_start();
          for (int64_t t = 0; t < chunk && ring_rc == 0; t++) {  // uniform
            const TileRec V = W.recs[row * W.ntiles + t];
            TileRec T;
            T.key = rfl(V.key); T.in = (uint32_t)rfl((int)V.in); T.out = (uint32_t)rfl((int)V.out); T.cons = rfl(V.cons);
#pragma unroll
            for (int q = 0; q < 4; q++) {
              T.s.c[q] = rfl(V.s.c[q]);
              T.s.lo[q] = rfl(V.s.lo[q]);
              T.s.hi[q] = rfl(V.s.hi[q]);
            }
            if ((T.key >= 0 && apply(x, T.key, T.s)) || apply_point(x, T)) continue;
            recompute_tile_to_lds(term_src(), row, t, lane, s_tile);
            x = (uint32_t)rfl((int)serial_leaves(x, s_tile, 0, kLanes));
            __builtin_amdgcn_wave_barrier();
          }
          s = x;
          if (lane == 0) atomicAdd(&W.dbg[62], 1ull);
        }
        if ((W.selfcheck & 8) && lane == 0) s_wk[6] = (unsigned long long)(stat_clock(W) - t_b);  // (waited for the chunk before)
      } else if (W.ring && W.rank > 0) {  // uniform
        s = row_start();
        if ((W.selfcheck & 8) && lane == 0) s_wk[6] = (unsigned long long)(stat_clock(W) - t_b);  // (waited for the rank before)
      }
      // The runs of the chunk, sixteen at a time: the four lanes 4 j .. 4 j + 3 fetch run j's ends and its composed
      // record, lane 4 j + r the piece of class r.  The walk then takes a run out of the registers: its window, and --
      // once the state says which class it is in -- that class's bounds and step from lane 4 j + r: five v_readlane
      // for a run that covers the state.  (A wave issues one instruction every four cycles, scalar ones included: with
      // a whole record per lane the sixteen v_readlane and the scalar selects behind them made a run ~80
      // instructions, 0.33 us, and a row's fifteen runs 5 of the 7.5 us its walk takes.)
      // ---- (kSpec) the start state is here: one of the candidates?  Then the chunk's end state is that candidate's
      bool spec_hit = false;  // uniform
      if (kSpec && spec_done && ring_rc == 0) {
        const int32_t off = (int32_t)((s & 0x7fffffffu) - (spec_g & 0x7fffffffu)) + kSpecStates / 2;
        bool found = false;
        if (((s ^ spec_g) >> 31) == 0u && off >= 0 && off < kSpecStates) {  // uniform
          const int src = off / kSpecPer, k = off % kSpecPer;
          const uint32_t alive_src = (uint32_t)__builtin_amdgcn_readlane((int)spec_alive, src);
          uint32_t end = 0u;
#pragma unroll
          for (int q = 0; q < kSpecPer; q++) {
            const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)spec_s[q], src);
            end = q == k ? v : end;
          }
          if ((alive_src >> k) & 1u) {
            s = end;
            found = true;
          }
        }
        spec_hit = found;
        if (lane == 0) {
          atomicAdd(&W.dbg[10], 1ull);  // walks ahead of a wait, carried through
          if (found) atomicAdd(&W.dbg[11], 1ull);  // ... whose candidates held the state that came
          else if (!(((s ^ spec_g) >> 31) == 0u && off >= 0 && off < kSpecStates)) atomicAdd(&W.dbg[26], 1ull);  // ... the state lay outside them
        }
      }
      constexpr int kBatch = kLanes / 4;
      int seg_base[kChainSegs + 1];
      seg_base[0] = 0;
#pragma unroll
      for (int w = 0; w < kChainSegs; w++) seg_base[w + 1] = seg_base[w] + s_count[w];
      const int n_runs = spec_hit ? 0 : rfl(seg_base[kChainSegs]), n_runs_lo = rfl(seg_base[kChainSegs / 2]);
      __builtin_amdgcn_s_setprio(3);
      for (int r0 = 0, r1 = 0; r0 < n_runs && ring_rc == 0; r0 = r1) {
        // (a batch stays inside one half of the segments: the first half's compositions are there first)
        const int half = r0 < n_runs_lo ? 0 : 1;
        const int bound = half == 0 ? n_runs_lo : n_runs;
        r1 = r0 + kBatch < bound ? r0 + kBatch : bound;
        while (lds_get(&s_pre_cnt[half]) < kChainSegs / 2) __builtin_amdgcn_s_sleep(1);
        if ((W.selfcheck & 2) && last_chunk && r0 == 0 && lane == 0) s_wk[5] = (unsigned long long)trace_clock(W);  // the walk starts
        const long long t_f0 = stat_clock(W);
        const int my_run = r0 + (lane >> 2), my_class = lane & 3;
        int e_l = 0, h_l = 0, j_l = -1;
        int32_t key_l = -2, cons_l = 0, c_l = 0, lo_l = kBig, hi_l = -kBig, ord_l = -1;
        uint32_t in_l = 0u, out_l = 0u;
        if (my_run < r1) {
          int idx = 0;
#pragma unroll
          for (int w = 0; w < kChainSegs; w++)
            if (my_run >= seg_base[w] && my_run < seg_base[w + 1]) idx = w * 64 + (my_run - seg_base[w]);
          e_l = s_tail[idx];
          h_l = s_head[idx];
          key_l = s_pre[0][e_l];
          in_l = (uint32_t)s_pre[1][e_l];
          out_l = (uint32_t)s_pre[2][e_l];
          cons_l = s_pre[3][e_l];
          c_l = s_pre[4 + my_class][e_l];
          lo_l = s_pre[8 + my_class][e_l];
          hi_l = s_pre[12 + my_class][e_l];
          // a run of ONE tile that owns a slot (a tile without a window always is one): its place among the job tiles
          if (h_l == e_l) ord_l = s_auxord[e_l];
          // a longer run: its first tile that owns a slot (-1: none) -- where such a run fails, nearly always
          const unsigned long long owners = ((unsigned long long)s_auxmask[h_l >> 6][1] << 32 | s_auxmask[h_l >> 6][0]) >> (h_l & 63);
          const int len = e_l - h_l + 1;
          const unsigned long long inside = len < 64 ? owners & ((1ull << len) - 1ull) : owners;
          if (inside != 0ull) j_l = h_l + __builtin_ctzll(inside);
        }
        const int n_here = rfl(r1 - r0);  // (a scalar for the compiler: the loops below are uniform)
        // (the records are in their registers HERE: left to the compiler, the wait for them sits at the top of the
        // loop below, where every turn it also waits for the turn before's store of the progress word)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(key_l), "+v"(in_l), "+v"(out_l), "+v"(cons_l), "+v"(c_l), "+v"(lo_l), "+v"(hi_l), "+v"(e_l), "+v"(h_l),
                       "+v"(ord_l), "+v"(j_l));
        if ((W.selfcheck & 8)) {
          const unsigned long long tt = clock_after((uint32_t)(key_l ^ hi_l ^ e_l));
          if (lane == 0) s_wk[4] += tt - (unsigned long long)t_f0;
        }
        const long long t_j0 = stat_clock(W);
        // The runs that cover the state -- nearly all -- in a loop of their own: window, class, bounds, step, the
        // progress word; the first one that does not leaves it for the slow way below (its point record, its table,
        // its tiles one by one) and the loop is entered again behind it.  (As one loop the compiler's block layout
        // put three taken branches and a dozen register copies for the slow way's sake into every run: 75
        // instructions, 0.3 us.)
        auto run_covers = [&](int l0, uint32_t &state) -> bool {
          // apply() (strict_sum.h: state_to_n, the class's bounds, n_to_state) without a branch before the verdict
          const int32_t key = __builtin_amdgcn_readlane(key_l, l0);
          const int32_t sg = (int32_t)(state >> 31), E = (int32_t)((state >> 23) & 0xffu), e = key & 0xff;
          const int32_t mant = (int32_t)((state & 0x7fffffu) | 0x800000u);
          const int32_t n = E == e ? mant << 1 : mant;
          const int in_window = (int)(key >= 0) & (int)(sg == (key >> 8)) & ((int)(E == e) | (int)(E == e - 1));
          const int lc = l0 | (n & 3);
          const int32_t lo = __builtin_amdgcn_readlane(lo_l, lc), hi = __builtin_amdgcn_readlane(hi_l, lc);
          const int32_t m = n + __builtin_amdgcn_readlane(c_l, lc);
          const uint32_t sgb = (uint32_t)sg << 31;
          const uint32_t below = sgb | ((uint32_t)(e - 1) << 23) | ((uint32_t)m & 0x7fffffu);
          const uint32_t above = sgb | ((uint32_t)e << 23) | (((uint32_t)m >> 1) & 0x7fffffu);
          const bool ok = (in_window & (int)(n >= lo) & (int)(n <= hi)) != 0;
          state = ok ? (m < N24 ? below : above) : state;
          return ok;
        };
        int j = 0;
        while (true) {  // uniform
          if (!kCheck) {
            for (; j < n_here; j++) {
              if (!run_covers(4 * j, s)) break;
              n_run++;
              // (a hint for helpers whose tile the walk has passed: no ordering needed; every lane stores the same word)
              __hip_atomic_store(&s_progress, __builtin_amdgcn_readlane(e_l, 4 * j) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
          }
          if (j >= n_here) break;
          const int l0 = 4 * j;
          j++;
          n_run++;
          {
            const uint32_t s_in = s;
            const int32_t key = __builtin_amdgcn_readlane(key_l, l0);
            bool ok = kCheck ? run_covers(l0, s) : false;  // (kCheck: every run comes this way, for its term-by-term check)
            if (!ok && (__builtin_amdgcn_readlane(cons_l, l0) & 1) && (uint32_t)__builtin_amdgcn_readlane((int)in_l, l0) == s) {
              s = (uint32_t)__builtin_amdgcn_readlane((int)out_l, l0);  // apply_point()
              ok = true;
            }
            if (!ok) {
              // A job tile on its own whose candidate table is up (its helper holds it): the look-up HERE, not at the
              // end of the failure path below -- a row whose sum hovers around zero is a string of such tiles (none of
              // them has a window: their records never apply), and the way through "which tile of the run? its
              // record? its helper?" costs each 1-2 us of LDS round trips.
              const int ord = __builtin_amdgcn_readlane(ord_l, l0);
              if (ord >= 0 && lds_get(&s_tab_ord[ord % kTabSlots]) == ord + 1) {
                const uint32_t g0 = (uint32_t)__builtin_amdgcn_readlane((int)in_l, l0);
                const int32_t idx = (int32_t)((s & 0x7fffffffu) - (g0 & 0x7fffffffu)) + kCand / 2;
                if (((s ^ g0) >> 31) == 0u && idx >= 0 && idx < kCand) {
                  const uint32_t v = (uint32_t)rfl((int)s_tab[ord % kTabSlots][idx]);
                  if ((v & 0x7f800000u) != 0x7f800000u) {  // (NaN: no such candidate)
                    s = v;
                    ok = true;
                    if (key >= 0) n_tab_cross++;
                    else n_tab_nw++;
                  }
                }
              }
            }
            if (ok) {
              selfcheck<kCheck>(W, term_src, row, s_in, s, chunk + __builtin_amdgcn_readlane(h_l, l0),
                                chunk + __builtin_amdgcn_readlane(e_l, l0) + 1, 0, lane, s_tile);
              __hip_atomic_store(&s_progress, __builtin_amdgcn_readlane(e_l, l0) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              continue;
            }
          }
          const int e = __builtin_amdgcn_readlane(e_l, l0), h = __builtin_amdgcn_readlane(h_l, l0);
          n_runfail++;
          const long long t_fail = stat_clock(W);
          // Where a run of several tiles fails it is nearly always at a tile that owns a slot (a level crossing whose
          // record covers the states next to its guess only).  The run's first such tile j straight away: the
          // composition h .. j - 1 (the forward scan's), j's candidate table, the composition j + 1 .. e (the backward
          // scan's) -- three look-ups instead of the search below for the tile that fails, that tile's own record,
          // its table and the rest.  Any of the three not covering the state: the search below, from the run's head.
          if (!kCheck && e > h) {
            const int jt = __builtin_amdgcn_readlane(j_l, l0);
            if (jt >= 0) {
              const int ord = rfl((int)s_auxord[jt]);
              uint32_t st = s;
              bool okj = true;
              if (jt > h) {
                const TileRec P = load_rec_uniform(s_pre, jt - 1);
                okj = P.key >= 0 && apply(st, P.key, P.s);
              }
              if (okj) {
                okj = false;
                if (lds_get(&s_tab_ord[ord % kTabSlots]) == ord + 1) {  // the tile's candidate table is up
                  const uint32_t g0 = (uint32_t)rfl(s_rec[1][jt]);
                  const int32_t idx = (int32_t)((st & 0x7fffffffu) - (g0 & 0x7fffffffu)) + kCand / 2;
                  if (((st ^ g0) >> 31) == 0u && idx >= 0 && idx < kCand) {
                    const uint32_t v = (uint32_t)rfl((int)s_tab[ord % kTabSlots][idx]);
                    if ((v & 0x7f800000u) != 0x7f800000u) {  // (NaN: no such candidate)
                      st = v;
                      okj = true;
                    }
                  }
                }
              }
              if (okj && jt < e) {
                while (lds_get(&s_sufok[(jt + 1) >> 6]) == 0) {
                }
                const TileRec Sf = load_rec_uniform(s_suf, jt + 1);
                okj = (Sf.key >= 0 && apply(st, Sf.key, Sf.s)) || apply_point(st, Sf);
              }
              if (okj) {
                s = st;
                n_recfail++;
                n_tab_cross++;
                __hip_atomic_store(&s_progress, e + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                continue;
              }
            }
          }
          // which tile?  lane j tries the composition h .. h + j
          int f = h;
          if (e > h) {
            // (sixty-four tiles of the run at a time, every composition from the run's head and the state there: a run
            // longer than that used to be walked tile by tile behind the sixty-fourth, 0.5 us each)
            const uint32_t s_in = s;
            for (int base = h; base <= e; base += kLanes) {  // uniform
              uint32_t mine = s_in;
              bool ok = false;
              if (base + lane <= e) {
                const TileRec Pj = rec_get(s_pre, base + lane);
                ok = Pj.key >= 0 && apply(mine, Pj.key, Pj.s);
              }
              const unsigned long long good = __ballot(ok);
              const int ngood = good == ~0ull ? kLanes : __builtin_ctzll(~good);  // tiles base .. base + ngood - 1 are covered
              if (ngood > 0) s = (uint32_t)__builtin_amdgcn_readlane((int)mine, ngood - 1);
              f = base + ngood;
              if (ngood < kLanes) break;
            }
            selfcheck<kCheck>(W, term_src, row, s_in, s, chunk + h, chunk + f, 1, lane, s_tile);
          }
          while (f <= e) {
            // tile f does not cover the state (its place in the run's prefix composition failed, or it is
            // the run's head); its point record may still fit
            const uint32_t s_in = s;
            const TileRec T = load_rec_uniform(s_rec, f);
            if (!(f > h && ((T.key >= 0 && apply(s, T.key, T.s)) || apply_point(s, T)))) {
              n_recfail++;
              const int ord = s_auxord[f];
              if (lane == 0) lds_put(&s_progress, f);
              bool from_table = false;
              int why = 0;
              if (ord >= 0 && lds_get(&s_tab_ord[ord % kTabSlots]) == ord + 1) {  // the tile's candidate table is up
                const uint32_t g0 = (uint32_t)s_rec[1][f];
                const int32_t idx = (int32_t)((s & 0x7fffffffu) - (g0 & 0x7fffffffu)) + kCand / 2;
                why = 8;
                if (((s ^ g0) >> 31) == 0u && idx >= 0 && idx < kCand) {
                  why = 16;
                  // (rfl: the state stays a scalar for the compiler -- one VGPR source and every run's apply() moves
                  // from the scalar unit to the vector unit's dependent-instruction latency)
                  const uint32_t v = (uint32_t)rfl((int)s_tab[ord % kTabSlots][idx]);
                  if ((v & 0x7f800000u) != 0x7f800000u) {  // (NaN: no such candidate)
                    s = v;
                    from_table = true;
                    if (T.key >= 0) n_tab_cross++;
                    else n_tab_nw++;
                  }
                }
              }
              if (from_table) {
              } else if (ord >= 0) {  // its helper has the tile's terms and leaf records at hand
                n_mail += 1u << why;
                ChainMail *M = &s_mail[ord % kHelpers];
                if (lane == 0) {
                  __hip_atomic_store(&M->s_in, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                  lds_put(&M->req, ord + 1);
                }
                const long long t_m = stat_clock(W);
                while (lds_get(&M->ack) != ord + 1) {
                }
                if ((W.selfcheck & 8) && lane == 0) { s_wk[1] += (unsigned long long)(stat_clock(W) - t_m); s_wk[2] += 1ull; }
                s = (uint32_t)rfl((int)__hip_atomic_load(&M->s_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
              } else {  // a tile without a slot: its terms are formed again from the pairs, all 2048 are added
                const long long t_begin = stat_clock(W);
                recompute_tile_to_lds(term_src(), row, chunk + f, lane, s_tile);
                LeafAux none;
                int serial, tried, applied;
                s = resolve_staged<false>(s, 0, -1, none, lane, s_tile, nullptr, serial, tried, applied);
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) resolve_stats(W, 0, serial, tried, applied, stat_clock(W) - t_begin);
              }
            }
            selfcheck<kCheck>(W, term_src, row, s_in, s, chunk + f, chunk + f + 1, 2, lane, s_tile);
            f++;
            if (f > e) break;
            const long long t_s = stat_clock(W);
            while (lds_get(&s_sufok[f >> 6]) == 0) {
            }
            if ((W.selfcheck & 8) && lane == 0) s_wk[3] += (unsigned long long)(stat_clock(W) - t_s);
            const TileRec Sf = load_rec_uniform(s_suf, f);  // the rest of the run in one step
            const uint32_t s_in2 = s;
            if ((Sf.key >= 0 && apply(s, Sf.key, Sf.s)) || apply_point(s, Sf)) {
              selfcheck<kCheck>(W, term_src, row, s_in2, s, chunk + f, chunk + e + 1, 3, lane, s_tile);
              break;
            }
          }
          if (lane == 0) lds_put(&s_progress, e + 1);
          if ((W.selfcheck & 8) && lane == 0) s_wk[0] += (unsigned long long)(stat_clock(W) - t_fail);
        }
        if (W.selfcheck & 8) {
          const unsigned long long tt = clock_after(s);
          if (lane == 0) s_wk[5] += tt - (unsigned long long)t_j0;
        }
      }
      if (!last_chunk && lane == 0 && ring_rc == 0)  // the next chunk's walker starts here
        __hip_atomic_store(W.chunk_state + ((size_t)row * W.nchunks + my_chunk + 1) * 16, (unsigned long long)W.epoch << 32 | s,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // (a broken ring: a later chunk's walker finds it broken too, in its own wait for the row's first state)
      if (last_chunk && W.ring && ring_rc == 0 && lane == 0) {
        // the ring form: this rank's end state to the rank behind it -- or, from the last rank, the sum itself to all
        const RingLayout RL{W.world};
        if (W.rank < W.world - 1) ring_put(ring_inbox(W, W.rank + 1) + RL.start(row), s, W.ring_epoch);
        else
          for (int k = 0; k < W.world - 1; k++) ring_put(ring_inbox(W, k) + RL.final(row), s, W.ring_epoch);
      }
      if (lane == 0) {
        lds_put(&s_progress, 1 << 30);  // helpers that still wait for a tile: the chunk is done
        ticks_scan += (unsigned long long)(t_b - t_a);
        ticks_walk += (unsigned long long)(stat_clock(W) - t_b);
        // (the row's counters so far: a helper adds them to the launch's behind the last chunk -- a dozen atomics
        // that the walker's own release, in front of its ticket, would wait for)
        s_stat[0] = n_run; s_stat[1] = n_runfail; s_stat[2] = n_recfail; s_stat[3] = n_tab_nw; s_stat[4] = n_tab_cross; s_stat[5] = n_mail;
        s_stat_ticks[0] = ticks_scan; s_stat_ticks[1] = ticks_walk;
      }
    }
    __syncthreads();
  }
  if (wave == 0 && lane == 0) {
    const unsigned long long scan_ticks = s_stat_ticks[0], walk_ticks = s_stat_ticks[1];
    atomicAdd(&W.dbg[0], (unsigned long long)s_stat[0]);
    atomicAdd(&W.dbg[1], (unsigned long long)s_stat[1]);
    atomicAdd(&W.dbg[4], (unsigned long long)s_stat[2]);
    atomicAdd(&W.dbg[8], scan_ticks);
    atomicAdd(&W.dbg[9], walk_ticks);
    atomicMax(&W.dbg[14 + 32], walk_ticks);   // slowest row of the launch
    atomicAdd(&W.dbg[27 + row], walk_ticks);  // per row: walk ticks, tiles resolved
    if (W.selfcheck & 8) {  // (the rows' own counts of tiles resolved give way to the walkers' tick columns)
      for (int k = 0; k < 6; k++) atomicAdd(&W.dbg[36 + k], s_wk[k]);
      atomicAdd(&W.dbg[60], s_wk[6]);  // walkers of later chunks: waiting for their start state (part of the walk ticks)
      if (last_chunk) atomicAdd(&W.dbg[61], s_wk[6]);  // ... the rows' last chunks alone: the row's walk up to there
    } else {
      atomicAdd(&W.dbg[36 + row], (unsigned long long)s_stat[2]);
    }
    atomicAdd(&W.dbg[24], (unsigned long long)s_stat[3]);  // tiles without a window seen, and found in their candidate tables
    atomicAdd(&W.dbg[25], (unsigned long long)s_stat[3]);
    atomicAdd(&W.dbg[45], (unsigned long long)s_stat[4]);
    if (W.selfcheck & 8) {  // (PCGX_STRICT_CLOCKS: why tiles went to their helpers)
      atomicAdd(&W.dbg[42], (unsigned long long)(s_stat[5] & 0xffu));
      atomicAdd(&W.dbg[43], (unsigned long long)((s_stat[5] >> 8) & 0xffu));
      atomicAdd(&W.dbg[44], (unsigned long long)((s_stat[5] >> 16) & 0xffu));
    }
  }
  if (!last_chunk) return;  // uniform
  if (walker && lane == 0 && (W.selfcheck & 2)) {  // (row r's stamps ride in tile r's line: words 11 .. 15)
    W.stamps[row * 16 + 11] = (unsigned long long)t_enter;
    W.stamps[row * 16 + 10] = (unsigned long long)__builtin_readcyclecounter() - c_enter;  // shader clocks in the kernel
    W.stamps[row * 16 + 12] = s_wk[4];                                   // the row's records have arrived
    W.stamps[row * 16 + 13] = s_wk[5];                                   // the walk starts
    W.stamps[row * 16 + 14] = (unsigned long long)n_run | ((unsigned long long)n_runfail << 16) |
                              ((unsigned long long)n_recfail << 32) | ((unsigned long long)(n_tab_nw + n_tab_cross) << 48);
    W.stamps[row * 16 + 15] = (unsigned long long)trace_clock(W);
  }
  if (W.ring && walker) {  // uniform
    // The ring form: the pairs go round with sum 0 (every rank adds its own count to what it was sent); then every
    // rank's walkers wait for the LAST rank's end states -- the sums of the whole target -- and go on as on one GPU:
    // sums10, ticket, evaluate tail + pose update on every rank alike.
    const RingLayout RL{W.world};
    unsigned long long pairs = 0ull;
    if (row == 0) {
      while (lds_get(&s_np_ok) == 0) __builtin_amdgcn_s_sleep(1);
      pairs = s_np;
      if (W.rank > 0 && ring_rc == 0) {
        uint32_t lo = 0u, hi = 0u;
        ring_rc = ring_wait(W, RL.start_pairs(), lo, kRingWalkTicks);
        if (ring_rc == 0) ring_rc = ring_wait(W, RL.start_pairs() + 1, hi, kRingWalkTicks);
        pairs += (unsigned long long)hi << 32 | lo;
      }
      if (ring_rc == 0 && lane == 0) {
        if (W.rank < W.world - 1) {
          unsigned long long *dst = ring_inbox(W, W.rank + 1) + RL.start_pairs();
          ring_put(dst, (uint32_t)pairs, W.ring_epoch);
          ring_put(dst + 1, (uint32_t)(pairs >> 32), W.ring_epoch);
        } else {
          for (int k = 0; k < W.world - 1; k++) {
            unsigned long long *dst = ring_inbox(W, k) + RL.final_pairs();
            ring_put(dst, (uint32_t)pairs, W.ring_epoch);
            ring_put(dst + 1, (uint32_t)(pairs >> 32), W.ring_epoch);
          }
        }
      }
    }
    if (W.rank < W.world - 1 && ring_rc == 0) {
      uint32_t v = 0u;
      ring_rc = ring_wait(W, RL.final(row), v, kRingWalkTicks);
      s = (uint32_t)rfl((int)v);
      if (row == 0 && ring_rc == 0) {
        uint32_t lo = 0u, hi = 0u;
        ring_rc = ring_wait(W, RL.final_pairs(), lo, kRingWalkTicks);
        if (ring_rc == 0) ring_rc = ring_wait(W, RL.final_pairs() + 1, hi, kRingWalkTicks);
        pairs = (unsigned long long)hi << 32 | lo;
      }
    }
    if (ring_rc != 0) {  // uniform: the ring is broken -- this rank's Fit ends here, like everybody's
      if (lane == 0) {
        if (ring_rc == 2) ring_raise_abort(W, 0x200u | (uint32_t)row);
        state->status = PCGX_E_RCCL;
        state->done = 1;
      }
      return;
    }
    if (row == 0 && lane == 0) __hip_atomic_store(&sums10[S_PAIRS], (double)pairs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else if (row == 0 && walker && lane == 0) {
    while (lds_get(&s_np_ok) == 0) __builtin_amdgcn_s_sleep(1);
    if (!W.hop_out) __hip_atomic_store(&sums10[S_PAIRS], (double)s_np, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // component order of sums10: Value, G0..G5, DistRMS, Weight, Pairs
  if (walker) {
    unsigned ticket = 0u;
    if (lane == 0) {
      const int slot = row == 0 ? S_VALUE : (row <= 6 ? S_G0 + row - 1 : (row == 7 ? S_DIST_RMS : S_WEIGHT));
      if (W.hop_out) W.hop_out[row] = (double)s;  // the state's BITS, as a number: the next rank's walk starts there
      else __hip_atomic_store(&sums10[slot], (double)u2f(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // the row that finishes last has all nine sums: evaluate tail + pose update (evaluator.go:156-186,
      // updater.go:44-71) in the same launch
      // (the sum went out write-through, and the last row reads the sums past the caches: what the ticket has to wait
      // for is that one store's arrival, not a write-back of everything the workgroup has written -- a release fence
      // here and an acquire fence behind the ticket were 0.6 us of the launch's tail)
      if (W.hop_out) __threadfence();
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      ticket = atomicAdd(W.done_rows, 1u);
    }
    ticket = (unsigned)rfl((int)ticket);  // (lane 0's, in every lane: the launch's last steps are the whole wave's)
    if (ticket == (unsigned)W.nrows - 1u) {  // uniform
      // the counters of this iteration's exchange and job slots, for the next one: a word per lane (one lane took
      // eighty stores, one after the other, between the ticket and the pose update)
      static_assert(kAuxShards == kLanes, "a shard's counter per lane");
      W.aux_count[lane * 32] = 0u;
      if (W.exchange)
        for (int64_t k = lane, n = (W.ntiles + 31) / 32, m = n + (n + 31) / 32; k < m; k += kLanes) W.tile_arrived[32 * k] = 0u;
      if (lane == 0) *W.done_rows = 0u;
      if (!W.hop_out && lane == 0) {  // (sharded: the sums are put together behind the last rank's walk)
        double sums[S_COUNT];
        for (int k = 0; k < S_COUNT; k++) sums[k] = __hip_atomic_load(&sums10[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long np = (unsigned long long)sums[S_PAIRS];  // row 0 stored it before its ticket
        if (W.nrows < kStrictRows) {
          // default weight: every term of the ninth sum is 1.0f, and 0 + 1 + 1 + ... in float32 is the pair
          // count up to 2^24, where it stays (2^24 + 1 rounds back to 2^24): no chain to evaluate
          sums[S_WEIGHT] = (double)(np < (1ull << 24) ? np : (1ull << 24));
          sums10[S_WEIGHT] = sums[S_WEIGHT];
        }
        const long long t_u0 = trace_clock(W);
        if (fuse_update) icp_update_step(state, sums, kp);
        if (W.selfcheck & 2) {  // (the launch's last steps, in the line behind the rows')
          __threadfence();
          W.stamps[kStrictRows * 16 + 14] = (unsigned long long)t_u0;
          W.stamps[kStrictRows * 16 + 15] = (unsigned long long)trace_clock(W);
        }
      }
    }
  }
}

// ---- host side ---------------------------------------------------------------------------------------
struct StrictBuffers {
  StrictWork w;
  void *block = nullptr;
  // a target spread over ranks (strict_enqueue_sharded): [world][16] slots of the all-gathers, then row_base[16],
  // err_base[16] ([9] = pairs of all ranks), hop[16], start_bits[16] (uint32)
  double *shard = nullptr;
  int shard_world = 0;
  bool shard_ring = false;  // `shard` was last laid out (and zeroed) for the ring form
  void *counters = nullptr;  // slot / ticket counters (strict_reset)
  size_t counters_bytes = 0, arrived_bytes = 0;
};

pcgx_status strict_create(int64_t nt, const float *tx, const float *ty, const float *tz, const uint32_t *pos_of,
                          StrictBuffers **out, hipStream_t st) {
  StrictBuffers *b = new StrictBuffers();
  StrictWork &W = b->w;
  W.nt = nt;
  W.raw_terms = nullptr;
  W.row_base = W.err_base = nullptr;
  W.start_bits = nullptr;
  W.hop_out = nullptr;
  W.first_exact = 1;
  W.ring = nullptr;
  W.ring_words = 0;
  W.ring_tab = nullptr;
  W.ring_mine = nullptr;
  W.ring_guess_ticks = kRingGuessTicks;
  W.rank = 0;
  W.world = 1;
  W.ring_epoch = 0u;
  W.ring_base = nullptr;
  W.ring_flag = nullptr;
  W.ntiles = nt > 0 ? (nt + kTile - 1) / kTile : 1;
  W.nrows = kStrictRows;
  W.spec_depth = getenv("PCGX_STRICT_SPEC_DEPTH") ? atoi(getenv("PCGX_STRICT_SPEC_DEPTH")) : 4;
  W.selfcheck = (getenv("PCGX_STRICT_SELFCHECK") ? 1 : 0) | (getenv("PCGX_STRICT_TRACE") ? 2 : 0) |
                (getenv("PCGX_STRICT_NOSPEC") ? 4 : 0) | (getenv("PCGX_STRICT_CLOCKS") ? 8 : 0) | (getenv("PCGX_TEST_SPEC_MISS") ? 16 : 0);
  // (4: the chain kernel ignores the candidate tables of tiles without a window; 8: tick columns of the debug counters)
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t sz_tile = up((size_t)kStrictRows * W.ntiles * sizeof(double));  // (twice: sums and errors)
  const size_t sz_pairs = up((size_t)W.ntiles * sizeof(uint32_t));
  W.ntiles_pad = (W.ntiles + 127) & ~(int64_t)127;
  const size_t sz_pub = up((size_t)W.ntiles_pad * 16 * sizeof(double));
  const int64_t n_groups = (W.ntiles + 31) / 32;
  const size_t sz_arr = up((size_t)(n_groups + (n_groups + 31) / 32) * 128);
  // (PCGX_STRICT_EXCHANGE=0: the tile sums are formed by a pass of their own in front of the summaries, as in round 3)
  W.exchange = (getenv("PCGX_STRICT_EXCHANGE") && atoi(getenv("PCGX_STRICT_EXCHANGE")) == 0) ? 0 : 1;
  const size_t sz_rec = up((size_t)kStrictRows * W.ntiles * sizeof(TileRec));
  // slots for the tiles that cross a level or have no window (6 KB of leaf records + 8 KB of terms each):
  // a quarter of all tiles, far more than ever seen (C4: ~2 %; a sum hovering around zero over the whole
  // row: ~15 %); a tile that finds none left is recomputed from the pairs by the chain kernel
  W.naux = (int32_t)(kAuxShards * ((kStrictRows * W.ntiles / 4 + kAuxShards - 1) / kAuxShards + 4));
  if (const char *e = getenv("PCGX_STRICT_SLOTS_PER_SHARD")) {  // tests: run out of slots (tiles then take the chain kernel's
    const int per = atoi(e);                                    // recompute-from-the-pairs path)
    if (per >= 0 && per * kAuxShards < W.naux) W.naux = per * kAuxShards;
  }
  const size_t sz_aux = up((size_t)W.naux * kLanes * sizeof(LeafAux));
  const size_t sz_auxt = up((size_t)W.naux * kTile * sizeof(float));
  const size_t sz_jobs = up((size_t)W.naux * sizeof(JobDesc));
  const size_t sz_cand = up((size_t)W.naux * kCand * sizeof(uint32_t));
  const size_t sz_xyz = up((size_t)(nt ? nt : 1) * 12 + 64);
  const size_t sz_stamps = up((size_t)W.ntiles * 16 * sizeof(unsigned long long));
  const size_t sz_ctr = 256 + (size_t)kAuxShards * 128;
  W.nchunks = (int32_t)((W.ntiles + kChainTiles - 1) / kChainTiles);
  W.epoch = 0u;
  const size_t sz_chunk = up((size_t)kStrictRows * W.nchunks * 16 * sizeof(unsigned long long));
  const size_t total = 2 * sz_tile + sz_pub + sz_arr + sz_pairs + sz_rec + sz_aux + sz_auxt + sz_jobs + sz_cand + sz_xyz + sz_ctr + 512 + sz_stamps + sz_chunk;
  hipError_t e = dev_cache_alloc(&b->block, total);
  if (e != hipSuccess) {
    delete b;
    return fail(PCGX_E_OOM, "strict sums: allocation of %zu bytes failed: %s", total, hipGetErrorString(e));
  }
  uint8_t *p = (uint8_t *)b->block;
  W.tile_sum = (double *)p; p += sz_tile;
  W.tile_err = (double *)p; p += sz_tile;
  W.tile_pub = (double *)p; p += sz_pub;  // (256-byte aligned: a tile's line is one 128-byte line)
  W.tile_arrived = (unsigned int *)p; p += sz_arr;
  W.tile_pairs = (uint32_t *)p; p += sz_pairs;
  W.recs = (TileRec *)p; p += sz_rec;
  W.aux = (LeafAux *)p; p += sz_aux;
  W.aux_terms = (float4 *)p; p += sz_auxt;
  W.jobs = (JobDesc *)p; p += sz_jobs;
  W.cand = (uint32_t *)p; p += sz_cand;
  W.xyz_caller = (const float *)p; p += sz_xyz;
  uint8_t *counters = p;
  b->counters = counters;
  b->counters_bytes = sz_ctr;
  b->arrived_bytes = sz_arr;
  W.done_rows = (unsigned int *)(p + 12);
  W.aux_count = (unsigned int *)(p + 256); p += sz_ctr;
  W.dbg = (unsigned long long *)p; p += 512;
  W.stamps = (unsigned long long *)p; p += sz_stamps;
  W.chunk_state = (unsigned long long *)p;
  // slot / ticket counters and debug counters start at zero (the chain kernel re-zeroes what it consumed)
  e = hipMemsetAsync(counters, 0, sz_ctr + 512, st);
  if (e == hipSuccess) e = hipMemsetAsync(W.tile_arrived, 0, sz_arr, st);
  if (e == hipSuccess) e = hipMemsetAsync(W.chunk_state, 0, sz_chunk, st);  // (epoch 0: no launch's)
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
  dev_cache_free(b->shard);
  dev_cache_free(b->block);
  delete b;
}

const StrictWork *strict_work(StrictBuffers *b, const IcpKernelParams &kp) {
  b->w.weight_fn = kp.weight_fn;
  b->w.weight_a = kp.weight_a;
  // (the row count may only change between iterations: the chain kernel's ticket counts to it)
  b->w.nrows = kp.weight_fn == PCGX_WEIGHT_ONE ? kStrictRows - 1 : kStrictRows;
  return &b->w;
}
// (PCGX_STRICT_REPAIR=0: no repair pass in a Fit's first Evaluate -- measurement)
static bool repair_enabled() {
  static const bool on = !(getenv("PCGX_STRICT_REPAIR") && atoi(getenv("PCGX_STRICT_REPAIR")) == 0);
  return on;
}
// a launch of the chain kernel: its epoch (the chunks' hand-over words, StrictWork::chunk_state)
static const StrictWork &next_epoch(StrictBuffers *b) {
  if (++b->w.epoch == 0u) b->w.epoch = 1u;  // (a wrap after 2^32 launches: stale words are 2^32 launches old by then)
  return b->w;
}

// (PCGX_STRICT_SPEC=0: no walk ahead of a wait -- measurement)
static bool spec_enabled() {
  static const bool on = !(getenv("PCGX_STRICT_SPEC") && atoi(getenv("PCGX_STRICT_SPEC")) == 0);
  return on;
}
// the chain kernel of a step; spec: some walker of the launch waits for its start state (several chunks, a rank behind
// another) -- the instantiation whose walkers walk ahead of that wait.  One chunk on one GPU: the kernel without that code.
static void launch_chain(const StrictWork &W, const float4 *match, const uint32_t *pos_of, IcpState *state, double *sums10,
                         const IcpKernelParams &kp, int fuse_update, bool waits, hipStream_t st) {
  const dim3 grid((unsigned)(W.nrows * W.nchunks)), block(kChainBlock);
  if (W.selfcheck & 1)
    hipLaunchKernelGGL((strict_chain_kernel<true, false>), grid, block, 0, st, match, pos_of, state, W, sums10, kp, fuse_update);
  else if (waits && spec_enabled() && (W.ring ? W.rank : 0) * W.nchunks + W.nchunks - 1 >= W.spec_depth)  // (some walker of THIS launch walks ahead:
    // the instantiation with that code spills thirty scalar registers more, which the real walk pays for -- 0.7 us a step)
    hipLaunchKernelGGL((strict_chain_kernel<false, true>), grid, block, 0, st, match, pos_of, state, W, sums10, kp, fuse_update);
  else
    hipLaunchKernelGGL((strict_chain_kernel<false, false>), grid, block, 0, st, match, pos_of, state, W, sums10, kp, fuse_update);
}

pcgx_status strict_enqueue(StrictBuffers *b, const float4 *match, const uint32_t *pos_of, IcpState *state,
                           double *sums10, const IcpKernelParams &kp, bool fuse_update, bool have_tile_sums, bool first_iter,
                           hipStream_t st) {
  (void)strict_work(b, kp);
  const StrictWork &W = next_epoch(b);
  if (!have_tile_sums && !W.exchange) {
    ProfScope prof(PCGX_PROF_STRICT_TERMS, st);
    hipLaunchKernelGGL(strict_tilesum_kernel, dim3((unsigned)W.ntiles), dim3(kTileSumBlock), 0, st, match, pos_of,
                       (const IcpState *)state, W);
  }
  {
    ProfScope prof(PCGX_PROF_STRICT_SUM, st);
    if (W.exchange)
      hipLaunchKernelGGL(strict_sum_kernel<true>, dim3((unsigned)W.ntiles), dim3(kSumBlock), 0, st, match, pos_of,
                         (const IcpState *)state, W);
    else
      hipLaunchKernelGGL(strict_sum_kernel<false>, dim3((unsigned)W.ntiles), dim3(kSumBlock), 0, st, match, pos_of,
                         (const IcpState *)state, W);
  }
  {
    ProfScope prof(PCGX_PROF_STRICT_JOB, st);
    // (the first Evaluate of a Fit: the plain tiles the rows' drift has carried across a binade's end become jobs)
    if (first_iter && W.naux > 0 && W.ntiles >= kRepairMinTiles && repair_enabled())
      hipLaunchKernelGGL(strict_repair_kernel, dim3((unsigned)(W.nrows * ((W.ntiles + kRepairBlock - 1) / kRepairBlock))), dim3(kRepairBlock), 0,
                         st, match, pos_of, (const IcpState *)state, W);
    if (W.naux > 0)
      hipLaunchKernelGGL(strict_job_kernel, dim3((unsigned)kJobRoles * (unsigned)W.naux), dim3(kJobBlock), 0, st, (const IcpState *)state, W);
  }
  {
    ProfScope prof(PCGX_PROF_STRICT_CHAIN, st);
    launch_chain(W, match, pos_of, state, sums10, kp, fuse_update ? 1 : 0, W.nchunks > 1, st);
  }
  PCGX_HIP_TRY(hipGetLastError());
  return PCGX_OK;
}

// ---- the reference's sums over a target spread over ranks ----------------------------------------------------------
// The sequential order is the ranks' tiles one after the other, rank 0's first: a sharded Fit then returns what the
// reference's Fit returns on that concatenated target, bit for bit.  Everything that is parallel on one GPU stays
// parallel and local -- correspondence, tile sums, summaries, jobs; what a rank needs from the ranks before it is
// three numbers per sum: the float64 total of their terms and of their chains' rounding errors (where its guesses
// start: two all-gathers of 9 doubles per rank) and the state their walk ended in (where its walk starts: the walk is
// one dependent chain, so it goes round the ranks, one hop of 9 states per rank).  2 + world small collectives per
// iteration instead of one; slot [10] of every one carries the ranks' error flag.
__global__ __launch_bounds__(64) void strict_row_totals_kernel(const double *__restrict__ per_tile, const uint32_t *__restrict__ tile_pairs,
                                                               int64_t ntiles, int nrows, const IcpState *__restrict__ state,
                                                               double *__restrict__ slot /* this rank's [16] */) {
  const int row = blockIdx.x, lane = threadIdx.x;
  if (state->done) return;
  if (row < nrows) {
    double v = 0.0;
    for (int64_t t = lane; t < ntiles; t += 64) v += per_tile[(int64_t)row * ntiles + t];
    v = wave_allsum_f64(v);
    if (lane == 0) slot[row] = v;
  } else if (row == kStrictRows && tile_pairs) {
    double v = 0.0;
    for (int64_t t = lane; t < ntiles; t += 64) v += (double)tile_pairs[t];
    v = wave_allsum_f64(v);
    if (lane == 0) slot[kStrictRows] = v;
  }
}
// out[r] = sum over the ranks before `rank` of their slot [r] (rank order); out[9] = the pairs of ALL ranks
__global__ void strict_base_kernel(const double *__restrict__ slots, int rank, int world, double *__restrict__ out) {
  const int r = threadIdx.x;
  if (r < kStrictRows) {
    double v = 0.0;
    for (int k = 0; k < rank; k++) v += slots[k * 16 + r];
    out[r] = v;
  } else if (r == kStrictRows) {
    double v = 0.0;
    for (int k = 0; k < world; k++) v += slots[k * 16 + r];
    out[r] = v;
  }
}
__global__ void strict_hop_kernel(const double *__restrict__ hop, uint32_t *__restrict__ start_bits) {
  if (threadIdx.x < kStrictRows) start_bits[threadIdx.x] = (uint32_t)hop[threadIdx.x];
}
__global__ void strict_zero_kernel(double *__restrict__ p, int n, double last) {
  if ((int)threadIdx.x < n) p[threadIdx.x] = ((int)threadIdx.x == n - 1) ? last : 0.0;
}
// the sums behind the last rank's walk + evaluate tail + pose update (every rank, redundantly)
__global__ void strict_finish_kernel(const uint32_t *__restrict__ end_bits, const double *__restrict__ err_base /* [9] = pairs */,
                                     const double *__restrict__ failed, int nrows, IcpState *__restrict__ state,
                                     double *__restrict__ sums10, IcpKernelParams kp) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (state->done) return;
  if (*failed != 0.0) {  // a rank could not go on: the Fit ends here on every rank
    state->status = PCGX_E_RCCL;
    state->done = 1;
    return;
  }
  double sums[S_COUNT];
  for (int row = 0; row < nrows; row++) {
    const int slot = row == 0 ? S_VALUE : (row <= 6 ? S_G0 + row - 1 : (row == 7 ? S_DIST_RMS : S_WEIGHT));
    sums[slot] = (double)u2f(end_bits[row]);
  }
  const double np = err_base[kStrictRows];
  sums[S_PAIRS] = np;
  if (nrows < kStrictRows) sums[S_WEIGHT] = np < 16777216.0 ? np : 16777216.0;  // 0 + 1 + 1 + ... in float32 (strict_chain_kernel)
  for (int k = 0; k < S_COUNT; k++) sums10[k] = sums[k];
  icp_update_step(state, sums, kp);
}

pcgx_status strict_enqueue_sharded(StrictBuffers *b, const float4 *match, const uint32_t *pos_of, IcpState *state, double *sums10,
                                   const IcpKernelParams &kp, pcgx_comm *c, int rank, int world, bool local_failed, hipStream_t st) {
  if (!b->shard || b->shard_world != world) {
    dev_cache_free(b->shard);
    b->shard = nullptr;
    if (dev_cache_alloc((void **)&b->shard, (size_t)(world + 4) * 16 * sizeof(double)) != hipSuccess)
      return fail(PCGX_E_OOM, "strict sums over ranks: no memory for the exchange");
    b->shard_world = world;
  }
  b->shard_ring = false;
  double *slots = b->shard, *row_base = slots + (size_t)world * 16, *err_base = row_base + 16, *hop = row_base + 32;
  uint32_t *start_bits = reinterpret_cast<uint32_t *>(hop + 16);
  (void)strict_work(b, kp);
  StrictWork W = next_epoch(b);  // (the session's descriptor stays in its one-GPU form)
  W.row_base = row_base;
  W.err_base = err_base;
  W.start_bits = start_bits;
  W.hop_out = hop;
  W.first_exact = rank == 0 ? 1 : 0;
  // A rank that cannot go on (`local_failed`: a launch error, an injected fault) launches nothing but keeps calling the
  // collectives with its flag up (word [10] of its slot): the others learn of it in a collective they all reach anyway,
  // nobody is left waiting; strict_finish_kernel ends the Fit on every rank in the same iteration.
  auto exchange = [&](double *d_buf, int count, int flag_at) -> pcgx_status {
    if (local_failed) {
      static const double one = 1.0;
      PCGX_HIP_TRY(hipMemcpyAsync(d_buf + flag_at, &one, sizeof one, hipMemcpyHostToDevice, st));
    }
    return pcgx_comm_allreduce_f64(c, d_buf, count, st);
  };
  const int nslots = world * 16;
  // 1. the float64 totals of this rank's terms -> everybody -> the totals of the ranks before this one
  PCGX_HIP_TRY(hipMemsetAsync(slots, 0, (size_t)nslots * sizeof(double), st));
  if (!local_failed) {
    ProfScope prof(PCGX_PROF_STRICT_TERMS, st);
    hipLaunchKernelGGL(strict_tilesum_kernel, dim3((unsigned)W.ntiles), dim3(kTileSumBlock), 0, st, match, pos_of,
                       (const IcpState *)state, W);
    hipLaunchKernelGGL(strict_row_totals_kernel, dim3(kStrictRows), dim3(64), 0, st, (const double *)W.tile_sum,
                       (const uint32_t *)nullptr, W.ntiles, W.nrows, (const IcpState *)state, slots + (size_t)rank * 16);
  }
  PCGX_TRY(exchange(slots, nslots, rank * 16 + 10));
  if (!local_failed) {
    hipLaunchKernelGGL(strict_base_kernel, dim3(1), dim3(64), 0, st, (const double *)slots, rank, world, row_base);
    ProfScope prof(PCGX_PROF_STRICT_SUM, st);
    hipLaunchKernelGGL(strict_sum_kernel<false>, dim3((unsigned)W.ntiles), dim3(kSumBlock), 0, st, match, pos_of,
                       (const IcpState *)state, W);
  }
  // 2. the chains' rounding errors and the pair counts -> everybody
  PCGX_HIP_TRY(hipMemsetAsync(slots, 0, (size_t)nslots * sizeof(double), st));
  if (!local_failed)
    hipLaunchKernelGGL(strict_row_totals_kernel, dim3(kStrictRows + 1), dim3(64), 0, st, (const double *)W.tile_err,
                       (const uint32_t *)W.tile_pairs, W.ntiles, W.nrows, (const IcpState *)state, slots + (size_t)rank * 16);
  PCGX_TRY(exchange(slots, nslots, rank * 16 + 10));
  hipLaunchKernelGGL(strict_base_kernel, dim3(1), dim3(64), 0, st, (const double *)slots, rank, world, err_base);
  if (!local_failed) {
    ProfScope prof(PCGX_PROF_STRICT_JOB, st);
    if (W.naux > 0)
      hipLaunchKernelGGL(strict_job_kernel, dim3((unsigned)kJobRoles * (unsigned)W.naux), dim3(kJobBlock), 0, st, (const IcpState *)state, W);
  }
  // 3. the walk goes round the ranks: every hop hands on nine states (their bits as float64: exact under the sum)
  hipLaunchKernelGGL(strict_zero_kernel, dim3(1), dim3(64), 0, st, reinterpret_cast<double *>(start_bits), 8, 0.0);  // 16 x 0.0f
  for (int k = 0; k < world; k++) {
    hipLaunchKernelGGL(strict_zero_kernel, dim3(1), dim3(64), 0, st, hop, 16, 0.0);
    if (k == rank && !local_failed) {
      ProfScope prof(PCGX_PROF_STRICT_CHAIN, st);
      launch_chain(W, match, pos_of, state, sums10, kp, 0, W.nchunks > 1, st);
    }
    PCGX_TRY(exchange(hop, 16, 10));
    hipLaunchKernelGGL(strict_hop_kernel, dim3(1), dim3(64), 0, st, (const double *)hop, start_bits);
  }
  hipLaunchKernelGGL(strict_finish_kernel, dim3(1), dim3(64), 0, st, (const uint32_t *)start_bits, (const double *)err_base,
                     (const double *)(hop + 10), W.nrows, state, sums10, kp);
  PCGX_HIP_TRY(hipGetLastError());
  return PCGX_OK;
}

// ---- the same sums, the RING form: no collective, every rank's kernels resident at once -------------------------------
// What a rank needs from the ranks before it has not changed -- the float64 totals of their terms (before its
// summaries' guesses), of their chains' rounding errors (before its job tiles' guesses), and the state their walk
// ended in (before its walk) -- but nothing is a collective any more: every rank owns an inbox in host-coherent memory
// (pinned host memory of the one process, or a shared-memory segment of the node's processes: comm.hip) that the other
// GPUs' kernels write with plain 64-bit stores and its own kernels poll; a word carries its step's number
// (StrictWork::ring, RingLayout).  Per step and rank: the summary kernel (its last tile sends the rank's totals on,
// its first tile fetches the earlier ranks'), one small kernel for the rounding errors' totals, the job kernel, the
// chain kernel -- every rank's compositions, run lists and candidate tables are made at once, only the walkers
// wait, each for ONE word from the rank before it; the last rank's walkers send the sums to everybody and every rank
// runs the evaluate tail + pose update in its own chain kernel, as on one GPU.  A rank that cannot go on raises the
// abort word in every inbox from the host (ring_abort_from_host); a wait for a state is bounded by kRingWalkTicks and
// raises it too.
__global__ __launch_bounds__(64) void strict_ring_err_kernel(const IcpState *__restrict__ state, StrictWork W) {
  const int lane = threadIdx.x, b = blockIdx.x;
  if (state->done) return;
  const RingLayout RL{W.world};
  if (b < kStrictRows) {  // this rank's total of sum b's rounding errors, to the ranks behind it
    if (b >= W.nrows || W.rank >= W.world - 1) return;
    double v = 0.0;
    for (int64_t t = lane; t < W.ntiles; t += 64) v += W.tile_err[(int64_t)b * W.ntiles + t];
    v = wave_allsum_f64(v);
    if (lane < W.world - 1 - W.rank)
      ring_put_f64(ring_inbox(W, W.rank + 1 + lane) + RL.err_tot(W.rank, b), v, W.ring_epoch);
    return;
  }
  // the totals of the ranks before this one, added up in rank order: where the job tiles' guesses start
  double acc = 0.0;
  for (int k0 = 0; k0 < W.rank; k0 += 7) {  // uniform
    const int k = k0 + lane / kStrictRows, row = lane % kStrictRows;
    double v = 0.0;
    if (lane < 7 * kStrictRows && k < W.rank && row < W.nrows) (void)ring_wait_f64(W, RL.err_tot(k, row), v, 10 * W.ring_guess_ticks);
#pragma unroll
    for (int j = 0; j < 7; j++) acc += __shfl(v, j * kStrictRows + (lane < kStrictRows ? lane : 0));
  }
  if (lane < kStrictRows) W.ring_base[16 + lane] = lane < W.nrows ? acc : 0.0;
}
__global__ void strict_ring_fail_kernel(IcpState *__restrict__ state) {
  if (threadIdx.x == 0 && blockIdx.x == 0 && !state->done) {
    state->status = PCGX_E_RCCL;
    state->done = 1;
  }
}

pcgx_status strict_enqueue_ring(StrictBuffers *b, const float4 *match, const uint32_t *pos_of, IcpState *state, double *sums10,
                                const IcpKernelParams &kp, const RingView &ring, bool local_failed, bool first_iter, hipStream_t st) {
  // (the ring form keeps 32 doubles + a flag word of it; a block the collective form used holds that form's doubles
  // where the flag word lies -- one equal to this step's number would let tiles read ring_base before tile 0 wrote it)
  if (!b->shard || b->shard_world != ring.world || !b->shard_ring) {
    if (!b->shard || b->shard_world != ring.world) {
      dev_cache_free(b->shard);
      b->shard = nullptr;
      if (dev_cache_alloc((void **)&b->shard, (size_t)(ring.world + 4) * 16 * sizeof(double)) != hipSuccess)
        return fail(PCGX_E_OOM, "strict sums over ranks: no memory for the exchange");
      b->shard_world = ring.world;
    }
    PCGX_HIP_TRY(hipMemsetAsync(b->shard, 0, (size_t)(ring.world + 4) * 16 * sizeof(double), st));
    b->shard_ring = true;
  }
  if (local_failed) {
    // this rank launches nothing more: the others learn of it from the abort word, in whatever wait they are in
    ring_abort_from_host(ring, 1u);
    hipLaunchKernelGGL(strict_ring_fail_kernel, dim3(1), dim3(64), 0, st, state);
    PCGX_HIP_TRY(hipGetLastError());
    return PCGX_OK;
  }
  (void)strict_work(b, kp);
  StrictWork W = next_epoch(b);  // (the session's descriptor stays in its one-GPU form)
  W.ring = ring.words;
  W.ring_words = ring.words_per_rank;
  W.ring_tab = ring.tab;
  W.ring_mine = ring.mine;
  W.ring_guess_ticks = ring.guess_ticks > 0 ? ring.guess_ticks : kRingGuessTicks;
  W.rank = ring.rank;
  W.world = ring.world;
  W.ring_epoch = ring.epoch;
  W.ring_base = b->shard;
  W.ring_flag = reinterpret_cast<unsigned int *>(b->shard + 32);
  W.first_exact = ring.rank == 0 ? 1 : 0;
  W.exchange = 1;  // (the ring form rides on the summary kernel's own exchange)
  {
    ProfScope prof(PCGX_PROF_STRICT_SUM, st);
    hipLaunchKernelGGL((strict_sum_kernel<true, true>), dim3((unsigned)W.ntiles), dim3(kSumBlock), 0, st, match, pos_of,
                       (const IcpState *)state, W);
  }
  if (ring.world > 1)
    hipLaunchKernelGGL(strict_ring_err_kernel, dim3(kStrictRows + (ring.rank > 0 ? 1 : 0)), dim3(64), 0, st, (const IcpState *)state, W);
  if (ring.rank > 0) {  // (the job tiles' guesses: + the ranks before this one)
    W.row_base = W.ring_base;
    W.err_base = W.ring_base + 16;
  }
  {
    ProfScope prof(PCGX_PROF_STRICT_JOB, st);
    if (first_iter && W.naux > 0 && W.ntiles >= kRepairMinTiles && repair_enabled())
      hipLaunchKernelGGL(strict_repair_kernel, dim3((unsigned)(W.nrows * ((W.ntiles + kRepairBlock - 1) / kRepairBlock))), dim3(kRepairBlock), 0,
                         st, match, pos_of, (const IcpState *)state, W);
    if (W.naux > 0)
      hipLaunchKernelGGL(strict_job_kernel, dim3((unsigned)kJobRoles * (unsigned)W.naux), dim3(kJobBlock), 0, st, (const IcpState *)state, W);
  }
  {
    ProfScope prof(PCGX_PROF_STRICT_CHAIN, st);
    launch_chain(W, match, pos_of, state, sums10, kp, 1, true, st);
  }
  PCGX_HIP_TRY(hipGetLastError());
  return PCGX_OK;
}

// a session between two Fits: whatever a launch that ended early (a broken ring) left in the counters
pcgx_status strict_reset(StrictBuffers *b, hipStream_t st) {
  if (!b) return PCGX_OK;
  PCGX_HIP_TRY(hipMemsetAsync(b->counters, 0, b->counters_bytes, st));
  PCGX_HIP_TRY(hipMemsetAsync(b->w.tile_arrived, 0, b->arrived_bytes, st));
  return PCGX_OK;
}

pcgx_status strict_read_debug(StrictBuffers *b, unsigned long long out[64], hipStream_t st) {
  PCGX_HIP_TRY(hipMemcpyAsync(out, b->w.dbg, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
  PCGX_HIP_TRY(hipStreamSynchronize(st));
  if (const char *path = getenv("PCGX_STRICT_TRACE")) {  // measurement aid: the last launch's workgroup stamps
    std::vector<unsigned long long> h((size_t)b->w.ntiles * 16);
    PCGX_HIP_TRY(hipMemcpy(h.data(), b->w.stamps, h.size() * 8, hipMemcpyDeviceToHost));
    if (FILE *f = fopen(path, "a")) {
      for (int64_t k = 0; k < b->w.ntiles; k++) {
        for (int j = 0; j < 16; j++) fprintf(f, "%llu ", h[(size_t)k * 16 + j]);
        fprintf(f, "\n");
      }
      fprintf(f, "#\n");
      fclose(f);
    }
  }
  if (const char *path = getenv("PCGX_STRICT_KEYS")) {  // measurement aid: the windows of every row's tiles (last launch)
    std::vector<TileRec> h((size_t)kStrictRows * b->w.ntiles);
    PCGX_HIP_TRY(hipMemcpy(h.data(), b->w.recs, h.size() * sizeof(TileRec), hipMemcpyDeviceToHost));
    if (FILE *f = fopen(path, "a")) {
      for (int r = 0; r < kStrictRows; r++) {
        for (int64_t k = 0; k < b->w.ntiles; k++) fprintf(f, "%d:%d ", h[(size_t)r * b->w.ntiles + k].key, h[(size_t)r * b->w.ntiles + k].cons);
        fprintf(f, "\n");
      }
      fprintf(f, "#\n");
      fclose(f);
    }
  }
  PCGX_HIP_TRY(hipMemsetAsync(b->w.dbg, 0, 64 * sizeof(unsigned long long), st));
  return PCGX_OK;
}

}  // namespace pcgx

// The device pipeline (strict_sum / strict_job / strict_chain kernels) on terms given as they are: nine rows of n
// float32 terms -> the nine sequential float32 sums 0.0f + t0 + t1 + ...  What the GPU tests feed with the rows a
// registration never produces (ties at every step, cancellation to zero, subnormals, overflow, NaN): see
// include/pcgx.h.
extern "C" pcgx_status pcgx_debug_strict_sum_dev(const float *terms, int64_t n, float out[9], int64_t stats[64]) {
  PCGX_API_LOCK();
  using namespace pcgx;
  if (n < 1 || !terms || !out) return fail(PCGX_E_INVALID, "pcgx_debug_strict_sum_dev: bad argument");
  if (n > 0x7fffffffll) return fail(PCGX_E_TOO_LARGE, "pcgx_debug_strict_sum_dev: more than 2^31-1 terms per row");
  PCGX_TRY(ensure_init());
  hipStream_t st = ctx().stream;
  float *d_terms = nullptr, *d_zero = nullptr;
  uint32_t *d_pos = nullptr;
  IcpState *d_state = nullptr;
  double *d_sums = nullptr;
  StrictBuffers *b = nullptr;
  pcgx_status rc = PCGX_OK;
  auto cleanup = [&]() {
    (void)hipStreamSynchronize(st);
    if (b) strict_destroy(b);
    (void)hipFree(d_terms);
    (void)hipFree(d_zero);
    (void)hipFree(d_pos);
    (void)hipFree(d_state);
    (void)hipFree(d_sums);
  };
  hipError_t e = hipMalloc((void **)&d_terms, (size_t)kStrictRows * n * sizeof(float));
  if (e == hipSuccess) e = hipMalloc((void **)&d_zero, (size_t)n * sizeof(float));
  if (e == hipSuccess) e = hipMalloc((void **)&d_pos, (size_t)n * sizeof(uint32_t));
  if (e == hipSuccess) e = hipMalloc((void **)&d_state, sizeof(IcpState));
  if (e == hipSuccess) e = hipMalloc((void **)&d_sums, 16 * sizeof(double));
  if (e == hipSuccess) e = hipMemcpyAsync(d_terms, terms, (size_t)kStrictRows * n * sizeof(float), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemsetAsync(d_zero, 0, (size_t)n * sizeof(float), st);
  if (e == hipSuccess) e = hipMemsetAsync(d_pos, 0, (size_t)n * sizeof(uint32_t), st);
  if (e == hipSuccess) e = hipMemsetAsync(d_state, 0, sizeof(IcpState), st);
  if (e == hipSuccess) e = hipMemsetAsync(d_sums, 0, 16 * sizeof(double), st);
  if (e != hipSuccess) {
    cleanup();
    return fail(PCGX_E_HIP, "pcgx_debug_strict_sum_dev: %s", hipGetErrorString(e));
  }
  rc = strict_create(n, d_zero, d_zero, d_zero, d_pos, &b, st);
  if (rc != PCGX_OK) {
    cleanup();
    return rc;
  }
  b->w.raw_terms = d_terms;
  IcpKernelParams kp;
  memset(&kp, 0, sizeof kp);
  kp.weight_fn = PCGX_WEIGHT_CONSTANT;  // nine rows: the ninth sum is chained like the others
  kp.weight_a = 1.0f;
  rc = strict_enqueue(b, nullptr, nullptr, d_state, d_sums, kp, false, false, true, st);
  double h[16];
  if (rc == PCGX_OK) {
    e = hipMemcpyAsync(h, d_sums, sizeof h, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) rc = fail(PCGX_E_HIP, "pcgx_debug_strict_sum_dev: %s", hipGetErrorString(e));
  }
  if (rc == PCGX_OK) {
    // component order of sums10: Value, G0..G5, DistRMS, Weight, Pairs = rows 0..8, then the pair count
    for (int k = 0; k < kStrictRows; k++) out[k] = (float)h[k];
    if (stats) {
      unsigned long long dbg[64];
      rc = strict_read_debug(b, dbg, st);
      for (int k = 0; k < 64; k++) stats[k] = (int64_t)dbg[k];
    }
  }
  cleanup();
  return rc;
}

// Host model of the same pipeline (no GPU involved): the CPU tests run it against a plain
// sequential float32 loop.  See include/pcgx.h.
extern "C" pcgx_status pcgx_debug_strict_sum_host(const float *terms, int64_t n, int32_t mode, float *out,
                                                  int64_t stats[8]) {
  if (n < 0 || (n > 0 && !terms) || !out || !stats) return pcgx::fail(PCGX_E_INVALID, "pcgx_debug_strict_sum_host: bad argument");
  *out = pcgx::ss::ss_host_model(terms, n, stats, mode);
  return PCGX_OK;
}
