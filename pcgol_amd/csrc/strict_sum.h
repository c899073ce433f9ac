// strict_sum.h -- the reference's SEQUENTIAL float32 sums, evaluated exactly and in parallel.
//
// Reference: pc/registration/icp/evaluator.go:122-145 adds the evaluator's nine float32 terms of
// one pair after the other, in target order, in a single goroutine:  s <- fl32(s + t_i).  Float
// addition is not associative, so a tree reduction does not give those bits (at 1M pairs the
// reference's own rounding noise is 1.6e-5 on the final pose).  This file holds the arithmetic that
// lets thousands of waves reproduce that chain bit for bit; it is compiled for the device (the
// strict_* kernels, strict.hip) and for the host (ss_host_model below: the same arithmetic in plain
// loops, run by the CPU tests against a sequential float32 loop).
//
// Idea.  Inside one binade the float32 grid is uniform: for a state s = m * ulp and any term t,
// fl32(s + t) = (m + RN(t / ulp)) * ulp with round-half-even on the integer m, whatever m is, as
// long as the result stays in the binade.  So the effect of a whole run of additions on a state is
// a TRANSLATION of its mantissa integer that depends only on the parity class of m (ties) -- and it
// can be obtained by simply running the hardware's float adds from one representative state per
// class.  Two adjacent binades ("window" e: exponent fields e-1 and e, states counted in units of
// the lower binade's ulp, n in [2^23, 2^25), even above 2^24) work the same way with four classes
// (n mod 4), provided a translated state lands on the same side of 2^24 as the representative does
// at every step.  A Summary records, per class, the translation and the interval of inputs for
// which that proviso holds; summaries compose associatively (compose()), so leaves (kLeaf terms per
// lane) fold into tiles (one wave), tiles into runs, and a single wave applies the run summaries
// one after the other to the exact state.  Where the interval check fails -- a state that lands
// within a few ulps of a binade boundary differently from its representative -- the tile is
// recomputed exactly from the known state (resolve: leaf by leaf, a leaf whose own summary does not
// cover the state is simply added term by term).
// Every accepted step is proven equal to the sequential result, so the sum is bit-identical by
// construction; how often the slow path runs only affects speed.
//
// The representatives come from a GUESS of the state at every leaf: the float64 prefix sum of the
// terms (tile sums from strict_terms_kernel, leaf sums inside the tile), refined once inside the tile
// by the prefix of the ROUNDING ERRORS the chains make when started from those first guesses (error
// of a leaf = (end - start) - sum of its terms; translation invariance makes it the true error
// whenever the guess is in the right class interval).  A guess is typically within tens of ulps of
// the true state -- close enough everywhere except next to a level crossing, which is what the
// exact recomputation is for.  (A whole-row error prefix was tried as a kernel of its own: it cost
// 18 us per iteration at 1M pairs and removed only a third of the recomputations.)
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define SS_HD __host__ __device__ __forceinline__
#else
#define SS_HD inline
#endif

namespace pcgx {
namespace ss {

constexpr int kLeaf = 32;              // terms per lane
constexpr int kLanes = 64;             // leaves per tile (one wave)
constexpr int kTile = kLeaf * kLanes;  // 2048 terms
constexpr int kBinTiles = 64;          // tiles per level-1 bin of the prefix sums

constexpr int32_t N23 = 1 << 23, N24 = 1 << 24, N25 = 1 << 25;
constexpr int32_t kBig = 1 << 29;    // bound sentinels: lo = -kBig "any", lo = +kBig "none"
constexpr int32_t kClamp = 1 << 28;  // translations are clamped here (far outside any window)

SS_HD uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
SS_HD float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
SS_HD int32_t imin(int32_t a, int32_t b) { return a < b ? a : b; }
SS_HD int32_t imax(int32_t a, int32_t b) { return a > b ? a : b; }
SS_HD uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
SS_HD uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }
SS_HD int32_t iclamp(int32_t v, int32_t lo, int32_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

// For inputs n = r (mod 4) with lo[r] <= n <= hi[r]:  n_out = n + c[r].
struct Summary {
  int32_t c[4], lo[4], hi[4];
};

// key = (sign << 8) | e, e = exponent field of the window's UPPER binade (2..254); -1: no summary.
// Point record: the guess chains of the tile joined up exactly (every leaf's guess equals the end of
// the chain before it), so for the ONE input state `in` the result is `out` -- plain float adds, no
// class argument needed.  It catches what the class intervals must exclude: sums whose every step
// is exact (all-ones weights) land exactly on binade boundaries, and their guesses are exact.
struct TileRec {
  int32_t key;
  uint32_t in, out;
  int32_t cons;  // point record valid
  Summary s;
};
static_assert(sizeof(TileRec) == 64, "one 64-byte record");

SS_HD Summary summary_identity() {
  Summary S;
  for (int r = 0; r < 4; r++) {
    S.c[r] = 0;
    S.lo[r] = -kBig;
    S.hi[r] = kBig;
  }
  return S;
}

// v[r] for r in 0..3 with the four values already in registers (two selects, no memory indexing)
SS_HD int32_t pick4(int r, int32_t v0, int32_t v1, int32_t v2, int32_t v3) {
  const int32_t a = (r & 1) ? v1 : v0, b = (r & 1) ? v3 : v2;
  return (r & 2) ? b : a;
}

// X first, then Y
SS_HD Summary compose(const Summary &X, const Summary &Y) {
  Summary Z;
  const int32_t yc0 = Y.c[0], yc1 = Y.c[1], yc2 = Y.c[2], yc3 = Y.c[3];
  const int32_t yl0 = Y.lo[0], yl1 = Y.lo[1], yl2 = Y.lo[2], yl3 = Y.lo[3];
  const int32_t yh0 = Y.hi[0], yh1 = Y.hi[1], yh2 = Y.hi[2], yh3 = Y.hi[3];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int32_t xc = X.c[r];
    const int q = (r + xc) & 3;
    Z.c[r] = iclamp(xc + pick4(q, yc0, yc1, yc2, yc3), -kClamp, kClamp);
    Z.lo[r] = imax(X.lo[r], iclamp(pick4(q, yl0, yl1, yl2, yl3) - xc, -kBig, kBig));
    Z.hi[r] = imin(X.hi[r], iclamp(pick4(q, yh0, yh1, yh2, yh3) - xc, -kBig, kBig));
  }
  return Z;
}

// state (float bits) -> units of the lower binade's ulp inside window `key`; -1: not in the window
SS_HD int32_t state_to_n(uint32_t bits, int32_t key) {
  if (key < 0) return -1;
  const int32_t sg = (int32_t)(bits >> 31), E = (int32_t)((bits >> 23) & 0xff), e = key & 0xff;
  if (sg != (key >> 8)) return -1;
  const int32_t mant = (int32_t)((bits & 0x7fffffu) | 0x800000u);
  if (E == e - 1) return mant;
  if (E == e) return mant << 1;
  return -1;
}

SS_HD uint32_t n_to_state(int32_t n, int32_t key) {
  const uint32_t sg = (uint32_t)(key >> 8) << 31;
  const uint32_t e = (uint32_t)(key & 0xff);
  if (n < N24) return sg | ((e - 1u) << 23) | ((uint32_t)n & 0x7fffffu);
  return sg | (e << 23) | (((uint32_t)n >> 1) & 0x7fffffu);
}

// the exact effect of the summarised additions on `bits`, if the summary covers that state
SS_HD bool apply(uint32_t &bits, int32_t key, const Summary &S) {
  const int32_t c0 = S.c[0], c1 = S.c[1], c2 = S.c[2], c3 = S.c[3];
  const int32_t l0 = S.lo[0], l1 = S.lo[1], l2 = S.lo[2], l3 = S.lo[3];
  const int32_t h0 = S.hi[0], h1 = S.hi[1], h2 = S.hi[2], h3 = S.hi[3];
  const int32_t n = state_to_n(bits, key);
  if (n < 0) return false;
  const int r = n & 3;
  const int32_t lo = pick4(r, l0, l1, l2, l3), hi = pick4(r, h0, h1, h2, h3), c = pick4(r, c0, c1, c2, c3);
  if (n < lo || n > hi) return false;
  bits = n_to_state(n + c, key);
  return true;
}

// ---- leaf: pass 1, the guess chain ------------------------------------------------------------
struct ChainRange {
  uint32_t mn, mx;  // smallest / largest magnitude bits over the kLeaf + 1 states
  uint32_t sg_or, sg_and;
  uint32_t end;
};

// The chains take the leaf's terms four at a time from `quad(v)` (v = 0 .. kLeaf / 4 - 1: anything with
// members x, y, z, w): the kernels read them from LDS as they go instead of holding 32 registers.
struct Quad {
  float x, y, z, w;
};
struct ArrayQuads {  // a leaf in an array
  const float *t;
  SS_HD Quad operator()(int v) const { return Quad{t[4 * v], t[4 * v + 1], t[4 * v + 2], t[4 * v + 3]}; }
};

// The extremes of a chain are tracked as float min / max of the states themselves (one instruction each per
// term; the bit patterns' magnitude, sign-or and sign-and cost five) and turned into a ChainRange at the end:
// with one sign throughout, the smallest and largest magnitude are |min| and |max| in some order; with both
// signs sg_or != sg_and says so and no caller looks at mn / mx then.  min / max drop a NaN operand, but a chain
// that has seen one ends in one: mx then reads as NaN's magnitude, as it did.
SS_HD ChainRange chain_range_from(float lo, float hi, float end) {
  ChainRange R;
  const uint32_t bl = f2u(lo), bh = f2u(hi), ml = bl & 0x7fffffffu, mh = bh & 0x7fffffffu;
  R.mn = umin(ml, mh);
  R.mx = umax(ml, mh);
  R.sg_or = (bl | bh) >> 31;
  R.sg_and = (bl & bh) >> 31;
  R.end = f2u(end);
  if (end != end) R.mx = 0x7fc00000u;
  return R;
}

template <class Q>
SS_HD ChainRange guess_chain_q(Q quad, uint32_t g0) {
  float s = u2f(g0), lo = s, hi = s;
  auto step = [&](float term) {
    s = s + term;
    lo = __builtin_fminf(lo, s);
    hi = __builtin_fmaxf(hi, s);
  };
#pragma unroll
  for (int v = 0; v < kLeaf / 4; v++) {
    const auto a = quad(v);
    step(a.x); step(a.y); step(a.z); step(a.w);
  }
  return chain_range_from(lo, hi, s);
}

// Two chains over the same terms at once, from g0 and from g0 ^ 1 (the neighbour of the other parity): the
// additions as packed float32 adds.
template <class Q>
SS_HD void guess_chain_pair_q(Q quad, uint32_t g0, ChainRange &A, ChainRange &B) {
  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f s = {u2f(g0), u2f(g0 ^ 1u)};
  float lo0 = s.x, hi0 = s.x, lo1 = s.y, hi1 = s.y;
  auto step = [&](float term) {
    s = s + (v2f){term, term};
    lo0 = __builtin_fminf(lo0, s.x);
    hi0 = __builtin_fmaxf(hi0, s.x);
    lo1 = __builtin_fminf(lo1, s.y);
    hi1 = __builtin_fmaxf(hi1, s.y);
  };
#pragma unroll
  for (int v = 0; v < kLeaf / 4; v++) {
    const auto a = quad(v);
    step(a.x); step(a.y); step(a.z); step(a.w);
  }
  A = chain_range_from(lo0, hi0, s.x);
  B = chain_range_from(lo1, hi1, s.y);
}
SS_HD ChainRange guess_chain(const float *t, uint32_t g0) { return guess_chain_q(ArrayQuads{t}, g0); }

// the chain's end only
template <class Q>
SS_HD uint32_t plain_chain_q(Q quad, uint32_t g0) {
  float s = u2f(g0);
#pragma unroll
  for (int v = 0; v < kLeaf / 4; v++) {
    const auto a = quad(v);
    s = (((s + a.x) + a.y) + a.z) + a.w;
  }
  return f2u(s);
}

// window of a range of magnitudes [mn, mx] with one sign: the two binades that hold it, the level
// between them as near (in log) to `ref` as possible; -1 if no window holds the range
SS_HD int32_t choose_window(uint32_t mn, uint32_t mx, uint32_t sign, uint32_t ref_mag) {
  const int32_t Emin = (int32_t)(mn >> 23), Emax = (int32_t)(mx >> 23);
  if (Emin < 1 || Emax > 254) return -1;  // zero / subnormal / inf / nan
  int32_t e;
  if (Emax - Emin >= 2) return -1;
  if (Emax == Emin + 1) {
    e = Emax;
  } else {
    e = (ref_mag & 0x7fffffu) >= 0x3504F3u ? Emin + 1 : Emin;  // mantissa >= sqrt(2): level above
    if (e < 2) e = Emin + 1;
    if (e > 254) e = Emin;
  }
  if (e < 2 || e > 254) return -1;
  return (int32_t)(sign << 8) | e;
}

// ---- leaf: pass 2, the class summaries ---------------------------------------------------------
// General form: up to four class representatives next to the guess, each run through the leaf with
// real float adds; per class the translation and the interval of inputs that land on the same side
// of the level 2^24 (and inside the window) as the representative at every step.
template <class Q>
SS_HD void class_chain_q(Q quad, uint32_t rep, int32_t key, int32_t &c, int32_t &lo, int32_t &hi) {
  const uint32_t sign = (uint32_t)(key >> 8), e = (uint32_t)(key & 0xff);
  const uint32_t Lbits = e << 23, firstLow = (e - 1u) << 23, lastUp = Lbits | 0x7fffffu;
  float s = u2f(rep);
  uint32_t maxB = 0u, minA = 0xffffffffu, mnAll = 0xffffffffu, mxAll = 0u, sg_bad = 0u;
  auto step = [&](float term) {
    s = s + term;
    const uint32_t b = f2u(s), m = b & 0x7fffffffu;
    sg_bad |= (b >> 31) ^ sign;
    const bool above = m >= Lbits;
    maxB = above ? maxB : umax(maxB, m);
    minA = above ? umin(minA, m) : minA;
    mnAll = umin(mnAll, m);
    mxAll = umax(mxAll, m);
  };
#pragma unroll
  for (int v = 0; v < kLeaf / 4; v++) {
    const auto a = quad(v);
    step(a.x); step(a.y); step(a.z); step(a.w);
  }
  const uint32_t r0 = rep & 0x7fffffffu;
  auto to_n = [&](uint32_t m) -> int32_t {
    const int32_t mant = (int32_t)((m & 0x7fffffu) | 0x800000u);
    return m >= Lbits ? mant << 1 : mant;
  };
  // the representative's own landings must respect their sides (then each of its steps rounds on
  // the grid of the side it lands on, and so does every translated copy within [lo, hi])
  const bool ok = !sg_bad && mnAll >= firstLow + 1u && mxAll <= lastUp && (minA == 0xffffffffu || minA >= Lbits + 1u);
  if (!ok || r0 < firstLow || r0 > lastUp || ((rep >> 31) != sign)) {
    c = 0;
    lo = kBig;
    hi = -kBig;
    return;
  }
  const int32_t n0 = to_n(r0);
  int32_t dlo = (N23 + 1) - to_n(mnAll), dhi = (N25 - 2) - to_n(mxAll);
  if (maxB != 0u) dhi = imin(dhi, (N24 - 1) - to_n(maxB));
  if (minA != 0xffffffffu) dlo = imax(dlo, (N24 + 2) - to_n(minA));
  // (the input state itself is an exact float of the window: where it lies does not matter, the
  // grid a step rounds on depends on where the step LANDS)
  c = to_n(f2u(s) & 0x7fffffffu) - n0;
  lo = n0 + dlo;
  hi = n0 + dhi;
}
SS_HD void class_chain(const float *t, uint32_t rep, int32_t key, int32_t &c, int32_t &lo, int32_t &hi) {
  class_chain_q(ArrayQuads{t}, rep, key, c, lo, hi);
}

SS_HD Summary leaf_summary_general(const float *t, uint32_t guess, int32_t key) {
  Summary S;
  const uint32_t e = (uint32_t)(key & 0xff);
  const uint32_t E = (guess >> 23) & 0xffu;
  if (E == e - 1u) {  // lower binade: n = mantissa, classes by its two low bits
#pragma unroll
    for (int r = 0; r < 4; r++) class_chain(t, (guess & ~3u) | (uint32_t)r, key, S.c[r], S.lo[r], S.hi[r]);
  } else {  // upper binade (or outside: class_chain rejects): n = 2 * mantissa, classes 0 and 2
    class_chain(t, guess & ~1u, key, S.c[0], S.lo[0], S.hi[0]);
    class_chain(t, guess | 1u, key, S.c[2], S.lo[2], S.hi[2]);
    S.c[1] = S.c[3] = 0;
    S.lo[1] = S.lo[3] = kBig;
    S.hi[1] = S.hi[3] = -kBig;
  }
  return S;
}

// One class of the general form, as a function of the class index: what leaf_summary_general puts
// into S.c[r] / S.lo[r] / S.hi[r].  strict_sum_kernel hands the four classes of a leaf to four waves.
template <class Q>
SS_HD void leaf_class_piece_q(Q quad, uint32_t guess, int32_t key, int r, int32_t &c, int32_t &lo, int32_t &hi) {
  c = 0;
  lo = kBig;
  hi = -kBig;
  if (key < 0) return;
  const uint32_t e = (uint32_t)(key & 0xff), E = (guess >> 23) & 0xffu;
  if (E == e - 1u) {
    class_chain_q(quad, (guess & ~3u) | (uint32_t)r, key, c, lo, hi);
  } else if ((r & 1) == 0) {
    class_chain_q(quad, (guess & ~1u) | (uint32_t)(r >> 1), key, c, lo, hi);
  }
}
SS_HD void leaf_class_piece(const float *t, uint32_t guess, int32_t key, int r, int32_t &c, int32_t &lo, int32_t &hi) {
  leaf_class_piece_q(ArrayQuads{t}, guess, key, r, c, lo, hi);
}

// Window of ONE leaf (tiles without a window of their own: sums that hover around zero): the two
// binades that hold its guess chain, -1 if it changes sign or runs through three binades
SS_HD int32_t leaf_key(const ChainRange &cr, uint32_t guess) {
  if (cr.sg_or != cr.sg_and) return -1;
  return choose_window(cr.mn, cr.mx, cr.sg_or, guess & 0x7fffffffu);
}

// Fast form for a leaf whose guess chain stays in ONE binade E of the window: two representatives
// (mantissa even / odd), tracked in bit space (inside a binade the bit pattern IS the mantissa
// integer and its order).
template <class Q>
SS_HD void binade_chain_q(Q quad, uint32_t rep, uint32_t &end, uint32_t &mn, uint32_t &mx, uint32_t &sg_bad) {
  const ChainRange R = guess_chain_q(quad, rep);
  const uint32_t sign = rep >> 31;
  sg_bad = (R.sg_or ^ sign) | (R.sg_and ^ sign);  // some state with the other sign
  mn = R.mn;
  mx = R.mx;
  end = R.end & 0x7fffffffu;
}
SS_HD void binade_chain(const float *t, uint32_t rep, uint32_t &end, uint32_t &mn, uint32_t &mx, uint32_t &sg_bad) {
  binade_chain_q(ArrayQuads{t}, rep, end, mn, mx, sg_bad);
}

SS_HD Summary leaf_summary_binade(const float *t, uint32_t guess, int32_t key) {
  Summary S;
  const uint32_t e = (uint32_t)(key & 0xff), sign = (uint32_t)(key >> 8);
  const uint32_t E = (guess >> 23) & 0xffu;
  const bool upper = E == e;
  const uint32_t first = E << 23, last = first | 0x7fffffu;
#pragma unroll
  for (int p = 0; p < 2; p++) {
    const uint32_t rep = p ? (guess | 1u) : (guess & ~1u);
    uint32_t end, mn, mx, bad;
    binade_chain(t, rep, end, mn, mx, bad);
    const uint32_t r0 = rep & 0x7fffffffu;
    int32_t c, lo, hi;
    if (bad || (rep >> 31) != sign || (E != e && E != e - 1u) || mn < first + 1u || mx > last) {
      c = 0;
      lo = kBig;
      hi = -kBig;
    } else {
      const int32_t f = upper ? 2 : 1;
      const int32_t mant0 = (int32_t)((r0 & 0x7fffffu) | 0x800000u);
      const int32_t n0 = mant0 * f;
      const int32_t dlo = (int32_t)(first + 1u) - (int32_t)mn;
      const int32_t dhi = (int32_t)last - (int32_t)mx;
      c = ((int32_t)end - (int32_t)r0) * f;
      lo = n0 + dlo * f;
      hi = n0 + dhi * f;
    }
    if (upper) {  // n = 2 * mantissa: mantissa parity p <-> n = 2p (mod 4)
      S.c[2 * p] = c;
      S.lo[2 * p] = lo;
      S.hi[2 * p] = hi;
      S.c[2 * p + 1] = 0;
      S.lo[2 * p + 1] = kBig;
      S.hi[2 * p + 1] = -kBig;
    } else {  // n = mantissa: parity p <-> classes p and p + 2
      S.c[p] = S.c[p + 2] = c;
      S.lo[p] = S.lo[p + 2] = lo;
      S.hi[p] = S.hi[p + 2] = hi;
    }
  }
  return S;
}

// ---- tiles that stay in ONE binade: parity summaries --------------------------------------------------
// Inside one binade E only the parity of the mantissa matters (classes r and r + 2 of the lower binade
// coincide; the upper binade has the even classes only), so a summary needs two classes, not four, and
// its composition a quarter of the work.  Units: magnitude bits relative to the binade's first float
// (rel = bits - (E << 23), 0 .. 2^23 - 1).  For inputs of parity p with lo[p] <= rel <= hi[p]:
// rel_out = rel + c[p].  par_expand() restates it as the four-class Summary of a window that holds E.
struct Par {
  int32_t c[2], lo[2], hi[2];
};

SS_HD void par_class(bool ok, uint32_t r0_mag, uint32_t end_mag, uint32_t mn, uint32_t mx, uint32_t E, int32_t &c, int32_t &lo,
                       int32_t &hi) {
  const uint32_t first = E << 23, last = first | 0x7fffffu;
  if (!ok || (r0_mag >> 23) != E || mn < first + 1u || mx > last) {
    c = 0;
    lo = kBig;
    hi = -kBig;
    return;
  }
  const int32_t rel0 = (int32_t)(r0_mag - first);
  c = (int32_t)end_mag - (int32_t)r0_mag;
  lo = rel0 + ((int32_t)(first + 1u) - (int32_t)mn);
  hi = rel0 + ((int32_t)last - (int32_t)mx);
}

// X first, then Y
SS_HD Par par_compose(const Par &X, const Par &Y) {
  Par Z;
#pragma unroll
  for (int p = 0; p < 2; p++) {
    const int32_t xc = X.c[p];
    const bool odd = ((p + xc) & 1) != 0;
    Z.c[p] = iclamp(xc + (odd ? Y.c[1] : Y.c[0]), -kClamp, kClamp);
    Z.lo[p] = imax(X.lo[p], iclamp((odd ? Y.lo[1] : Y.lo[0]) - xc, -kBig, kBig));
    Z.hi[p] = imin(X.hi[p], iclamp((odd ? Y.hi[1] : Y.hi[0]) - xc, -kBig, kBig));
  }
  return Z;
}

SS_HD Par par_identity() {
  Par S;
  S.c[0] = S.c[1] = 0;
  S.lo[0] = S.lo[1] = -kBig;
  S.hi[0] = S.hi[1] = kBig;
  return S;
}

// the four-class summary (units of window `key`) of a parity summary of binade E (E == e or e - 1)
SS_HD Summary par_expand(const Par &P, uint32_t E, int32_t key) {
  Summary S;
  const uint32_t e = (uint32_t)(key & 0xff);
  const bool upper = E == e;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int p = upper ? (r >> 1) : (r & 1);
    const bool none = (upper && (r & 1)) || P.lo[p] > P.hi[p] || (E != e && E != e - 1u);
    const int32_t f = upper ? 2 : 1;
    S.c[r] = none ? 0 : P.c[p] * f;
    S.lo[r] = none ? kBig : iclamp((N23 + P.lo[p]) * f, -kBig, kBig);
    S.hi[r] = none ? -kBig : iclamp((N23 + P.hi[p]) * f, -kBig, kBig);
  }
  return S;
}

// one leaf: the guess chain `cr` (run from `guess`) serves as the chain of the guess's own parity, the
// other parity gets a chain of its own
// (crb: the chain from guess ^ 1, guess_chain_pair_q)
SS_HD Par leaf_parity_summary_pair(uint32_t guess, const ChainRange &cr, const ChainRange &crb, uint32_t E, uint32_t sign) {
  int32_t c0, lo0, hi0, c1, lo1, hi1;  // the guess's own parity | the other one
  par_class(cr.sg_or == sign && cr.sg_and == sign, guess & 0x7fffffffu, cr.end & 0x7fffffffu, cr.mn, cr.mx, E, c0, lo0, hi0);
  const uint32_t rep = guess ^ 1u;
  par_class(crb.sg_or == sign && crb.sg_and == sign, rep & 0x7fffffffu, crb.end & 0x7fffffffu, crb.mn, crb.mx, E, c1, lo1, hi1);
  const bool odd = (guess & 1u) != 0u;  // (selects, no indexing by a run-time value: that would put S into scratch memory)
  Par S;
  S.c[0] = odd ? c1 : c0; S.lo[0] = odd ? lo1 : lo0; S.hi[0] = odd ? hi1 : hi0;
  S.c[1] = odd ? c0 : c1; S.lo[1] = odd ? lo0 : lo1; S.hi[1] = odd ? hi0 : hi1;
  return S;
}
template <class Q>
SS_HD Par leaf_parity_summary_q(Q quad, uint32_t guess, const ChainRange &cr, uint32_t E, uint32_t sign) {
  int32_t c0, lo0, hi0, c1, lo1, hi1;  // the guess's own parity | the other one
  par_class(cr.sg_or == sign && cr.sg_and == sign, guess & 0x7fffffffu, cr.end & 0x7fffffffu, cr.mn, cr.mx, E, c0, lo0, hi0);
  const uint32_t rep = guess ^ 1u;
  uint32_t end, mn, mx, bad;
  binade_chain_q(quad, rep, end, mn, mx, bad);
  par_class(!bad && (rep >> 31) == sign, rep & 0x7fffffffu, end, mn, mx, E, c1, lo1, hi1);
  const bool odd = (guess & 1u) != 0u;  // (selects, no indexing by a run-time value: that would put S into scratch memory)
  Par S;
  S.c[0] = odd ? c1 : c0; S.lo[0] = odd ? lo1 : lo0; S.hi[0] = odd ? hi1 : hi0;
  S.c[1] = odd ? c0 : c1; S.lo[1] = odd ? lo0 : lo1; S.hi[1] = odd ? hi0 : hi1;
  return S;
}
SS_HD Par leaf_parity_summary(const float *t, uint32_t guess, const ChainRange &cr, uint32_t E, uint32_t sign) {
  return leaf_parity_summary_q(ArrayQuads{t}, guess, cr, E, sign);
}

// ---- host model ---------------------------------------------------------------------------------
// The whole pipeline in plain loops (what the strict_* kernels do wave-parallel), for the CPU
// tests.  terms[n] -> the sequential float32 sum of 0.0f + t0 + t1 + ... ; stats: [0] tiles,
// [1] tiles without a record, [2] runs applied, [3] runs that failed, [4] tiles resolved exactly,
// [5] leaves added serially, [6] leaves in the general (crossing) form, [7] tile records that failed.
// mode bit 0: never use the in-binade fast form; bit 1: no refinement of the guesses inside a tile;
// bit 2: no parity summaries (tiles in one binade take the four-class forms).
inline float ss_host_model(const float *terms_in, int64_t n, int64_t stats[8], int mode) {
  for (int k = 0; k < 8; k++) stats[k] = 0;
  const int64_t ntiles = n > 0 ? (n + kTile - 1) / kTile : 1;
  const int64_t npad = ntiles * kTile;
  float *terms = new float[(size_t)npad];
  for (int64_t i = 0; i < npad; i++) terms[i] = i < n ? terms_in[i] : -0.0f;
  double *tile_sum = new double[(size_t)ntiles];
  TileRec *recs = new TileRec[(size_t)ntiles];
  stats[0] = ntiles;
  // strict_terms_kernel: float64 sums per tile
  for (int64_t k = 0; k < ntiles; k++) {
    double v = 0.0;
    for (int i = 0; i < kTile; i++) v += (double)terms[k * kTile + i];
    tile_sum[k] = v;
  }
  // leaf guesses of a tile whose first state is about `base` (tile_guesses of strict.hip)
  auto tile_guesses = [&](const float *tt, double base, uint32_t *guess, ChainRange *cr) {
    double lsum[kLanes], pre[kLanes], err[kLanes];
    double p = 0.0;
    for (int l = 0; l < kLanes; l++) {
      double v = 0.0;
      for (int j = 0; j < kLeaf; j++) v += (double)tt[l * kLeaf + j];
      lsum[l] = v;
      pre[l] = p;
      p += v;
    }
    bool changed = false;
    for (int l = 0; l < kLanes; l++) {
      guess[l] = f2u((float)(base + pre[l]));
      cr[l] = guess_chain(tt + l * kLeaf, guess[l]);
      err[l] = ((double)u2f(cr[l].end) - (double)u2f(guess[l])) - lsum[l];
    }
    if (mode & 2) return;
    double e = 0.0;
    uint32_t g2[kLanes];
    for (int l = 0; l < kLanes; l++) {
      g2[l] = f2u((float)(base + pre[l] + e));
      changed = changed || g2[l] != guess[l];
      e += err[l];
    }
    if (changed)
      for (int l = 0; l < kLanes; l++) {
        guess[l] = g2[l];
        cr[l] = guess_chain(tt + l * kLeaf, guess[l]);
      }
  };
  // strict_sum_kernel: one record per tile
  {
    double P0 = 0.0;
    for (int64_t k = 0; k < ntiles; k++) {
      const float *tt = terms + k * kTile;
      TileRec T;
      memset(&T, 0, sizeof T);
      if (k < 1) {  // the first tiles (kExactTiles of strict.hip): added up from 0.0f -> point records
        static thread_local float carry;
        if (k == 0) carry = 0.0f;
        float x = carry;
        for (int i = 0; i < kTile; i++) x = x + tt[i];
        T.key = -1;
        T.in = f2u(carry);
        T.out = f2u(x);
        carry = x;
        T.cons = 1;
        stats[1]++;
        recs[k] = T;
        P0 += tile_sum[k];
        continue;
      }
      uint32_t guess[kLanes];
      ChainRange cr[kLanes];
      tile_guesses(tt, P0, guess, cr);
      uint32_t mn = 0xffffffffu, mx = 0u, sg_or = 0u, sg_and = 1u;
      for (int l = 0; l < kLanes; l++) {
        mn = umin(mn, cr[l].mn);
        mx = umax(mx, cr[l].mx);
        sg_or |= cr[l].sg_or;
        sg_and &= cr[l].sg_and;
      }
      T.key = sg_or == sg_and ? choose_window(mn, mx, sg_or, guess[0] & 0x7fffffffu) : -1;
      T.in = guess[0];
      T.out = cr[kLanes - 1].end;
      T.cons = 1;
      for (int l = 0; l + 1 < kLanes; l++) T.cons &= guess[l + 1] == cr[l].end;
      if (T.key >= 0 && (mn >> 23) == (mx >> 23) && !(mode & 4)) {  // strict_sum_kernel's plain tiles
        const uint32_t E = mn >> 23;
        Par acc = par_identity();
        for (int l = 0; l < kLanes; l++) acc = par_compose(acc, leaf_parity_summary(tt + l * kLeaf, guess[l], cr[l], E, sg_or));
        T.s = par_expand(acc, E, T.key);
      } else if (T.key >= 0) {
        Summary acc = summary_identity();
        for (int l = 0; l < kLanes; l++) {
          const float *t = tt + l * kLeaf;
          const bool one_binade = (cr[l].mn >> 23) == (cr[l].mx >> 23);
          Summary S;
          if (one_binade && !(mode & 1)) {
            S = leaf_summary_binade(t, guess[l], T.key);
          } else {
            S = leaf_summary_general(t, guess[l], T.key);
            stats[6]++;
          }
          acc = compose(acc, S);
        }
        T.s = acc;
      } else {
        stats[1]++;
      }
      recs[k] = T;
      P0 += tile_sum[k];
    }
  }
  // strict_chain_kernel: runs of equal windows, applied in order; exact recomputation on failure
  uint32_t s = f2u(0.0f);
  auto resolve_tile = [&](int64_t k) {  // resolve_tile of strict.hip
    const float *tt = terms + k * kTile;
    stats[4]++;
    const int32_t key = recs[k].key;
    if (key >= 0) {
      // what the summary kernel keeps for a tile with a level crossing: the leaves' summaries (there as
      // compositions 0..l and l..63); a leaf whose summary does not cover the state is added term by term
      uint32_t guess[kLanes];
      ChainRange cr[kLanes];
      double P0 = 0.0;
      for (int64_t q = 0; q < k; q++) P0 += tile_sum[q];
      tile_guesses(tt, P0, guess, cr);
      for (int l = 0; l < kLanes; l++) {
        const float *t = tt + l * kLeaf;
        const Summary S = (cr[l].mn >> 23) == (cr[l].mx >> 23) && !(mode & 1) ? leaf_summary_binade(t, guess[l], key)
                                                                             : leaf_summary_general(t, guess[l], key);
        if (apply(s, key, S)) continue;
        float x = u2f(s);
        for (int j = 0; j < kLeaf; j++) x = x + t[j];
        s = f2u(x);
        stats[5]++;
      }
      return;
    }
    // a tile without a window (strict_sum_kernel's leaf records): every leaf under a window of its own
    // where it has one, neighbouring leaves of equal windows composed into runs; a run that does not
    // cover the state, and a leaf without a window, are added term by term
    uint32_t guess[kLanes];
    ChainRange cr[kLanes];
    double P0 = 0.0;
    for (int64_t q = 0; q < k; q++) P0 += tile_sum[q];
    tile_guesses(tt, P0, guess, cr);
    int l = 0;
    while (l < kLanes) {
      const int32_t kl = leaf_key(cr[l], guess[l]);
      int e = l;
      Summary acc = summary_identity();
      if (kl >= 0) {
        while (true) {
          Summary S;
          for (int r = 0; r < 4; r++) leaf_class_piece(tt + e * kLeaf, guess[e], kl, r, S.c[r], S.lo[r], S.hi[r]);
          acc = compose(acc, S);
          if (e + 1 < kLanes && leaf_key(cr[e + 1], guess[e + 1]) == kl) e++;
          else break;
        }
        if (apply(s, kl, acc)) {
          l = e + 1;
          continue;
        }
      }
      float x = u2f(s);
      for (int i = l * kLeaf; i < (e + 1) * kLeaf; i++) x = x + tt[i];
      s = f2u(x);
      stats[5] += e + 1 - l;
      l = e + 1;
    }
  };
  int64_t k = 0;
  while (k < ntiles) {
    int64_t k1 = k + 1;
    TileRec R = recs[k];
    while (R.key >= 0 && k1 < ntiles && recs[k1].key == R.key && (k1 % 64) != 0) {  // runs end at wave boundaries
      R.s = compose(R.s, recs[k1].s);
      R.cons = R.cons && recs[k1].cons && R.out == recs[k1].in;
      R.out = recs[k1].out;
      k1++;
    }
    if ((R.key >= 0 && apply(s, R.key, R.s))) {
      stats[2]++;
    } else if (R.cons && R.in == s) {
      s = R.out;
      stats[2]++;
    } else {
      if (R.key >= 0) stats[3]++;
      for (int64_t q = k; q < k1; q++) {
        if (recs[q].key >= 0 && apply(s, recs[q].key, recs[q].s)) continue;
        if (recs[q].cons && recs[q].in == s) {
          s = recs[q].out;
          continue;
        }
        if (recs[q].key >= 0) stats[7]++;
        resolve_tile(q);
      }
    }
    k = k1;
  }
  delete[] terms;
  delete[] tile_sum;
  delete[] recs;
  return u2f(s);
}

}  // namespace ss
}  // namespace pcgx
