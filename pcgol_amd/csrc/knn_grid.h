// knn_grid.h -- certified nearest neighbour on a uniform grid (fast path of KDTree.Nearest).
//
// Reference: pc/storage/kdtree/kdtree.go:83-146.  Nearest is an EXACT search: its pruning
// (`fromPivotSq > best -> skip`, :111-115) never discards a point whose float32 DistSq is below the
// best one (float subtraction, multiplication and addition of non-negative terms are monotonic, so a
// point beyond a splitting plane has a computed DistSq >= the computed plane distance).  It therefore
// returns the point with the smallest computed DistSq -- and WHICH of several points with exactly
// the same DistSq only when there is such a tie, or when DistSq == maxRange^2 exactly (the leaf and
// pivot rules differ there, :100-103 vs :117).  Any other exact search that evaluates DistSq with the
// same expression returns the same {ID, DistSq}.
//
// So every tree also gets a uniform grid over its bounding box (~1.5 points per cell, points stored
// in cell order).  A query scans the cells that can hold a point within its current bound (the
// nearest point found in the cells around it, or the hint the ICP loop carries over), keeps the
// minimum and whether it is tied, and accepts the result only when the scanned cells provably hold
// EVERY point at or below that minimum (grid_cover: monotonic float binning, no tolerances).
// Everything else -- ties, DistSq == maxRange^2, a search region beyond 9 x 9 x 9 cells, non-finite
// queries, MinDistSq > 0 (approximate search, depends on the visit order) -- goes to the tree walk
// (knn_walk.h), which reproduces the reference's visit order.  Results are the walk's, bit for bit;
// the grid only removes the dependent-load chains of the descent for the queries where the answer
// does not depend on them.
#pragma once
#include "pcgx_internal.h"

namespace pcgx {

__device__ __forceinline__ int grid_cell(float v, float lo, float inv_h, int n) {
  // monotonic in v (float subtraction, multiplication by a positive number, clamping and truncation
  // all are): v <= w implies grid_cell(v) <= grid_cell(w).  Everything below rests on that alone.
  float f = (v - lo) * inv_h;
  f = fminf(fmaxf(f, 0.0f), (float)(n - 1));  // NaN -> 0
  return (int)f;
}

struct GridBox {
  int x0, x1, y0, y1, z0, z1;  // inclusive cell ranges
};

// Cells that hold every point whose computed DistSq to q is <= lim.  Such a point has
// |p.x - q.x| <= sqrt(lim) (1 + 2e-7) (DistSq >= fl(dx^2), dx = fl(p.x - q.x)); rad below is larger,
// fl(q.x + rad) >= p.x and fl(q.x - rad) <= p.x by monotonic rounding, and grid_cell is monotonic.
__device__ __forceinline__ GridBox grid_cover(const GridView &g, float qx, float qy, float qz, float lim) {
  const float rad = sqrtf(lim) * 1.0001f;
  GridBox c;
  c.x0 = grid_cell(qx - rad, g.lo[0], g.inv_h, g.nx);
  c.x1 = grid_cell(qx + rad, g.lo[0], g.inv_h, g.nx);
  c.y0 = grid_cell(qy - rad, g.lo[1], g.inv_h, g.ny);
  c.y1 = grid_cell(qy + rad, g.lo[1], g.inv_h, g.ny);
  c.z0 = grid_cell(qz - rad, g.lo[2], g.inv_h, g.nz);
  c.z1 = grid_cell(qz + rad, g.lo[2], g.inv_h, g.nz);
  return c;
}

constexpr int kGridWide = 4;  // a search region may reach this many cells from the query's own one (9 x 9 x 9)

struct GridBest {
  float4 p;  // {x, y, z, bits(id)}; id < 0: none seen
  float d;   // the smallest DistSq seen; +inf: none seen
  float d2;  // the second smallest (of the multiset): == d exactly when a second point sits at d
  __device__ __forceinline__ bool tie() const { return d2 == d; }  // (asked only of a finite d)
};

// start[row + cx - 1 .. row + cx + 2]: the bounds of the three cells cx - 1 .. cx + 1 of a row in one
// 16-byte load (4-byte aligned; start[] is padded by one element on either side)
struct __attribute__((packed, aligned(4))) GridQuad {
  uint32_t v[4];
};
__device__ __forceinline__ uint32_t grid_quad_at(const GridQuad &q, int k) {  // k in 0..3
  return k == 0 ? q.v[0] : (k == 1 ? q.v[1] : (k == 2 ? q.v[2] : q.v[3]));
}

enum GridVerdict { GRID_FOUND = 0, GRID_NONE = 1, GRID_WALK = 2 };

// Tuning / measurement aid (nullptr in the product kernels, where it compiles away): why a query
// was left to the walk and what the scan read.  why: 1 non-finite query, 2 nothing in the 125 cells, 3
// beyond 9 x 9 x 9 cells, 4 DistSq == maxRange^2, 5 tie, 6 bound not met, 7 took the slab-by-slab scan
// (not a walk).
struct GridTrace {
  int why = 0;
  uint32_t points = 0;  // float4 point records read
  uint32_t words = 0;   // uint32 cell bounds read
  uint32_t wave_slots = 0;  // in the first active lane of a scan: 64 x the rounds of the scan loop x 4 (what the wave pays for)
  int rounds9 = -1, rounds4 = -1;  // rounds of 4 of this lane's (last) 9- / 4-segment scan (-1: none)
  uint32_t slots_n[3] = {0, 0, 0}, points_n[3] = {0, 0, 0};  // the same / the records, per kind of scan (4, 9, 5 segments)
};

// The points of up to N segments [seg_s[j], seg_e[j]) of pts[] as ONE sequence, four loads in flight.
template <int N>
__device__ __forceinline__ void grid_scan_segments(const GridView &g, const uint32_t (&seg_s)[N],
                                                   const uint32_t (&seg_e)[N], float qx, float qy, float qz,
                                                   GridBest &best, GridTrace *tr = nullptr) {
  // flat position f lives in segment j iff first[j] <= f < first[j + 1]; its point is pts[f + shift[j]]
  uint32_t first[N], shift[N];
  uint32_t total = 0;
#pragma unroll
  for (int j = 0; j < N; j++) {
    first[j] = total;
    shift[j] = seg_s[j] - total;
    total += seg_e[j] - seg_s[j];
    if (tr && seg_e[j] != seg_s[j]) tr->words += 2;  // (a row without cells to scan costs no useful word)
  }
  if (tr) {
    tr->points += total;
    unsigned long long act = __ballot(1);
    const int first_lane = __ffsll(act) - 1;
    uint32_t mx = 0;
    while (act) {
      const int l = __ffsll(act) - 1;
      mx = max(mx, (uint32_t)__builtin_amdgcn_readlane((int)((total + 3) / 4 * 4), l));
      act &= act - 1;
    }
    if ((int)(threadIdx.x & 63) == first_lane) {
      tr->wave_slots += 64 * mx;
      tr->slots_n[N == 4 ? 0 : (N == 9 ? 1 : 2)] += 64 * mx;
    }
    tr->points_n[N == 4 ? 0 : (N == 9 ? 1 : 2)] += total;
    if (N == 9) tr->rounds9 = (int)((total + 3) / 4);
    if (N == 4) tr->rounds4 = (int)((total + 3) / 4);
  }
  // The loop is what the searches spend their vector instructions on (the C2 search kernel: two thirds of its time is
  // vector issue), so per point, beside the distance (dx, dy as one packed operation each): the smallest and the second
  // smallest distance so far -- a minimum and a median of three, no comparison; "tied" is their equality at the end --
  // one comparison and one select for the winner's id (three more for its x, y, z where the caller wants them).  Whole
  // groups of four are taken without any masking; the last, partial group reads up to three positions behind the
  // sequence (pts[] is padded by three records, the segments behind the last used one begin at 0) and counts them as
  // infinitely far.
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 qxy = {qx, qy};
  float d1 = best.d, d2 = best.d2, bw = best.p.w, bx = best.p.x, by = best.p.y, bz = best.p.z;
  auto group = [&](uint32_t f0, bool partial) {
    float4 p[4];
    uint32_t at[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint32_t f = f0 + u;
      uint32_t sh = shift[0];
#pragma unroll
      for (int j = 1; j < N; j++) sh = f >= first[j] ? shift[j] : sh;
      at[u] = f + sh;
      p[u] = g.pts[at[u]];
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      asm volatile("" : "+v"(p[u].w));  // (the id arrives with x, y, z: not fetched by a branch of its own when the point wins)
      const f32x2 pxy = {p[u].x, p[u].y};
      const f32x2 dxy = pxy - qxy, sq = dxy * dxy;
      const float dz = p[u].z - qz;
      float d = (sq.x + sq.y) + dz * dz;  // the reference's expression (mat/vec3.go:18-20,38-40)
      if (partial && u > 0) d = f0 + u < total ? d : __builtin_inff();
      // selects, not branches: the lanes of a wave rarely agree on which of them improves
      const bool lt = d < d1;
      bw = lt ? p[u].w : bw;
      bx = lt ? p[u].x : bx;  // (a caller that does not look at the winner's coordinates does not pay for these)
      by = lt ? p[u].y : by;
      bz = lt ? p[u].z : bz;
      d2 = __builtin_amdgcn_fmed3f(d1, d2, d);  // the second smallest of {d1 <= d2, d}
      d1 = fminf(d1, d);
    }
  };
  uint32_t f0 = 0;
  for (; f0 + 4 <= total; f0 += 4) group(f0, false);
  if (f0 < total) group(f0, true);
  best.p = make_float4(bx, by, bz, bw);
  best.d = d1;
  best.d2 = d2;
}

// Nearest of an exact-mode query (MinDistSq == 0) if the grid can certify it.  ub: squared distance
// (the same float32 expression) from q to ANY point of the tree, +inf if unknown.  GRID_FOUND: best /
// best_d are the reference's answer; GRID_NONE: {-1, maxRange^2} is (kdtree.go:100-103); GRID_WALK:
// ask the tree walk.
//
// With a useful bound (the ICP loop's hint) the cells covering it are scanned at once.  Without one,
// the 2 x 2 x 2 cells nearest to the query come first (the nearest point is among them most of the
// time), then whatever else of the 3 x 3 x 3 block the distance found there still covers; a
// region reaching beyond that block (sparse spots, queries off a thin cloud) is scanned slab by slab
// up to 9 x 9 x 9 cells.
//
// In three steps, so that a kernel may hand the second scan of its few queries that need one to
// lanes of their own (grid_nearest_kernel): grid_nearest_begin (hinted scan, or the octant and the
// segments of what else must be read), the scan of GridSearch::seg_s/seg_e when `more`, and
// grid_nearest_end (the rest: guesses checked, sparse spots, the verdict).
struct GridSearch {
  GridBest b;
  GridBox box;    // cells that hold every point with DistSq <= min(b.d, bound) once the scans are done
  float bound;    // the answer is a point with DistSq <= bound (ub is attained by a real point)
  bool early;     // verdict known in grid_nearest_begin (GRID_WALK)
  bool cold;      // no useful hint: octant first
  bool guess;     // the block was a guess: grid_nearest_end checks that it covers what was found in it
  bool more;      // seg_s / seg_e hold cells still to be scanned
  uint32_t seg_s[9], seg_e[9];
};

#define PCGX_GRID_WHY(code) do { if (tr) tr->why = (code); } while (0)

__device__ __forceinline__ void grid_nearest_begin(const GridView &g, const float qx, const float qy, const float qz,
                                                   const float max_range_sq, const float ub, GridSearch &S,
                                                   GridTrace *tr = nullptr) {
  S.early = S.cold = S.guess = S.more = false;
  S.b.p = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(-1));
  S.b.d = S.b.d2 = __builtin_inff();
  S.bound = fminf(ub, max_range_sq);
  // non-finite queries: NaN distances follow the walk's comparisons, not an ordering
  if (!(fabsf(qx) < 3.0e38f && fabsf(qy) < 3.0e38f && fabsf(qz) < 3.0e38f) || max_range_sq != max_range_sq) {
    PCGX_GRID_WHY(1);  // (a NaN maxRange^2 compares false with everything: the walk's rules decide)
    S.early = true;
    return;
  }
  const float bound = S.bound;
  GridBest &b = S.b;
  GridBox &box = S.box;
  bool covered = false;
  if (bound < 3.0e38f) {
    box = grid_cover(g, qx, qy, qz, bound);
    covered = box.x1 - box.x0 < 3 && box.y1 - box.y0 < 3 && box.z1 - box.z0 < 3;
  }
  if (covered) {
    // ---- hinted: up to 3 x 3 rows of up to 3 cells, all bounds fetched at once; mostly 2 x 2 rows
    //      or fewer, which take the cheaper 4-segment scan
    if (box.y1 - box.y0 < 2 && box.z1 - box.z0 < 2) {  // (a branch of its own for a single row of cells: slower)
      uint32_t seg_s[4], seg_e[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int y = box.y0 + j % 2, z = box.z0 + j / 2;
        const bool on = y <= box.y1 && z <= box.z1;
        const uint32_t row = (uint32_t)((z * g.ny + y) * g.nx);
        seg_s[j] = on ? g.start[row + (uint32_t)box.x0] : 0u;
        seg_e[j] = on ? g.start[row + (uint32_t)box.x1 + 1u] : 0u;
      }
      grid_scan_segments<4>(g, seg_s, seg_e, qx, qy, qz, b, tr);
    } else {
      uint32_t seg_s[9], seg_e[9];
#pragma unroll
      for (int j = 0; j < 9; j++) {
        const int y = box.y0 + j % 3, z = box.z0 + j / 3;
        const bool on = y <= box.y1 && z <= box.z1;
        const uint32_t row = (uint32_t)((z * g.ny + y) * g.nx);
        seg_s[j] = on ? g.start[row + (uint32_t)box.x0] : 0u;
        seg_e[j] = on ? g.start[row + (uint32_t)box.x1 + 1u] : 0u;
      }
      grid_scan_segments<9>(g, seg_s, seg_e, qx, qy, qz, b, tr);
    }
    return;
  }
  // ---- cold: the octant of the 3 x 3 x 3 block the query sits in, 2 x 2 x 2 cells (fewer at the
  //      grid's border).  Rows (cy, cz), (cy + sy, cz), (cy, cz + sz), (cy + sy, cz + sz).
  S.cold = true;
  const int cx = grid_cell(qx, g.lo[0], g.inv_h, g.nx), cy = grid_cell(qy, g.lo[1], g.inv_h, g.ny),
            cz = grid_cell(qz, g.lo[2], g.inv_h, g.nz);
  const int bx0 = max(cx - 1, 0), bx1 = min(cx + 1, g.nx - 1), by0 = max(cy - 1, 0), by1 = min(cy + 1, g.ny - 1),
            bz0 = max(cz - 1, 0), bz1 = min(cz + 1, g.nz - 1);
  const float ux = (qx - g.lo[0]) * g.inv_h - (float)cx, uy = (qy - g.lo[1]) * g.inv_h - (float)cy,
              uz = (qz - g.lo[2]) * g.inv_h - (float)cz;
  const int sx = ux < 0.5f ? -1 : 1, sy = uy < 0.5f ? -1 : 1, sz = uz < 0.5f ? -1 : 1;  // any choice is valid
  const int ox0 = max(min(cx, cx + sx), bx0), ox1 = min(max(cx, cx + sx), bx1);
  const bool y_on = cy + sy >= by0 && cy + sy <= by1, z_on = cz + sz >= bz0 && cz + sz <= bz1;
  const int oy0 = y_on ? min(cy, cy + sy) : cy, oy1 = y_on ? max(cy, cy + sy) : cy;
  const int oz0 = z_on ? min(cz, cz + sz) : cz, oz1 = z_on ? max(cz, cz + sz) : cz;
  auto load_quad = [&](int y, int z, bool on) {
    GridQuad r;
    if (on) r = *reinterpret_cast<const GridQuad *>(g.start + ((z * g.ny + y) * g.nx + cx - 1));
    else r.v[0] = r.v[1] = r.v[2] = r.v[3] = 0u;
    return r;
  };
  // cells [xa, xb] (within cx - 1 .. cx + 1) of a row
  auto range = [&](const GridQuad &r, int xa, int xb, bool on, uint32_t &s2, uint32_t &e2) {
    on = on && xa <= xb;
    s2 = on ? grid_quad_at(r, xa - (cx - 1)) : 0u;
    e2 = on ? grid_quad_at(r, xb + 1 - (cx - 1)) : 0u;
  };
  const GridQuad r00 = load_quad(cy, cz, true), r10 = load_quad(cy + sy, cz, y_on),
                 r01 = load_quad(cy, cz + sz, z_on), r11 = load_quad(cy + sy, cz + sz, y_on && z_on);
  {
    uint32_t seg_s[4], seg_e[4];
    range(r00, ox0, ox1, true, seg_s[0], seg_e[0]);
    range(r10, ox0, ox1, y_on, seg_s[1], seg_e[1]);
    range(r01, ox0, ox1, z_on, seg_s[2], seg_e[2]);
    range(r11, ox0, ox1, y_on && z_on, seg_s[3], seg_e[3]);
    grid_scan_segments<4>(g, seg_s, seg_e, qx, qy, qz, b, tr);
  }
  const float lim = fminf(b.d, bound);
  S.guess = true;  // nothing found and no bound: the whole block, checked afterwards
  box.x0 = bx0; box.x1 = bx1; box.y0 = by0; box.y1 = by1; box.z0 = bz0; box.z1 = bz1;
  if (lim < 3.0e38f) {
    const GridBox need = grid_cover(g, qx, qy, qz, lim);
    // a far point in a thinly filled octant may cover more than the block: the rest of the block
    // most likely holds a nearer one, so that stays a guess as well
    if (need.x0 >= bx0 && need.x1 <= bx1 && need.y0 >= by0 && need.y1 <= by1 && need.z0 >= bz0 && need.z1 <= bz1) {
      box = need;
      S.guess = false;
    }
  }
  S.more = box.x0 < ox0 || box.x1 > ox1 || box.y0 < oy0 || box.y1 > oy1 || box.z0 < oz0 || box.z1 > oz1;
  if (S.more) {
    // the cells of `box` outside the octant: the far cell in x of the octant's rows, and the five
    // rows on the far side in y or z
    const int fy = cy - sy, fz = cz - sz;  // far rows (may lie outside the grid or the box)
    auto in_box = [&](int y, int z) { return y >= box.y0 && y <= box.y1 && z >= box.z0 && z <= box.z1; };
    const bool f0 = in_box(fy, cz - 1), f1 = in_box(fy, cz), f2 = in_box(fy, cz + 1), f3 = in_box(cy, fz),
               f4 = in_box(cy + sy, fz);
    const GridQuad q0 = load_quad(fy, cz - 1, f0), q1 = load_quad(fy, cz, f1), q2 = load_quad(fy, cz + 1, f2),
                   q3 = load_quad(cy, fz, f3), q4 = load_quad(cy + sy, fz, f4);
    int xa = box.x0, xb = box.x1;  // what is left of the octant's rows: beyond [ox0, ox1]
    if (sx < 0) xa = max(xa, ox1 + 1);
    else xb = min(xb, ox0 - 1);
    range(r00, xa, xb, in_box(cy, cz), S.seg_s[0], S.seg_e[0]);
    range(r10, xa, xb, y_on && in_box(cy + sy, cz), S.seg_s[1], S.seg_e[1]);
    range(r01, xa, xb, z_on && in_box(cy, cz + sz), S.seg_s[2], S.seg_e[2]);
    range(r11, xa, xb, y_on && z_on && in_box(cy + sy, cz + sz), S.seg_s[3], S.seg_e[3]);
    range(q0, box.x0, box.x1, f0, S.seg_s[4], S.seg_e[4]);
    range(q1, box.x0, box.x1, f1, S.seg_s[5], S.seg_e[5]);
    range(q2, box.x0, box.x1, f2, S.seg_s[6], S.seg_e[6]);
    range(q3, box.x0, box.x1, f3, S.seg_s[7], S.seg_e[7]);
    range(q4, box.x0, box.x1, f4, S.seg_s[8], S.seg_e[8]);
  }
}

__device__ __forceinline__ GridVerdict grid_nearest_end(const GridView &g, const float qx, const float qy, const float qz,
                                                        const float max_range_sq, GridSearch &S, float4 &best,
                                                        float &best_d, GridTrace *tr = nullptr) {
  if (S.early) return GRID_WALK;
  GridBest &b = S.b;
  GridBox &box = S.box;
  const float bound = S.bound;
  if (S.cold) {
    const int cx = grid_cell(qx, g.lo[0], g.inv_h, g.nx), cy = grid_cell(qy, g.lo[1], g.inv_h, g.ny),
              cz = grid_cell(qz, g.lo[2], g.inv_h, g.nz);
    const int bx0 = max(cx - 1, 0), bx1 = min(cx + 1, g.nx - 1), by0 = max(cy - 1, 0), by1 = min(cy + 1, g.ny - 1),
              bz0 = max(cz - 1, 0), bz1 = min(cz + 1, g.nz - 1);
    bool wide = false;
    bool guess5 = false;  // nothing in the 27 cells: the 125 around the query, checked afterwards
    if (S.guess) {  // the block was a guess: it must cover what was found in it
      if (b.d < 3.0e38f) {
        box = grid_cover(g, qx, qy, qz, fminf(b.d, bound));
        wide = box.x0 < bx0 || box.x1 > bx1 || box.y0 < by0 || box.y1 > by1 || box.z0 < bz0 || box.z1 > bz1;
      } else {
        box.x0 = max(cx - 2, 0); box.x1 = min(cx + 2, g.nx - 1);
        box.y0 = max(cy - 2, 0); box.y1 = min(cy + 2, g.ny - 1);
        box.z0 = max(cz - 2, 0); box.z1 = min(cz + 2, g.nz - 1);
        wide = guess5 = true;
      }
    }
    if (wide) {
      // ---- sparse spot: `box` (the cover of the best distance so far) reaches beyond the block.
      //      Scan all of it afresh, slab by slab; give up beyond 9 x 9 x 9 cells (kGridWide).
      if (box.x0 < cx - kGridWide || box.x1 > cx + kGridWide || box.y0 < cy - kGridWide || box.y1 > cy + kGridWide ||
          box.z0 < cz - kGridWide || box.z1 > cz + kGridWide) {
        PCGX_GRID_WHY(3);
        return GRID_WALK;
      }
      PCGX_GRID_WHY(7);
      b.p = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(-1));
      b.d = b.d2 = __builtin_inff();
      for (int z = box.z0; z <= box.z1; z++)
        for (int y0 = box.y0; y0 <= box.y1; y0 += 5) {  // up to 5 rows at a time: bounds in one round, points in one sequence
          uint32_t seg_s[5], seg_e[5];
#pragma unroll
          for (int j = 0; j < 5; j++) {
            const int y = y0 + j;
            const bool on = y <= box.y1;
            const int row = (z * g.ny + y) * g.nx;
            seg_s[j] = on ? g.start[row + box.x0] : 0u;
            seg_e[j] = on ? g.start[row + box.x1 + 1] : 0u;
          }
          grid_scan_segments<5>(g, seg_s, seg_e, qx, qy, qz, b, tr);
        }
      if (guess5) {  // the 125 cells must cover what was found in them
        if (!(b.d < 3.0e38f)) {
          PCGX_GRID_WHY(2);
          return GRID_WALK;
        }
        const GridBox need = grid_cover(g, qx, qy, qz, fminf(b.d, bound));
        if (need.x0 < box.x0 || need.x1 > box.x1 || need.y0 < box.y0 || need.y1 > box.y1 || need.z0 < box.z0 ||
            need.z1 > box.z1) {
          PCGX_GRID_WHY(3);
          return GRID_WALK;
        }
      }
    }
  }
  // every point with DistSq <= min(b.d, bound) was looked at
  if (b.d <= bound) {  // so b.d is the minimum over the whole tree
    if (b.d == max_range_sq) {  // a leaf is accepted there, a pivot is not (:100-103, :117)
      PCGX_GRID_WHY(4);
      return GRID_WALK;
    }
    if (b.tie()) {  // the winner depends on the visit order
      PCGX_GRID_WHY(5);
      return GRID_WALK;
    }
    best = b.p;
    best_d = b.d;
    return GRID_FOUND;
  }
  if (bound == max_range_sq) return GRID_NONE;  // nothing within maxRange^2
  PCGX_GRID_WHY(6);
  return GRID_WALK;  // ub promised a point the scan did not see: cannot happen, let the walk answer
}
#undef PCGX_GRID_WHY

__device__ __forceinline__ GridVerdict grid_nearest(const GridView &g, const float qx, const float qy, const float qz,
                                                    const float max_range_sq, const float ub, float4 &best,
                                                    float &best_d, GridTrace *tr = nullptr) {
  GridSearch S;
  grid_nearest_begin(g, qx, qy, qz, max_range_sq, ub, S, tr);
  if (S.more) grid_scan_segments<9>(g, S.seg_s, S.seg_e, qx, qy, qz, S.b, tr);
  return grid_nearest_end(g, qx, qy, qz, max_range_sq, S, best, best_d, tr);
}

// knn_grid.hip
pcgx_status grid_build(pcgx_kdtree *t, const float *d_xyz, const int32_t *d_labels, hipStream_t st);
void grid_free(pcgx_kdtree *t);
bool grid_enabled(const pcgx_kdtree *t);
pcgx_status grid_launch_nearest(const pcgx_kdtree *t, const float *d_q, const int32_t *d_perm, int64_t nq,
                                float max_range_sq, int32_t *d_ids, float *d_dsq, hipStream_t st);
// the same for a batch whose search order is the library's to choose: the queries are partitioned by a coarse cell
// once ({x, y, z, index} records) and searched in that order
pcgx_status grid_launch_nearest_partitioned(const pcgx_kdtree *t, const float *d_q, int64_t nq, float max_range_sq,
                                            int32_t *d_ids, float *d_dsq, hipStream_t st);

}  // namespace pcgx
