// kdtree_build.cpp -- host-side canonical KD-tree construction.
//
// Produces the in-order id sequence of the reference's tree
// (pc/storage/kdtree/kdtree.go:348-370 newNode): at depth d the sub-slice is
// sorted ascending by coordinate d%3 (indiceSorter.Less = strict <, :407-409),
// the element at len/2 becomes the node, the halves recurse.  After the
// recursion the slice, read left to right, is the in-order traversal of the
// tree and therefore defines it completely (see pcgx_internal.h).
//
// Go's sort.Sort is unstable, so the order of points with EQUAL split
// coordinates is not defined by the reference (and not pinned by its tests);
// this library defines it: stable with respect to the current order of the
// sub-slice.  Sorting moves whole {x,y,z,id} records (no gathers), with a
// merge sort that is safe for any comparator outcome (NaN coordinates).
#include <stdint.h>
#include <string.h>

#include <thread>
#include <vector>

namespace pcgx {

namespace {

struct Rec {
  float c[3];
  int32_t id;
};

// Stable merge sort of r[0..n) by c[dim]; tmp has n entries.
void sort_by_dim(Rec *r, Rec *tmp, int64_t n, int dim) {
  if (n < 2) return;
  constexpr int64_t kRun = 24;
  for (int64_t s = 0; s < n; s += kRun) {
    int64_t e = s + kRun < n ? s + kRun : n;
    for (int64_t i = s + 1; i < e; i++) {
      Rec v = r[i];
      int64_t j = i;
      while (j > s && v.c[dim] < r[j - 1].c[dim]) {
        r[j] = r[j - 1];
        j--;
      }
      r[j] = v;
    }
  }
  Rec *src = r, *dst = tmp;
  for (int64_t w = kRun; w < n; w *= 2) {
    for (int64_t s = 0; s < n; s += 2 * w) {
      int64_t m = s + w < n ? s + w : n;
      int64_t e = s + 2 * w < n ? s + 2 * w : n;
      int64_t a = s, b = m, o = s;
      while (a < m && b < e) {
        if (src[b].c[dim] < src[a].c[dim]) dst[o++] = src[b++];  // right only if strictly less
        else dst[o++] = src[a++];
      }
      while (a < m) dst[o++] = src[a++];
      while (b < e) dst[o++] = src[b++];
    }
    Rec *t = src;
    src = dst;
    dst = t;
  }
  if (src != r) memcpy(r, src, (size_t)n * sizeof(Rec));
}

void build_rec(Rec *r, Rec *tmp, int64_t n, int depth, int spawn_levels) {
  while (n > 1) {
    sort_by_dim(r, tmp, n, depth % 3);
    int64_t mid = n / 2;
    int64_t nr = n - mid - 1;
    if (spawn_levels > 0 && n > (1 << 14)) {
      std::thread th(build_rec, r, tmp, mid, depth + 1, spawn_levels - 1);
      build_rec(r + mid + 1, tmp + mid + 1, nr, depth + 1, spawn_levels - 1);
      th.join();
      return;
    }
    build_rec(r, tmp, mid, depth + 1, 0);
    // tail-iterate on the right half
    r += mid + 1;
    tmp += mid + 1;
    n = nr;
    depth++;
  }
}

}  // namespace

void build_inorder(const float *xyz, int64_t n, int32_t *inorder_ids) {
  std::vector<Rec> recs((size_t)n), tmp((size_t)n);
  for (int64_t i = 0; i < n; i++) {
    recs[i].c[0] = xyz[3 * i + 0];
    recs[i].c[1] = xyz[3 * i + 1];
    recs[i].c[2] = xyz[3 * i + 2];
    recs[i].id = (int32_t)i;
  }
  unsigned hw = std::thread::hardware_concurrency();
  int spawn = 0;
  while ((1u << spawn) < hw && spawn < 4) spawn++;
  build_rec(recs.data(), tmp.data(), n, 0, spawn);
  for (int64_t i = 0; i < n; i++) inorder_ids[i] = recs[i].id;
}

}  // namespace pcgx
