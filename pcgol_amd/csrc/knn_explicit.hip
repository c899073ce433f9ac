// knn_explicit.hip -- KDTree.DeletePoint with the reference's own tree.
//
// Reference: pc/storage/kdtree/kdtree.go:224-262 (findMinimumImpl), :264-320 (deleteNodeImpl),
// :322-332 (DeletePoint); searches :83-146 (Nearest / nearestImpl), :148-197 (Range / rangeImpl),
// :199-222 (searchLeafNode).
//
// DeletePoint patches the pointer tree: the deleted node takes over the id of the minimum (along
// its own axis) of its right subtree -- or of its left subtree, which then BECOMES the right one --
// and that minimum is deleted recursively.  The result is no longer the shape-by-N tree of the
// implicit BFS layout (pcgx_internal.h), and what Nearest returns on exact-distance ties or with
// MinDistSq > 0 depends on that shape.  So a handle that has seen deletions keeps a host mirror of
// the reference's nodes {id, dim, child0, child1}, patched exactly as deleteNodeImpl does it, an
// explicit device copy of it (32 B per node), and Nearest / Range walk THAT tree with the
// reference's visit order (one query per lane, explicit frames in LDS; no speculation -- this is the
// exact path for mutated trees, the fast path is the implicit tree).  ICP sessions on such a handle
// walk this patched tree too (icp_corr_xkernel, icp.hip; also when the deletion comes after the
// session was created); region growing uses the rebuilt canonical tree of resolve_tree (knn.hip):
// Range hits are a set, whatever the tree's shape.
#include <string.h>

#include <algorithm>
#include <atomic>
#include <functional>
#include <thread>
#include <vector>

#include "knn_xwalk.h"

namespace pcgx {

template <bool kMinDist>
__global__ __launch_bounds__(kXBlock) void xnearest_kernel(XTreeView xv, const float *__restrict__ q,
                                                           const int32_t *__restrict__ perm, int64_t nq,
                                                           float max_range_sq, float min_dist_sq,
                                                           int32_t *__restrict__ out_id, float *__restrict__ out_dsq,
                                                           int64_t guard) {
  extern __shared__ uint32_t s_stack[];
  const int64_t pos = (int64_t)blockIdx.x * kXBlock + threadIdx.x;
  if (pos >= nq) return;
  const int64_t i = perm ? (int64_t)perm[pos] : pos;
  const float qx = q[3 * i], qy = q[3 * i + 1], qz = q[3 * i + 2];
  int32_t best_id = -1;
  float best_d = max_range_sq;  // root == nil or nothing in range: {-1, maxRange^2} (kdtree.go:84-86,100-103)
  xwalk(
      xv, s_stack + threadIdx.x, kXBlock, qx, qy, qz, guard, [&]() { return best_d; },
      [&](const float4 &nd, float d) {  // leaf: replaces unless d > best (kdtree.go:95-103,138-139)
        if (!(d > best_d)) {
          best_id = __float_as_int(nd.w);
          best_d = d;
        }
        return !(kMinDist && best_d < min_dist_sq);  // :104-106 (checked on DistSq alone), :140-142
      },
      [&](const float4 &nd, float d) {  // pivot: strict < (kdtree.go:116-123)
        if (d < best_d) {
          best_id = __float_as_int(nd.w);
          best_d = d;
          if (kMinDist && best_d < min_dist_sq) return false;
        }
        return true;
      });
  out_id[i] = best_id;
  out_dsq[i] = best_d;
}

// same contract as range_kernel (range.hip)
template <bool kFill>
__global__ __launch_bounds__(kXBlock) void xrange_kernel(XTreeView xv, const float *__restrict__ q,
                                                         const int32_t *__restrict__ perm, int64_t nq, float bound,
                                                         int64_t *__restrict__ counts,
                                                         const int64_t *__restrict__ offsets, int64_t total,
                                                         int32_t *__restrict__ out_id, uint32_t *__restrict__ out_key,
                                                         uint32_t *__restrict__ out_query, int64_t guard) {
  extern __shared__ uint32_t s_stack[];
  const int64_t pos = (int64_t)blockIdx.x * kXBlock + threadIdx.x;
  if (pos >= nq) return;
  const int64_t i = perm ? (int64_t)perm[pos] : pos;
  const float qx = q[3 * i], qy = q[3 * i + 1], qz = q[3 * i + 2];
  int64_t found = 0;
  const int64_t out0 = kFill ? offsets[i] : 0;
  const int64_t cap = kFill ? offsets[i + 1] - out0 : 0;
  const bool slice_ok = kFill && out0 >= 0 && cap >= 0 && out0 + cap <= total;
  auto hit = [&](const float4 &nd, float d) {
    if (d < bound) {  // kdtree.go:166-169,178-181
      if (kFill && slice_ok && found < cap) {
        out_id[out0 + found] = __float_as_int(nd.w);
        out_key[out0 + found] = __float_as_uint(d);
        out_query[out0 + found] = (uint32_t)i;
      }
      ++found;
    }
    return true;
  };
  xwalk(xv, s_stack + threadIdx.x, kXBlock, qx, qy, qz, guard, [&]() { return bound; }, hit, hit);
  if (!kFill) counts[i] = found;
}

// The device copy out of the host mirror as it is ({id, child0, child1, dim} per node, uploaded in one piece) and the
// cloud's points by id (uploaded once per handle): node k's point record and links.  (The host used to put both arrays
// together point by point and copy them from pageable memory: 8-14 ms per refresh at 1M nodes.)
__global__ __launch_bounds__(256) void xtree_expand_kernel(const int4 *__restrict__ xnodes, const float *__restrict__ pts_by_id,
                                                           int64_t n, float4 *__restrict__ pts, int4 *__restrict__ links) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  const int4 nd = xnodes[k];  // {id, c0, c1, dim}
  const int64_t id = nd.x >= 0 && nd.x < n ? nd.x : 0;
  pts[k] = make_float4(pts_by_id[3 * id], pts_by_id[3 * id + 1], pts_by_id[3 * id + 2], __int_as_float(nd.x));
  links[k] = make_int4(nd.y, nd.z, nd.w, 0);
}

}  // namespace pcgx

using namespace pcgx;

// ---- the reference's tree on the host ------------------------------------------
namespace {

// kdtree.go:348-370 newNode on the already sorted in-order ids: node index = in-order position
int32_t build_xnodes(pcgx_kdtree *t, int64_t lo, int64_t cnt, int depth) {
  if (cnt <= 0) return -1;
  const int64_t half = cnt / 2, mid = lo + half;
  pcgx_kdtree::XNode &n = t->xnodes[(size_t)mid];
  n.id = t->inorder[(size_t)mid];
  n.dim = depth % 3;
  n.c0 = build_xnodes(t, lo, half, depth + 1);
  n.c1 = build_xnodes(t, mid + 1, cnt - half - 1, depth + 1);
  return (int32_t)mid;
}

inline float coord(const pcgx_kdtree *t, int32_t id, int dim) { return t->points[3 * (size_t)id + (size_t)dim]; }

// kdtree.go:224-262 findMinimumImpl: id with the smallest coordinate `dim` in the subtree, -1 if empty
int32_t find_minimum(const pcgx_kdtree *t, int32_t n, int dim) {
  if (n < 0) return -1;
  const pcgx_kdtree::XNode &nd = t->xnodes[(size_t)n];
  if (nd.dim == dim) {
    if (nd.c0 < 0) return nd.id;
    return find_minimum(t, nd.c0, dim);
  }
  const int32_t m0 = find_minimum(t, nd.c0, dim), m1 = find_minimum(t, nd.c1, dim);
  int32_t m = nd.id;  // minNode(dim, n.id, min0, min1): strict <, in this order (:234-241)
  if (m0 != -1 && coord(t, m0, dim) < coord(t, m, dim)) m = m0;
  if (m1 != -1 && coord(t, m1, dim) < coord(t, m, dim)) m = m1;
  return m;
}

// kdtree.go:264-320 deleteNodeImpl; returns the (possibly nil) node that replaces n
int32_t delete_node(pcgx_kdtree *t, int32_t n, int32_t pid) {
  if (n < 0) return -1;
  pcgx_kdtree::XNode &nd = t->xnodes[(size_t)n];
  if (pid == nd.id) {
    if (nd.c1 >= 0) {
      const int32_t m = find_minimum(t, nd.c1, nd.dim);
      const int32_t child = delete_node(t, nd.c1, m);
      nd.id = m;
      nd.c1 = child;
    } else if (nd.c0 >= 0) {
      const int32_t m = find_minimum(t, nd.c0, nd.dim);
      const int32_t child = delete_node(t, nd.c0, m);
      nd.id = m;
      nd.c0 = -1;
      nd.c1 = child;
    } else {
      return -1;
    }
    return n;
  }
  const float at = coord(t, nd.id, nd.dim), p = coord(t, pid, nd.dim);
  if (p <= at) nd.c0 = delete_node(t, nd.c0, pid);
  if (p >= at) t->xnodes[(size_t)n].c1 = delete_node(t, t->xnodes[(size_t)n].c1, pid);
  return n;
}

}  // namespace

static void xtree_init(pcgx_kdtree *t);

// Applies DeletePoint(pid) to the host mirror (created from the canonical tree on first use).
// Caller holds t->mu.
void xtree_delete(pcgx_kdtree *t, int64_t pid) {
  xtree_init(t);
  t->xroot = delete_node(t, t->xroot, (int32_t)pid);
  t->x_dirty = true;
}

namespace {
int host_threads() {
  static const int n = [] {
    int v = (int)std::thread::hardware_concurrency();
    if (const char *e = getenv("PCGX_HOST_THREADS")) v = atoi(e);
    return v < 1 ? 1 : (v > 16 ? 16 : v);
  }();
  return n;
}

// the original tree's nodes at depth `cut` (the tree's shape depends on n alone: node = middle of its in-order range)
struct SubTree {
  int64_t lo, cnt;  // in-order index range of the original subtree
  int32_t root;     // its root's node index (= lo + cnt / 2)
  int depth;
};
void collect_subtrees(int64_t lo, int64_t cnt, int depth, int cut, std::vector<SubTree> &out, std::vector<int32_t> *top) {
  if (cnt <= 0) return;
  const int64_t half = cnt / 2, mid = lo + half;
  if (depth == cut) {
    out.push_back(SubTree{lo, cnt, (int32_t)mid, depth});
    return;
  }
  if (top) top->push_back((int32_t)mid);
  collect_subtrees(lo, half, depth + 1, cut, out, top);
  collect_subtrees(mid + 1, cnt - half - 1, depth + 1, cut, out, top);
}
}  // namespace

static void xtree_init(pcgx_kdtree *t) {
  if (t->x_init) return;
  t->xnodes.resize((size_t)t->n);
  const int nth = host_threads();
  if (t->n >= (1 << 16) && nth > 1) {
    // the levels above depth 6 by this thread (it needs the subtrees' roots as children: their indices are arithmetic),
    // the 64 subtrees below by the host's threads: disjoint node ranges
    const int cut = 6;
    std::vector<SubTree> subs;
    collect_subtrees(0, t->n, 0, cut, subs, nullptr);
    std::vector<std::thread> th;
    std::atomic<size_t> next{0};
    for (int k = 0; k < nth; k++)
      th.emplace_back([&]() {
        for (size_t j = next++; j < subs.size(); j = next++) (void)build_xnodes(t, subs[j].lo, subs[j].cnt, subs[j].depth);
      });
    struct Top {
      static int32_t build(pcgx_kdtree *t, int64_t lo, int64_t cnt, int depth, int cut) {
        if (cnt <= 0) return -1;
        const int64_t half = cnt / 2, mid = lo + half;
        if (depth == cut) return (int32_t)mid;  // (a subtree's root: built by the threads)
        pcgx_kdtree::XNode &n = t->xnodes[(size_t)mid];
        n.id = t->inorder[(size_t)mid];
        n.dim = depth % 3;
        n.c0 = build(t, lo, half, depth + 1, cut);
        n.c1 = build(t, mid + 1, cnt - half - 1, depth + 1, cut);
        return (int32_t)mid;
      }
    };
    t->xroot = Top::build(t, 0, t->n, 0, cut);
    for (auto &x : th) x.join();
  } else {
    t->xroot = build_xnodes(t, 0, t->n, 0);
  }
  t->x_init = true;
  t->x_dirty = true;
}

// DeletePoint(ids[0]), DeletePoint(ids[1]), ... (kdtree.go:322-332) with the work shared out over the host's threads
// where the ORDER allows it.  A deletion touches the subtree below the node that holds the point (findMinimumImpl /
// deleteNodeImpl recurse downwards, kdtree.go:224-320) and only reads the nodes above it; points only ever move UP
// (a node takes over the minimum of a subtree below it).  So, with the tree cut at depth 6: deletions of points that sit
// in different depth-6 subtrees commute -- each subtree's deletions run in call order on one thread, the subtrees side
// by side -- and a deletion of a point that sits in one of the 63 nodes above the cut (a few per 100k) is carried out
// alone, in its place in the order, between two such batches.  Which subtree a point sits in: the one its ORIGINAL
// node lies in (in-order index ranges: nodes never move), unless it has been pulled up above the cut (the 63 ids there
// are looked up).  Caller holds t->mu.
void xtree_delete_batch(pcgx_kdtree *t, const int64_t *ids, int64_t m) {
  xtree_init(t);
  t->x_dirty = true;
  const int nth = host_threads();
  static const int64_t min_batch = getenv("PCGX_DELETE_PARALLEL_MIN") ? atoll(getenv("PCGX_DELETE_PARALLEL_MIN")) : 4096;
  if (nth < 2 || m < min_batch || t->n < (1 << 16)) {
    for (int64_t i = 0; i < m; i++) t->xroot = delete_node(t, t->xroot, (int32_t)ids[i]);
    return;
  }
  const int cut = 6;
  std::vector<SubTree> subs;
  std::vector<int32_t> top;
  collect_subtrees(0, t->n, 0, cut, subs, &top);  // (in-order: ascending lo)
  static_assert((1 << 6) < 255, "a subtree's number fits a byte");
  if (t->xsub.empty()) {  // the subtree of every id's original node
    t->xsub.assign((size_t)t->n, 255);
    std::atomic<size_t> next{0};
    auto fill = [&]() {
      for (size_t j = next++; j < subs.size(); j = next++)
        for (int64_t k = subs[j].lo; k < subs[j].lo + subs[j].cnt; k++) t->xsub[(size_t)t->inorder[(size_t)k]] = (uint8_t)j;
    };
    std::vector<std::thread> th;
    for (int k = 1; k < nth; k++) th.emplace_back(fill);
    fill();
    for (auto &x : th) x.join();
  }
  // The call's threads are started once and handed one piece of work after the other (a generation counter they poll:
  // the call is ten milliseconds long, starting fifteen threads per phase was a tenth of that).
  struct Pool {
    std::vector<std::thread> th;
    std::atomic<int> gen{0}, finished{0};
    std::atomic<bool> stop{false};
    std::function<void(int)> job;
    explicit Pool(int n) {
      for (int k = 1; k < n; k++)
        th.emplace_back([this, k]() {
          int seen = 0;
          for (;;) {
            while (gen.load(std::memory_order_acquire) == seen && !stop.load(std::memory_order_acquire)) std::this_thread::yield();
            if (stop.load(std::memory_order_acquire)) return;
            seen++;
            job(k);
            finished.fetch_add(1, std::memory_order_release);
          }
        });
    }
    void run(const std::function<void(int)> &f) {  // f(0) on the caller's thread, f(1 ..) on the others; returns when all are through
      job = f;
      finished.store(0, std::memory_order_relaxed);
      gen.fetch_add(1, std::memory_order_release);
      f(0);
      while (finished.load(std::memory_order_acquire) != (int)th.size()) std::this_thread::yield();
    }
    ~Pool() {
      stop.store(true, std::memory_order_release);
      for (auto &x : th) x.join();
    }
  } pool(nth);
  std::vector<uint8_t> is_above((size_t)t->n, 0);  // the ids that sit above the cut now
  auto top_ids = [&]() {  // the ids that sit above the cut NOW (nodes still attached)
    std::vector<int32_t> v;
    struct W {
      static void walk(const pcgx_kdtree *t, int32_t n, int depth, int cut, std::vector<int32_t> &v) {
        if (n < 0 || depth >= cut) return;
        v.push_back(t->xnodes[(size_t)n].id);
        walk(t, t->xnodes[(size_t)n].c0, depth + 1, cut, v);
        walk(t, t->xnodes[(size_t)n].c1, depth + 1, cut, v);
      }
    };
    W::walk(t, t->xroot, 0, cut, v);
    std::sort(v.begin(), v.end());
    return v;
  };
  std::vector<int32_t> above = top_ids();
  for (int32_t id : above) is_above[(size_t)id] = 1;
  // bucket[k][j]: thread k's share of a window of the ids, those of subtree j, in call order; the threads' shares one
  // behind the other are the window's order
  const size_t ns = subs.size();
  std::vector<std::vector<std::vector<int32_t>>> bucket((size_t)nth, std::vector<std::vector<int32_t>>(ns));
  std::vector<int64_t> first_top((size_t)nth);
  int n_flush = 0;
  auto flush = [&](int upto_thread) {  // the deletions bucketed by threads 0 .. upto_thread, subtree by subtree
    n_flush++;
    std::vector<size_t> work;
    for (size_t j = 0; j < ns; j++)
      for (int k = 0; k <= upto_thread; k++)
        if (!bucket[(size_t)k][j].empty()) {
          work.push_back(j);
          break;
        }
    if (work.empty()) return;
    std::vector<int32_t> new_root(ns);
    std::atomic<size_t> next{0};
    pool.run([&](int) {
      for (size_t w = next++; w < work.size(); w = next++) {
        const size_t j = work[w];
        int32_t r = subs[j].root;
        for (int k = 0; k <= upto_thread && r >= 0; k++)
          for (int32_t pid : bucket[(size_t)k][j]) {
            if (r < 0) break;  // (the subtree is gone: nothing left to find)
            r = delete_node(t, r, pid);
          }
        new_root[j] = r;
      }
    });
    // a subtree that went empty: its parent's link (deleteNodeImpl's `n.children[k] = child` with child == nil)
    for (size_t j : work) {
      if (new_root[j] >= 0) {
        continue;
      }
      const int32_t r = subs[j].root;
      if (t->xroot == r) t->xroot = -1;
      for (int32_t p : top) {
        pcgx_kdtree::XNode &nd = t->xnodes[(size_t)p];
        if (nd.c0 == r) nd.c0 = -1;
        if (nd.c1 == r) nd.c1 = -1;
      }
    }
  };
  // Windows of the ids, a slice per thread: every thread buckets its slice up to the first id that sits above the cut;
  // what lies in front of the FIRST such id of the window is flushed, that deletion is carried out alone, and the next
  // window begins behind it (the slices behind it were bucketed for nothing: a few times per call).
  const int64_t window = (int64_t)nth * 8192;
  for (int64_t w0 = 0; w0 < m;) {
    const int64_t w1 = std::min(m, w0 + window);
    pool.run([&](int k) {
      const int64_t a = w0 + (w1 - w0) * k / nth, b = w0 + (w1 - w0) * (k + 1) / nth;
      auto &mine = bucket[(size_t)k];
      for (auto &v : mine) v.clear();
      first_top[(size_t)k] = -1;
      for (int64_t i = a; i < b; i++) {
        const int32_t pid = (int32_t)ids[i];
        if (is_above[(size_t)pid]) {
          first_top[(size_t)k] = i;
          return;
        }
        // the subtree the point's original node lies in (255: an original node above the cut whose point has been
        // deleted already -- nothing to find)
        const uint8_t j = t->xsub[(size_t)pid];
        if (j != 255) mine[j].push_back(pid);
      }
    });
    int hit = -1;
    for (int k = 0; k < nth && hit < 0; k++)
      if (first_top[(size_t)k] >= 0) hit = k;
    if (hit < 0) {
      flush(nth - 1);
      w0 = w1;
      continue;
    }
    flush(hit);
    const int64_t i = first_top[(size_t)hit];
    t->xroot = delete_node(t, t->xroot, (int32_t)ids[i]);  // (reads and writes across the cut: alone)
    for (int32_t id : above) is_above[(size_t)id] = 0;
    above = top_ids();
    for (int32_t id : above) is_above[(size_t)id] = 1;
    w0 = i + 1;
  }
  if (getenv("PCGX_DELETE_TRACE"))
    fprintf(stderr, "delete batch: %lld ids, %d flushes, %d threads, %zu subtrees\n", (long long)m, n_flush, nth, ns);
}

namespace {
int64_t dump_rec(const pcgx_kdtree *t, int32_t n, int64_t *out, int64_t cap, int64_t *k) {
  if (n < 0) return -1;
  const int64_t me = (*k)++;
  const pcgx_kdtree::XNode &nd = t->xnodes[(size_t)n];
  const int64_t a = dump_rec(t, nd.c0, out, cap, k), b = dump_rec(t, nd.c1, out, cap, k);
  if (me < cap) {
    out[4 * me + 0] = nd.id;
    out[4 * me + 1] = nd.dim;
    out[4 * me + 2] = a;
    out[4 * me + 3] = b;
  }
  return me;
}
int depth_rec(const pcgx_kdtree *t, int32_t n, int d) {
  if (n < 0) return d;
  const pcgx_kdtree::XNode &nd = t->xnodes[(size_t)n];
  const int a = depth_rec(t, nd.c0, d + 1), b = depth_rec(t, nd.c1, d + 1);
  return a > b ? a : b;
}
}  // namespace

// The tree as the reference holds it (kdtree.go:25-29), pre-order: node k = {id, dim, index of
// child0, index of child1} (-1 = nil) -- after DeletePoint the patched tree.  *n_nodes = nodes in
// the tree; the first min(n_nodes, cap_nodes) are written.
extern "C" pcgx_status pcgx_kdtree_dump(const pcgx_kdtree *tc, int64_t *out4, int64_t cap_nodes, int64_t *n_nodes) {
  PCGX_API_LOCK();
  if (!tc || !n_nodes || cap_nodes < 0 || (cap_nodes > 0 && !out4)) return fail(PCGX_E_INVALID, "pcgx_kdtree_dump: bad argument");
  pcgx_kdtree *t = const_cast<pcgx_kdtree *>(tc);
  std::lock_guard<std::mutex> lock(t->mu);
  xtree_init(t);
  int64_t k = 0;
  dump_rec(t, t->xroot, out4, cap_nodes, &k);
  *n_nodes = k;
  return PCGX_OK;
}

// node.maxDepth(0) of the patched tree (kdtree.go:385-395); caller holds t->mu
int xtree_max_depth(const pcgx_kdtree *t) { return depth_rec(t, t->xroot, 0); }

void xtree_free(pcgx_kdtree *t) {
  dev_cache_free(t->d_xpts);
  dev_cache_free(t->d_xlinks);
  dev_cache_free(t->d_xsrc);
  t->d_xpts = nullptr;
  t->d_xlinks = nullptr;
  t->d_xsrc = nullptr;
}

namespace pcgx {
// Device copy of the patched tree (uploaded again after further deletions).
pcgx_status xtree_view(const pcgx_kdtree *tc, XTreeView *xv, hipStream_t st) {
  pcgx_kdtree *t = const_cast<pcgx_kdtree *>(tc);
  std::lock_guard<std::mutex> lock(t->mu);
  if (!t->x_init) return fail(PCGX_E_INVALID, "explicit tree requested for a handle without deletions");
  const size_t n = (size_t)t->n;
  if (!t->d_xpts) {
    hipError_t e = dev_cache_alloc((void **)&t->d_xpts, n * sizeof(float4));
    if (e == hipSuccess) e = dev_cache_alloc((void **)&t->d_xlinks, n * sizeof(int4));
    if (e != hipSuccess) return fail(PCGX_E_OOM, "explicit tree allocation failed: %s", hipGetErrorString(e));
    t->x_dirty = true;
  }
  if (t->x_dirty) {
    // uploaded in place: walks other streams still run on the previous copy must have finished
    dev_cache_quiesce();
    static_assert(sizeof(pcgx_kdtree::XNode) == sizeof(int4), "the mirror's nodes are uploaded as they are");
    if (!t->d_xsrc) {  // the cloud's points by id, once per handle
      PCGX_HIP_TRY(dev_cache_alloc((void **)&t->d_xsrc, n * 3 * sizeof(float)));
      PCGX_TRY(staged_upload(t->d_xsrc, t->points.data(), n * 3 * sizeof(float), st));
    }
    int4 *d_raw = nullptr;
    PCGX_HIP_TRY(dev_cache_alloc((void **)&d_raw, n * sizeof(int4)));
    pcgx_status rc = staged_upload(d_raw, t->xnodes.data(), n * sizeof(int4), st);
    if (rc == PCGX_OK) {
      hipLaunchKernelGGL(xtree_expand_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const int4 *)d_raw,
                         (const float *)t->d_xsrc, (int64_t)n, t->d_xpts, (int4 *)t->d_xlinks);
      if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) rc = fail(PCGX_E_HIP, "explicit tree: expansion failed");
    }
    dev_cache_free(d_raw);
    PCGX_TRY(rc);
    t->x_dirty = false;
  }
  xv->pts = t->d_xpts;
  xv->links = (const int4 *)t->d_xlinks;
  xv->root = t->xroot;
  xv->depth = t->depth;
  return PCGX_OK;
}
}  // namespace pcgx

pcgx_status xtree_launch_nearest(const pcgx_kdtree *t, const float *d_q, const int32_t *d_perm, int64_t nq,
                                 float max_range_sq, float min_dist_sq, int32_t *d_ids, float *d_dsq, hipStream_t st) {
  if (nq == 0) return PCGX_OK;
  XTreeView xv;
  PCGX_TRY(xtree_view(t, &xv, st));
  const size_t lds = (size_t)(xv.depth > 0 ? xv.depth : 1) * kXBlock * sizeof(uint32_t);
  const unsigned blocks = (unsigned)((nq + kXBlock - 1) / kXBlock);
  const int64_t guard = 4 * t->n + 8;  // a walk takes at most two steps per node
  ProfScope prof(PCGX_PROF_KNN_WALK, st);
  if (min_dist_sq > 0.0f)
    hipLaunchKernelGGL(xnearest_kernel<true>, dim3(blocks), dim3(kXBlock), lds, st, xv, d_q, d_perm, nq, max_range_sq,
                       min_dist_sq, d_ids, d_dsq, guard);
  else
    hipLaunchKernelGGL(xnearest_kernel<false>, dim3(blocks), dim3(kXBlock), lds, st, xv, d_q, d_perm, nq, max_range_sq,
                       min_dist_sq, d_ids, d_dsq, guard);
  PCGX_HIP_TRY(hipGetLastError());
  return PCGX_OK;
}

pcgx_status xtree_launch_range(const pcgx_kdtree *t, bool fill, const float *d_q, const int32_t *d_perm, int64_t nq,
                               float bound, int64_t *d_counts, const int64_t *d_offsets, int64_t total, int32_t *d_id,
                               uint32_t *d_key, uint32_t *d_query, hipStream_t st) {
  if (nq == 0) return PCGX_OK;
  XTreeView xv;
  PCGX_TRY(xtree_view(t, &xv, st));
  const size_t lds = (size_t)(xv.depth > 0 ? xv.depth : 1) * kXBlock * sizeof(uint32_t);
  const unsigned blocks = (unsigned)((nq + kXBlock - 1) / kXBlock);
  const int64_t guard = 4 * t->n + 8;
  if (fill)
    hipLaunchKernelGGL(xrange_kernel<true>, dim3(blocks), dim3(kXBlock), lds, st, xv, d_q, d_perm, nq, bound, d_counts,
                       d_offsets, total, d_id, d_key, d_query, guard);
  else
    hipLaunchKernelGGL(xrange_kernel<false>, dim3(blocks), dim3(kXBlock), lds, st, xv, d_q, d_perm, nq, bound, d_counts,
                       d_offsets, total, d_id, d_key, d_query, guard);
  PCGX_HIP_TRY(hipGetLastError());
  return PCGX_OK;
}

// ---- a few points at a time: the same walk on the host -------------------------------------------------------------
// storage.Search.Nearest / Range for ONE point (pc/storage/search.go:13-17; callers that loop: correspondence.go:25-36,
// regiongrowing.go:26,47) used to be a blocking GPU call each -- upload, launch, two PCIe round trips: 57-74 us against
// the 0.1-2.2 us the reference's own loop takes per point (profiles/r05e_rows.json).  The handle keeps a host mirror
// of the reference's nodes {id, child0, child1, dim} for DeletePoint (above) and the cloud's points by id; batches of
// up to host_walk_max() queries are answered from it by the walk of knn_xwalk.h restated for the host -- the same
// visits in the same order, the same float32 expressions (this file is compiled without contraction on both sides),
// so ids, DistSq bits, tie winners and the approximate search's answers are the device path's.  The mirror is made
// on the handle's first such call (host threads, ~1.5 ms at 1M points).  The GPU keeps every batch above the bound.
namespace {
template <class Bound, class Leaf, class Pivot>
inline void host_xwalk(const pcgx_kdtree *t, const float qv[3], Bound &&bound, Leaf &&on_leaf, Pivot &&on_pivot) {
  int32_t cur = t->xroot;
  if (cur < 0) return;  // root == nil (kdtree.go:84-86,150-152)
  constexpr int kFrames = 96;  // a frame per level: depth <= 27 (kMaxTreePoints); deletions never deepen the tree
  uint32_t stk[kFrames];
  int sp = 0;
  bool desc = true;
  const float *P = t->points.data();
  const pcgx_kdtree::XNode *X = t->xnodes.data();
  for (int64_t guard = 4 * t->n + 8; guard > 0; --guard) {
    if (desc) {  // searchLeafNode step (kdtree.go:202-221)
      const pcgx_kdtree::XNode &nd = X[cur];
      const float *p = P + 3 * (size_t)nd.id;
      if (nd.c0 < 0 && nd.c1 < 0) {
        const float dx = p[0] - qv[0], dy = p[1] - qv[1], dz = p[2] - qv[2];
        if (!on_leaf(nd.id, (dx * dx + dy * dy) + dz * dz)) return;
        desc = false;
        continue;
      }
      const int side = nd.c0 < 0 ? 1 : (nd.c1 < 0 ? 0 : (p[nd.dim] > qv[nd.dim] ? 0 : 1));  // pivotVal > val -> child0
      if (sp >= kFrames) return;
      stk[sp++] = (uint32_t)cur | ((uint32_t)side << 27);
      cur = side ? nd.c1 : nd.c0;
    } else {
      if (sp == 0) return;
      const uint32_t fw = stk[--sp];
      const pcgx_kdtree::XNode &nd = X[fw & 0x07FFFFFFu];
      const int side = (int)(fw >> 27);
      const float *p = P + 3 * (size_t)nd.id;
      const float fp = qv[nd.dim] - p[nd.dim];  // p[dim] - pivot[dim]
      if (fp * fp > bound()) continue;           // kdtree.go:111-115 / :173-177
      const float dx = p[0] - qv[0], dy = p[1] - qv[1], dz = p[2] - qv[2];
      if (!on_pivot(nd.id, (dx * dx + dy * dy) + dz * dz)) return;
      const int32_t other = side ? nd.c0 : nd.c1;  // the child that is not on the stack (:124-132)
      if (other >= 0) {
        cur = other;
        desc = true;
      }
    }
  }
}

void host_mirror(const pcgx_kdtree *tc) {
  pcgx_kdtree *t = const_cast<pcgx_kdtree *>(tc);
  std::lock_guard<std::mutex> lock(t->mu);
  xtree_init(t);
}
std::atomic<long long> g_host_walks{0};
}  // namespace

int64_t xtree_host_walk_max() {
  static const int64_t v = [] {
    if (const char *e = getenv("PCGX_HOST_WALK_MAX")) return (int64_t)atoll(e);
    return (int64_t)32;  // (one batched GPU call costs what ~40-60 host walks do: tests/perf_rows_ref.py)
  }();
  return v;
}
long long xtree_host_walks(bool reset) { return reset ? g_host_walks.exchange(0) : g_host_walks.load(); }

// Nearest (kdtree.go:83-146) of nq points, caller's arrays
void xtree_host_nearest(const pcgx_kdtree *t, const float *q, int64_t nq, float max_range, float min_dist_sq, int64_t *ids,
                        float *dist_sq) {
  host_mirror(t);
  g_host_walks += nq;
  const float max_range_sq = max_range * max_range;
  const bool cut = min_dist_sq > 0.0f;
  for (int64_t i = 0; i < nq; i++) {
    int32_t best_id = -1;
    float best_d = max_range_sq;  // nothing in range: {-1, maxRange^2} (kdtree.go:84-86,100-103)
    host_xwalk(
        t, q + 3 * i, [&]() { return best_d; },
        [&](int32_t id, float d) {  // leaf: replaces unless d > best (kdtree.go:95-103,138-139)
          if (!(d > best_d)) {
            best_id = id;
            best_d = d;
          }
          return !(cut && best_d < min_dist_sq);
        },
        [&](int32_t id, float d) {  // pivot: strict < (kdtree.go:116-123)
          if (d < best_d) {
            best_id = id;
            best_d = d;
            if (cut && best_d < min_dist_sq) return false;
          }
          return true;
        });
    ids[i] = best_id;
    dist_sq[i] = best_d;
  }
}

// Range (kdtree.go:148-197): counts[i] neighbours of point i; with offsets: filled in, a query's neighbours by DistSq,
// equal ones in the walk's order (what the device path leaves).  false: the offsets do not match the counts.
bool xtree_host_range(const pcgx_kdtree *t, const float *q, int64_t nq, float max_range, int64_t *counts, const int64_t *offsets,
                      int64_t *ids, float *dist_sq) {
  host_mirror(t);
  g_host_walks += nq;
  const float bound = max_range * max_range;
  std::vector<std::pair<float, int32_t>> found;
  for (int64_t i = 0; i < nq; i++) {
    found.clear();
    auto hit = [&](int32_t id, float d) {
      if (d < bound) found.emplace_back(d, id);  // kdtree.go:166-169,178-181
      return true;
    };
    host_xwalk(t, q + 3 * i, [&]() { return bound; }, hit, hit);
    if (counts) counts[i] = (int64_t)found.size();
    if (offsets) {
      if (offsets[i + 1] - offsets[i] != (int64_t)found.size()) return false;
      std::stable_sort(found.begin(), found.end(), [](const std::pair<float, int32_t> &a, const std::pair<float, int32_t> &b) { return a.first < b.first; });
      for (size_t k = 0; k < found.size(); k++) {
        ids[offsets[i] + (int64_t)k] = found[k].second;
        dist_sq[offsets[i] + (int64_t)k] = found[k].first;
      }
    }
  }
  return true;
}
