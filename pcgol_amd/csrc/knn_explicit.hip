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

static void xtree_init(pcgx_kdtree *t) {
  if (t->x_init) return;
  t->xnodes.resize((size_t)t->n);
  t->xroot = build_xnodes(t, 0, t->n, 0);
  t->x_init = true;
  t->x_dirty = true;
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
  t->d_xpts = nullptr;
  t->d_xlinks = nullptr;
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
    std::vector<float4> pts(n);
    std::vector<int4> links(n);
    for (size_t k = 0; k < n; k++) {
      const pcgx_kdtree::XNode &nd = t->xnodes[k];
      pts[k] = make_float4(t->points[3 * (size_t)nd.id], t->points[3 * (size_t)nd.id + 1], t->points[3 * (size_t)nd.id + 2],
                           __builtin_bit_cast(float, nd.id));
      links[k] = make_int4(nd.c0, nd.c1, nd.dim, 0);
    }
    PCGX_HIP_TRY(hipMemcpyAsync(t->d_xpts, pts.data(), n * sizeof(float4), hipMemcpyHostToDevice, st));
    PCGX_HIP_TRY(hipMemcpyAsync(t->d_xlinks, links.data(), n * sizeof(int4), hipMemcpyHostToDevice, st));
    PCGX_HIP_TRY(hipStreamSynchronize(st));  // host temporaries
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
