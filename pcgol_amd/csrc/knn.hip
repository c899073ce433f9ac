// knn.hip -- batched KDTree.Nearest on gfx950 + the pcgx_kdtree_* C ABI.
// Reference: pc/storage/kdtree/kdtree.go (New :33-56, Nearest :83-146).
#include <algorithm>
#include <stdlib.h>
#include <string.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "knn_grid.h"
#include "knn_walk.h"

namespace pcgx {

// Waves pull chunks of 64 launch positions (walk_queries) until the batch is done.
// Queries are packed xyz (AoS, 12 B).  `perm` (optional) maps the launch position to
// the query index (Morton order, sort.hip): results are written at the original
// index, so the permutation is invisible to the caller.
template <bool kMinDist, bool kStats = false>
__global__ __launch_bounds__(kKnnBlock) void nearest_kernel(TreeView tv, const float *__restrict__ q,
                                                            const int32_t *__restrict__ perm,
                                                            int64_t nq, float max_range_sq, float min_dist_sq,
                                                            int32_t *__restrict__ out_id,
                                                            float *__restrict__ out_dsq,
                                                            unsigned long long *__restrict__ stats = nullptr,
                                                            const float *__restrict__ hint = nullptr,
                                                            uint32_t *__restrict__ leaf_io = nullptr,
                                                            const uint32_t *__restrict__ nq_dev = nullptr) {
  extern __shared__ uint32_t s_stack[];
  __shared__ uint32_t s_next_chunk;
  if (nq_dev) {  // the queries perm[0 .. *nq_dev): what the grid pass left for the walk (knn_grid.hip)
    nq = (int64_t)*nq_dev;
    if (nq == 0) return;  // uniform
  }
  uint32_t *queue = s_stack + (size_t)(tv.depth > 1 ? tv.depth - 1 : 1) * kKnnBlock +
                    (threadIdx.x >> 6) * (kWalkQueueBytesPerWave / 4);
  float *top = reinterpret_cast<float *>(s_stack + (size_t)(tv.depth > 1 ? tv.depth - 1 : 1) * kKnnBlock +
                                         (kKnnBlock / 64) * (kWalkQueueBytesPerWave / 4));
  load_top_levels(tv, top);
  uint32_t chunk_begin, chunk_end;
  block_chunk_range(nq, blockIdx.x, gridDim.x, chunk_begin, chunk_end);
  if (threadIdx.x == 0) s_next_chunk = chunk_begin;
  __syncthreads();
  walk_queries<kMinDist, kStats>(
      tv, s_stack + threadIdx.x, kKnnBlock, queue, top, nq, &s_next_chunk, chunk_end, (int64_t)chunk_begin * 64,
      max_range_sq, min_dist_sq,
      [&](int64_t pos, float &x, float &y, float &z, float &ub, uint32_t &pred) {
        pred = (kStats && leaf_io) ? leaf_io[perm ? (int64_t)perm[pos] : pos] : 0u;
        const int64_t i = perm ? (int64_t)perm[pos] : pos;
        x = q[3 * i + 0];
        y = q[3 * i + 1];
        z = q[3 * i + 2];
        ub = __builtin_inff();
        if (kStats && hint) {  // instrumented runs only: a tree point per query (see pcgx_debug_walk_stats)
          const float dx = hint[3 * i + 0] - x, dy = hint[3 * i + 1] - y, dz = hint[3 * i + 2] - z;
          ub = (dx * dx + dy * dy) + dz * dz;
        }
      },
      [&](int64_t pos, const float4 &best, float best_d) {
        const int64_t i = perm ? (int64_t)perm[pos] : pos;
        out_id[i] = __float_as_int(best.w);
        out_dsq[i] = best_d;
      },
      [&](int64_t pos, uint32_t leaf) {
        if (kStats && leaf_io) leaf_io[perm ? (int64_t)perm[pos] : pos] = leaf;
      },
      stats);
}

// Leaf directory: one thread per grid cell descends from the root with the cell's centre
// (the reference's searchLeafNode rule, kdtree.go:202-221) and records the leaf reached.
// Nearest on a tree without nodes (root == nil, kdtree.go:84-86): {-1, maxRange^2} for every query
__global__ __launch_bounds__(256) void nearest_empty_kernel(int64_t nq, float max_range_sq, int32_t *__restrict__ ids,
                                                            float *__restrict__ dsq) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < nq) {
    ids[i] = -1;
    dsq[i] = max_range_sq;
  }
}

__global__ __launch_bounds__(256) void dir_build_kernel(TreeView tv, uint32_t *__restrict__ dir) {
  const int g = tv.dir_bits;
  const uint32_t cells = 1u << (3 * g);
  const uint32_t cell = blockIdx.x * 256u + threadIdx.x;
  if (cell >= cells) return;
  const uint32_t mask = (1u << g) - 1u;
  const uint32_t c[3] = {cell & mask, (cell >> g) & mask, cell >> (2 * g)};
  float p[3];
  for (int k = 0; k < 3; k++)
    p[k] = tv.dir_scale[k] > 0.0f ? tv.dir_lo[k] + ((float)c[k] + 0.5f) / tv.dir_scale[k] : tv.dir_lo[k];
  uint32_t b = 1u, n = (uint32_t)tv.n;
  int depth = 0;
  while (n > 1u) {
    const int dim = depth % 3;
    const float pv = node_comp(tv.nodes, b, dim);
    const bool left = n == 2u || pv > p[dim];
    const uint32_t half = n >> 1;
    n = left ? half : n - half - 1u;
    b = 2u * b + (left ? 0u : 1u);
    depth++;
  }
  dir[cell] = b;
}

// Idle lanes a wave tolerates before it runs the (divergent) emit + refill section.
int walk_refill_threshold() {
  static int v = -1;
  if (v < 0) {
    v = 16;
    if (const char *e = getenv("PCGX_WALK_REFILL")) {
      const int t = atoi(e);
      if (t >= 1 && t <= 64) v = t;
    }
  }
  return v;
}

// Levels the preparation of a chunk follows the real descent below a wrong prediction, all such
// lanes in lockstep, before it leaves the rest to the stepping loop (knn_walk.h).
int walk_tight_levels() {
  static int v = -1;
  if (v < 0) {
    v = 0;
    if (const char *e = getenv("PCGX_WALK_TIGHT")) {
      const int t = atoi(e);
      if (t >= 0 && t <= 32) v = t;
    }
  }
  return v;
}

// Chunks a wave may prepare in one refill section while its queue cannot serve every waiting lane.
int walk_chunks_per_refill() {
  static int v = -1;
  if (v < 0) {
    v = 1;
    if (const char *e = getenv("PCGX_WALK_CHUNKS_PER_REFILL")) {
      const int t = atoi(e);
      if (t >= 1 && t <= 64) v = t;
    }
  }
  return v;
}

// Grid oversubscription of the walk kernels: workgroups launched per resident slot.  Ranges are
// static per workgroup, so launching more workgroups than fit lets the dispatcher even out the
// differences between ranges.
int walk_oversubscribe() {
  static int v = -1;
  if (v < 0) {
    v = 1;
    if (const char *e = getenv("PCGX_WALK_OVERSUB")) {
      const int t = atoi(e);
      if (t >= 1 && t <= 16) v = t;
    }
  }
  return v;
}

static int g_walk_resident = 4;

// Blocks resident per CU for the walk kernels, limited by their LDS (walk_lds_bytes).
int walk_blocks_per_cu(const TreeView &tv) {
  const size_t lds = walk_lds_bytes(tv, kKnnBlock);
  int b = (int)((160 * 1024) / (lds ? lds : 1));
  if (b > 8) b = 8;
  if (b < 1) b = 1;
  g_walk_resident = b;
  if (const char *e = getenv("PCGX_WALK_BLOCKS_PER_CU")) {
    int v = atoi(e);
    if (v >= 1 && v <= 8) b = v;
  }
  return b;
}

pcgx_status launch_nearest(const TreeView &tv, const float *d_q, const int32_t *d_perm, int64_t nq,
                           float max_range_sq, float min_dist_sq, int32_t *d_ids, float *d_dsq,
                           hipStream_t st) {
  if (nq == 0) return PCGX_OK;
  const size_t lds = walk_lds_bytes(tv, kKnnBlock);
  int64_t blocks = (int64_t)ctx().num_cu * walk_blocks_per_cu(tv) * walk_oversubscribe();
  const int64_t max_blocks = (nq + kKnnBlock - 1) / kKnnBlock;
  if (blocks > max_blocks) blocks = max_blocks;
  if (blocks >= 8) blocks &= ~(int64_t)7;  // multiple of 8: see block_chunk_range
  ProfScope prof(PCGX_PROF_KNN_WALK, st);
  // `x < MinDistSq` can only hold for MinDistSq > 0 (or NaN distances, which compare false).
  if (min_dist_sq > 0.0f)
    hipLaunchKernelGGL(nearest_kernel<true>, dim3((unsigned)blocks), dim3(kKnnBlock), lds, st, tv, d_q,
                       d_perm, nq, max_range_sq, min_dist_sq, d_ids, d_dsq);
  else
    hipLaunchKernelGGL(nearest_kernel<false>, dim3((unsigned)blocks), dim3(kKnnBlock), lds, st, tv, d_q,
                       d_perm, nq, max_range_sq, min_dist_sq, d_ids, d_dsq);
  PCGX_HIP_TRY(hipGetLastError());
  return PCGX_OK;
}

pcgx_status launch_nearest_listed(const TreeView &tv, const float *d_q, const int32_t *d_list,
                                  const uint32_t *d_count, int64_t nq_max, float max_range_sq, int32_t *d_ids,
                                  float *d_dsq, hipStream_t st) {
  if (nq_max == 0) return PCGX_OK;
  const size_t lds = walk_lds_bytes(tv, kKnnBlock);
  int64_t blocks = (int64_t)ctx().num_cu * walk_blocks_per_cu(tv) * walk_oversubscribe();
  const int64_t max_blocks = (nq_max + kKnnBlock - 1) / kKnnBlock;
  if (blocks > max_blocks) blocks = max_blocks;
  if (blocks >= 8) blocks &= ~(int64_t)7;
  ProfScope prof(PCGX_PROF_KNN_WALK, st);
  hipLaunchKernelGGL(nearest_kernel<false>, dim3((unsigned)blocks), dim3(kKnnBlock), lds, st, tv, d_q, d_list, nq_max,
                     max_range_sq, 0.0f, d_ids, d_dsq, (unsigned long long *)nullptr, (const float *)nullptr,
                     (uint32_t *)nullptr, d_count);
  PCGX_HIP_TRY(hipGetLastError());
  return PCGX_OK;
}

}  // namespace pcgx

using namespace pcgx;

namespace {
// Device staging of a host-pointer call: bump-allocated from the grow-only host arena (no
// hipMalloc / hipFree in steady state; hipFree alone costs more than the PCIe copies of a 1M batch).
__global__ __launch_bounds__(256) void widen_ids_kernel(const int32_t *__restrict__ in, int64_t n, int64_t *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = (int64_t)in[i];
}

struct HostCallBufs {
  float *q = nullptr;
  int32_t *ids = nullptr;
  float *dsq = nullptr;
  pcgx_status alloc(int64_t nq, hipStream_t st) {
    Arena &ar = ctx().host_arena;
    PCGX_TRY(ar.begin(st));
    PCGX_TRY(ar.alloc_n((size_t)nq * 3, &q));
    PCGX_TRY(ar.alloc_n((size_t)nq, &ids));
    PCGX_TRY(ar.alloc_n((size_t)nq, &dsq));
    return PCGX_OK;
  }
};
}  // namespace

// In-order sequence -> BFS slots: the node of range [lo, lo+cnt) is element lo + cnt/2
// (kdtree.go:355-364); children 2b (left, cnt/2 points) and 2b+1 (right).
static void fill_bfs(const pcgx_kdtree &t, const int32_t *labels, std::vector<float4> &nodes, size_t b, int64_t lo,
                     int64_t cnt) {
  while (cnt > 0) {
    const int64_t half = cnt / 2, mid = lo + half;
    const int32_t id = t.inorder[mid];
    const int32_t label = labels ? labels[id] : id;
    nodes[b] = make_float4(t.points[3 * (int64_t)id], t.points[3 * (int64_t)id + 1],
                           t.points[3 * (int64_t)id + 2], __builtin_bit_cast(float, label));
    if (half > 0) fill_bfs(t, labels, nodes, 2 * b, lo, half);
    lo = mid + 1;  // iterate into the right child
    cnt = cnt - half - 1;
    b = 2 * b + 1;
  }
}

// ------------------------------------------------------------------ C ABI

// labels (optional, host, [n]): the id each point's node reports (default: its index).
static pcgx_status build_tree(const void *data, int64_t n, int32_t stride, int32_t xyz_off, const int32_t *labels,
                              pcgx_kdtree **out) {
  if (!out) return fail(PCGX_E_INVALID, "pcgx_kdtree_build: out is NULL");
  *out = nullptr;
  if (n < 0 || (n > 0 && !data)) return fail(PCGX_E_INVALID, "pcgx_kdtree_build: bad data/n");
  if (n == 0) return fail(PCGX_E_NO_POINT, "pcgx_kdtree_build: empty cloud (kdtree.New panics in the reference)");
  if (stride < 12 || xyz_off < 0 || xyz_off + 12 > stride)
    return fail(PCGX_E_BAD_FIELD, "pcgx_kdtree_build: stride %d / xyz offset %d do not hold an xyz triple", stride, xyz_off);
  if (n > kMaxTreePoints) return fail(PCGX_E_TOO_LARGE, "pcgx_kdtree_build: %lld points > 2^26", (long long)n);
  PCGX_TRY(ensure_init());
  pcgx_kdtree *t = new pcgx_kdtree();
  t->n = n;
  t->depth = tree_depth(n);
  const uint8_t *src = (const uint8_t *)data;
  const size_t slots = (size_t)1 << t->depth;
  hipError_t e = dev_cache_alloc((void **)&t->d_nodes, slots * sizeof(float4));
  if (e != hipSuccess) {
    delete t;
    return fail(PCGX_E_OOM, "hipMalloc for %lld tree nodes failed: %s", (long long)n, hipGetErrorString(e));
  }
  // Build on the device for large clouds; NaN coordinates (no consistent order under <) and
  // small clouds take the host build.  PCGX_BUILD=host|gpu forces one (tests).
  bool gpu_wanted = n >= 32768;
  if (const char *f = getenv("PCGX_BUILD")) {
    if (!strcmp(f, "host")) gpu_wanted = false;
    if (!strcmp(f, "gpu") && n >= 2) gpu_wanted = true;
  }
  // A packed cloud goes to the device straight from the caller's buffer and the device build is under way while
  // the host makes its own copy and looks for the bounding box and for NaNs (1 ms at 1M points); a NaN -- the
  // device build has no consistent order for one -- and the device's work is dropped.
  const bool early = gpu_wanted && stride == 12 && xyz_off == 0;
  hipStream_t st = ctx().stream;
  Arena &ar = ctx().arena;
  float *d_xyz = nullptr;
  uint32_t *d_order = nullptr;
  int32_t *d_labels = nullptr;
  auto start_device_build = [&](const void *host_xyz) -> pcgx_status {
    pcgx_status rc = ar.begin(st);
    if (rc == PCGX_OK) rc = ar.alloc_n((size_t)n * 3, &d_xyz);
    if (rc == PCGX_OK) rc = ar.alloc_n((size_t)n, &d_order);
    if (rc == PCGX_OK && labels) rc = ar.alloc_n((size_t)n, &d_labels);
    if (rc == PCGX_OK) {
      hipError_t e2 = hipMemcpyAsync(d_xyz, host_xyz, (size_t)n * 12, hipMemcpyHostToDevice, st);
      if (e2 == hipSuccess && labels) e2 = hipMemcpyAsync(d_labels, labels, (size_t)n * 4, hipMemcpyHostToDevice, st);
      if (e2 != hipSuccess) rc = fail(PCGX_E_HIP, "tree upload failed: %s", hipGetErrorString(e2));
    }
    if (rc == PCGX_OK) rc = build_tree_device(d_xyz, n, t->depth, d_order, t->d_nodes, d_labels, st);
    return rc;
  };
  if (early) {
    const pcgx_status rc = start_device_build(src);
    if (rc != PCGX_OK) {
      (void)hipStreamSynchronize(st);
      dev_cache_free(t->d_nodes);
      delete t;
      return rc;
    }
  }
  if (stride == 12 && xyz_off == 0) {
    t->points.assign(reinterpret_cast<const float *>(src), reinterpret_cast<const float *>(src) + (size_t)n * 3);
  } else {
    t->points.resize((size_t)n * 3);
    for (int64_t i = 0; i < n; i++) memcpy(&t->points[3 * i], src + i * (int64_t)stride + xyz_off, 12);
  }
  // one pass: bounding box (first point, then strict < / >, so NaNs never replace a bound) and NaN scan
  float blo[3], bhi[3];
  bool has_nan = false;
  {
    const float *pp = t->points.data();
    float l0 = pp[0], l1 = pp[1], l2 = pp[2], h0 = l0, h1 = l1, h2 = l2;
    int nan_count = 0;
    for (int64_t i = 0; i < n; i++) {
      const float x = pp[3 * i], y = pp[3 * i + 1], z = pp[3 * i + 2];
      l0 = x < l0 ? x : l0; h0 = x > h0 ? x : h0;
      l1 = y < l1 ? y : l1; h1 = y > h1 ? y : h1;
      l2 = z < l2 ? z : l2; h2 = z > h2 ? z : h2;
      nan_count += (x != x) | (y != y) | (z != z);
    }
    blo[0] = l0; blo[1] = l1; blo[2] = l2;
    bhi[0] = h0; bhi[1] = h1; bhi[2] = h2;
    has_nan = nan_count != 0;
    t->has_nan = has_nan;
    // Do coordinates repeat?  (up to 1024 points, evenly spaced: of each axis' values, how many are different.)  A walk of
    // the reference's kind rules a sub-tree out by its distance from a split plane; where the points of both sides lie ON
    // the plane -- a ground plane's z, a lattice's columns -- it rules out nothing, and a 16384-point Fit's walks are ten
    // times a random cloud's.  The one-launch Fit (icp_small.hip) takes such clouds up to sizes where it would not pay else.
    if (n >= 64 && n <= 65535 && !has_nan) {
      const int64_t m = n < 1024 ? n : 1024, step = n / m;
      std::vector<float> v((size_t)m);
      for (int k = 0; k < 3 && !t->many_ties; k++) {
        for (int64_t j = 0; j < m; j++) v[(size_t)j] = pp[3 * (j * step) + k];
        std::sort(v.begin(), v.end());
        const int64_t distinct = (int64_t)(std::unique(v.begin(), v.end()) - v.begin());
        if (2 * distinct <= m) t->many_ties = true;
      }
    }
    for (int k = 0; k < 3; k++) { t->bbox_lo[k] = blo[k]; t->bbox_hi[k] = bhi[k]; }
  }
  t->inorder.resize((size_t)n);
  const bool on_gpu = gpu_wanted && !has_nan;
  if (early && !on_gpu) (void)hipStreamSynchronize(st);  // (a NaN: the host builds, what the device did is overwritten)
  if (on_gpu) {
    pcgx_status rc = early ? PCGX_OK : start_device_build(t->points.data());
    // (has_nan is false here.  A tree rebuilt over the points left after DeletePoint -- labels -- only
    // serves region growing's Range walks: no grid for it)
    if (rc == PCGX_OK && !labels) rc = grid_build(t, d_xyz, d_labels, st);
    if (rc == PCGX_OK) {
      e = hipMemcpyAsync(t->inorder.data(), d_order, (size_t)n * 4, hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = hipStreamSynchronize(st);
      if (e != hipSuccess) rc = fail(PCGX_E_HIP, "tree build failed: %s", hipGetErrorString(e));
    }
    if (rc != PCGX_OK) {
      grid_free(t);
      dev_cache_free(t->d_nodes);
      delete t;
      return rc;
    }
  } else {
    build_inorder(t->points.data(), n, t->inorder.data());
    std::vector<float4> nodes(slots, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
    fill_bfs(*t, labels, nodes, 1, 0, n);
    e = hipMemcpy(t->d_nodes, nodes.data(), slots * sizeof(float4), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      dev_cache_free(t->d_nodes);
      delete t;
      return fail(PCGX_E_HIP, "tree upload failed: %s", hipGetErrorString(e));
    }
    if (!has_nan && !labels) {  // the grid of the certified fast path (knn_grid.h)
      pcgx_status rc = ar.begin(st);
      if (rc == PCGX_OK) rc = ar.alloc_n((size_t)n * 3, &d_xyz);
      if (rc == PCGX_OK && labels) rc = ar.alloc_n((size_t)n, &d_labels);
      if (rc == PCGX_OK) {
        e = hipMemcpyAsync(d_xyz, t->points.data(), (size_t)n * 12, hipMemcpyHostToDevice, st);
        if (e == hipSuccess && labels) e = hipMemcpyAsync(d_labels, labels, (size_t)n * 4, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) rc = fail(PCGX_E_HIP, "grid upload failed: %s", hipGetErrorString(e));
      }
      if (rc == PCGX_OK) rc = grid_build(t, d_xyz, d_labels, st);
      if (rc == PCGX_OK && hipStreamSynchronize(st) != hipSuccess) rc = fail(PCGX_E_HIP, "grid build failed");
      if (rc != PCGX_OK) {
        grid_free(t);
        dev_cache_free(t->d_nodes);
        delete t;
        return rc;
      }
    }
  }
  // leaf directory: ~2 cells per point, at most 2^27 cells
  {
    const float *lo = blo, *hi = bhi;
    for (int k = 0; k < 3; k++) { t->bbox_lo[k] = lo[k]; t->bbox_hi[k] = hi[k]; }
    int g = 0;
    while (g < 9 && ((int64_t)1 << (3 * g)) < 2 * n) g++;
    if (const char *e = getenv("PCGX_DIR_BITS")) {  // tuning knob: cells per axis = 2^g
      const int v = atoi(e);
      if (v >= 0 && v <= 9) g = v;
    }
    t->dir_bits = g;
    for (int k = 0; k < 3; k++) {
      const float ext = hi[k] - lo[k];
      t->dir_lo[k] = lo[k];
      t->dir_scale[k] = (ext > 0.0f && ext == ext && ext < 3.0e38f) ? (float)(1 << g) / ext : 0.0f;
      if (!(lo[k] == lo[k])) { t->dir_lo[k] = 0.0f; t->dir_scale[k] = 0.0f; }  // NaN coordinates
    }
    const size_t cells = (size_t)1 << (3 * g);
    e = dev_cache_alloc((void **)&t->d_dir, cells * sizeof(uint32_t));
    if (e != hipSuccess) {
      grid_free(t);
      dev_cache_free(t->d_nodes);
      delete t;
      return fail(PCGX_E_OOM, "hipMalloc for the leaf directory (%zu cells) failed: %s", cells, hipGetErrorString(e));
    }
    hipLaunchKernelGGL(dir_build_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, ctx().stream,
                       t->view(), t->d_dir);
    e = hipStreamSynchronize(ctx().stream);
    if (e != hipSuccess) {
      grid_free(t);
      dev_cache_free(t->d_nodes);
      dev_cache_free(t->d_dir);
      delete t;
      return fail(PCGX_E_HIP, "leaf directory build failed: %s", hipGetErrorString(e));
    }
  }
  *out = t;
  return PCGX_OK;
}

extern "C" pcgx_status pcgx_kdtree_build(const void *data, int64_t n, int32_t stride,
                                         int32_t xyz_off, pcgx_kdtree **out) {
  PCGX_API_CALL();
  return build_tree(data, n, stride, xyz_off, nullptr, out);
}

extern "C" pcgx_status pcgx_kdtree_free(pcgx_kdtree *t) {
  PCGX_API_LOCK();
  if (!t) return PCGX_OK;
  // the buffers go back to the block cache and may be handed out again at once: kernels a caller
  // enqueued on a stream of its own (NearestBatchDev, ICP steps) must be done with them
  dev_cache_quiesce();
  if (t->live) pcgx_kdtree_free(t->live);
  for (pcgx_kdtree *r : t->retired) pcgx_kdtree_free(r);
  xtree_free(t);
  grid_free(t);
  if (t->d_nodes) dev_cache_free(t->d_nodes);
  if (t->d_dir) dev_cache_free(t->d_dir);
  if (t->d_inv) dev_cache_free(t->d_inv);
  delete t;
  return PCGX_OK;
}

// KDTree.DeletePoint (kdtree.go:322-332) for a batch of ids.  An id outside [0, Len()) is the
// reference's error (:323-325) and nothing is deleted; deleting a point twice is a no-op, as in the
// reference (kdtree_test.go "TwiceTheSamePoint").  The tree over the remaining points is rebuilt
// lazily by the next query (resolve_tree).
extern "C" pcgx_status pcgx_kdtree_delete_points(pcgx_kdtree *t, const int64_t *ids, int64_t m) {
  PCGX_API_LOCK();
  if (!t || m < 0 || (m > 0 && !ids)) return fail(PCGX_E_INVALID, "pcgx_kdtree_delete_points: bad argument");
  for (int64_t i = 0; i < m; i++)
    if (ids[i] < 0 || ids[i] > t->n - 1)
      return fail(PCGX_E_OUT_OF_RANGE, "%lld does not correspond to any point in the tree", (long long)ids[i]);
  std::lock_guard<std::mutex> lock(t->mu);
  if (t->deleted.empty()) t->deleted.assign((size_t)t->n, 0);
  xtree_delete_batch(t, ids, m);  // the reference's patching, in call order (a repeated id finds nothing)
  for (int64_t i = 0; i < m; i++) {
    if (t->deleted[(size_t)ids[i]]) continue;
    t->deleted[(size_t)ids[i]] = 1;
    t->n_deleted++;
    t->dirty = true;
  }
  return PCGX_OK;
}

extern "C" pcgx_status pcgx_kdtree_live_count(const pcgx_kdtree *t, int64_t *n_live) {
  PCGX_API_LOCK();
  if (!t || !n_live) return fail(PCGX_E_INVALID, "pcgx_kdtree_live_count: NULL argument");
  *n_live = t->n - t->n_deleted;
  return PCGX_OK;
}

pcgx_status resolve_tree(const pcgx_kdtree *tc, const pcgx_kdtree **active, bool *empty) {
  *empty = false;
  *active = tc;
  if (!tc || tc->n_deleted == 0) return PCGX_OK;
  pcgx_kdtree *t = const_cast<pcgx_kdtree *>(tc);  // deletion state is the handle's own, guarded by mu
  std::lock_guard<std::mutex> lock(t->mu);
  if (t->dirty) {
    const int64_t n_live = t->n - t->n_deleted;
    pcgx_kdtree *nt = nullptr;
    if (n_live > 0) {
      std::vector<float> xyz((size_t)n_live * 3);
      std::vector<int32_t> labels((size_t)n_live);
      int64_t k = 0;
      for (int64_t i = 0; i < t->n; i++) {
        if (t->deleted[(size_t)i]) continue;
        memcpy(&xyz[3 * (size_t)k], &t->points[3 * (size_t)i], 12);
        labels[(size_t)k++] = (int32_t)i;
      }
      PCGX_TRY(build_tree(xyz.data(), n_live, 12, 0, labels.data(), &nt));
      for (auto &id : nt->inorder) id = labels[(size_t)id];  // in-order sequence in original ids
    }
    // drop the replaced tree and earlier retired ones unless a session still walks them
    if (t->live) t->retired.push_back(t->live);
    std::vector<pcgx_kdtree *> keep;
    for (pcgx_kdtree *r : t->retired) {
      if (r->sessions.load() == 0) pcgx_kdtree_free(r);
      else keep.push_back(r);
    }
    t->retired.swap(keep);
    t->live = nt;
    t->dirty = false;
  }
  *active = t->live;
  *empty = t->live == nullptr;
  return PCGX_OK;
}

extern "C" pcgx_status pcgx_kdtree_len(const pcgx_kdtree *t, int64_t *n) {
  PCGX_API_LOCK();
  if (!t || !n) return fail(PCGX_E_INVALID, "pcgx_kdtree_len: NULL argument");
  *n = t->n;
  return PCGX_OK;
}

extern "C" pcgx_status pcgx_kdtree_max_depth(const pcgx_kdtree *t, int32_t *depth) {
  PCGX_API_LOCK();
  if (!t || !depth) return fail(PCGX_E_INVALID, "pcgx_kdtree_max_depth: NULL argument");
  if (t->n_deleted > 0) {  // the reference's patched tree (knn_explicit.hip)
    pcgx_kdtree *m = const_cast<pcgx_kdtree *>(t);
    std::lock_guard<std::mutex> lock(m->mu);
    *depth = xtree_max_depth(t);
    return PCGX_OK;
  }
  *depth = t->depth;
  return PCGX_OK;
}

extern "C" pcgx_status pcgx_kdtree_inorder(const pcgx_kdtree *t, int64_t *ids) {
  PCGX_API_LOCK();
  if (!t || !ids) return fail(PCGX_E_INVALID, "pcgx_kdtree_inorder: NULL argument");
  const pcgx_kdtree *a = nullptr;
  bool empty = false;
  PCGX_TRY(resolve_tree(t, &a, &empty));  // after DeletePoint: the remaining (live_count) ids
  if (empty) return PCGX_OK;
  for (int64_t i = 0; i < a->n; i++) ids[i] = a->inorder[i];
  return PCGX_OK;
}

extern "C" pcgx_status pcgx_kdtree_points(const pcgx_kdtree *t, const int64_t *ids, int64_t m,
                                          float *xyz) {
  PCGX_API_LOCK();
  if (!t || (m > 0 && (!ids || !xyz))) return fail(PCGX_E_INVALID, "pcgx_kdtree_points: NULL argument");
  for (int64_t i = 0; i < m; i++) {
    if (ids[i] < 0 || ids[i] >= t->n) return fail(PCGX_E_INVALID, "pcgx_kdtree_points: id %lld out of range", (long long)ids[i]);
    memcpy(xyz + 3 * i, &t->points[3 * ids[i]], 12);
  }
  return PCGX_OK;
}

extern "C" pcgx_status pcgx_kdtree_nearest_batch_dev(const pcgx_kdtree *t, const float *d_q,
                                                     int64_t nq, float max_range,
                                                     float min_dist_sq, uint32_t flags,
                                                     int32_t *d_ids, float *d_dist_sq,
                                                     void *stream) {
  PCGX_API_LOCK();
  if (!t || nq < 0 || (nq > 0 && (!d_q || !d_ids || !d_dist_sq)))
    return fail(PCGX_E_INVALID, "pcgx_kdtree_nearest_batch_dev: bad argument");
  PCGX_TRY(ensure_init());
  hipStream_t st = pick_stream(stream);
  const float max_range_sq = max_range * max_range;  // kdtree.go:91
  if (t->n_deleted > 0) {
    // a handle that has seen DeletePoint: the reference's patched tree, walked in its visit order
    // (exact ties and MinDistSq > 0 as the Go code answers them; knn_explicit.hip)
    PCGX_TRY(ctx().arena.begin(st));
    int32_t *perm = nullptr;
    if ((flags & PCGX_KNN_PRESORT) && nq > 1) {
      PCGX_TRY(ctx().arena.alloc_n((size_t)nq, &perm));
      PCGX_TRY(morton_order(d_q, nq, t->bbox_lo, t->bbox_hi, perm, st));
    }
    return xtree_launch_nearest(t, d_q, perm, nq, max_range_sq, min_dist_sq, d_ids, d_dist_sq, st);
  }
  bool empty = false;
  PCGX_TRY(resolve_tree(t, &t, &empty));  // after DeletePoint: the tree over the remaining points
  if (empty) {
    if (nq > 0)
      hipLaunchKernelGGL(nearest_empty_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, nq, max_range_sq,
                         d_ids, d_dist_sq);
    PCGX_HIP_TRY(hipGetLastError());
    return PCGX_OK;
  }
  PCGX_TRY(ctx().arena.begin(st));
  int32_t *perm = nullptr;
  const bool on_grid = !(min_dist_sq > 0.0f) && grid_enabled(t);
  static const bool partition_off = getenv("PCGX_KNN_PARTITION") && atoi(getenv("PCGX_KNN_PARTITION")) == 0;
  if (on_grid && (flags & PCGX_KNN_PRESORT) && nq > 1 && nq < 0x7fffffffll && !partition_off)
    return grid_launch_nearest_partitioned(t, d_q, nq, max_range_sq, d_ids, d_dist_sq, st);  // (the queries themselves, by coarse cell)
  if ((flags & PCGX_KNN_PRESORT) && nq > 1) {
    PCGX_TRY(ctx().arena.alloc_n((size_t)nq, &perm));
    PCGX_TRY(morton_order(d_q, nq, t->bbox_lo, t->bbox_hi, perm, st));
  }
  // exact mode: answers the uniform grid can certify come from there, the rest from the walk
  // (knn_grid.h); MinDistSq > 0 depends on the visit order: walk only
  if (on_grid)
    return grid_launch_nearest(t, d_q, perm, nq, max_range_sq, d_ids, d_dist_sq, st);
  return launch_nearest(t->view(), d_q, perm, nq, max_range_sq, min_dist_sq, d_ids, d_dist_sq, st);
}

extern "C" pcgx_status pcgx_debug_host_walks(int64_t *queries, int32_t reset) {
  if (!queries) return fail(PCGX_E_INVALID, "pcgx_debug_host_walks: NULL argument");
  *queries = (int64_t)xtree_host_walks(reset != 0);
  return PCGX_OK;
}

extern "C" pcgx_status pcgx_kdtree_nearest_batch(const pcgx_kdtree *t, const float *q, int64_t nq,
                                                 float max_range, float min_dist_sq, int64_t *ids,
                                                 float *dist_sq) {
  PCGX_API_CALL();
  if (!t || nq < 0 || (nq > 0 && (!q || !ids || !dist_sq)))
    return fail(PCGX_E_INVALID, "pcgx_kdtree_nearest_batch: bad argument");
  if (nq == 0) return PCGX_OK;
  PCGX_TRY(ensure_init());
  // a few points: the reference-order walk on the handle's host mirror (knn_explicit.hip) -- a launch and two PCIe round
  // trips cost what sixty such walks do; same ids, same DistSq bits, MinDistSq > 0 included
  if (nq <= xtree_host_walk_max() && !t->points.empty()) {
    xtree_host_nearest(t, q, nq, max_range, min_dist_sq, ids, dist_sq);
    return PCGX_OK;
  }
  hipStream_t st = ctx().stream;
  HostCallBufs b;
  PCGX_TRY(b.alloc(nq, st));
  PCGX_TRY(staged_upload(b.q, q, (size_t)nq * 12, st));
  // large batches are walked in Morton order (the ordering pass costs ~0.08 ms, the walk of 1M
  // random queries gains 0.27 ms); results come back in the caller's order either way
  PCGX_TRY(pcgx_kdtree_nearest_batch_dev(t, b.q, nq, max_range, min_dist_sq,
                                         nq >= (1 << 18) ? PCGX_KNN_PRESORT : 0u, b.ids, b.dsq, st));
  // Go's int is 64 bits wide (Neighbor.ID, search.go:8-11): widened on the device, so that the ids go straight
  // into the caller's slice (a host loop over 1M ids cost as much as the whole search)
  int64_t *d_ids64 = nullptr;
  PCGX_TRY(ctx().host_arena.alloc_n((size_t)nq, &d_ids64));
  hipLaunchKernelGGL(widen_ids_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, (const int32_t *)b.ids, nq, d_ids64);
  PCGX_HIP_TRY(hipGetLastError());
  PCGX_TRY(staged_download(ids, d_ids64, (size_t)nq * 8, st));
  PCGX_TRY(staged_download(dist_sq, b.dsq, (size_t)nq * 4, st));
  return PCGX_OK;
}

// Debug / profiling aid: runs the instrumented exact-mode walk over device queries and returns
// its 32 counters (knn_walk.h).  d_hint_xyz (optional): per query the packed xyz of ANY point of
// the tree, used as the pruning hint the ICP loop takes from its previous iteration; d_leaf_io
// (optional, uint32 per query, 0 = none): predicted first-descent leaves in, the real ones out,
// as the ICP loop carries them from one iteration to the next.  Not part of the drop-in surface.
extern "C" pcgx_status pcgx_debug_walk_stats(const pcgx_kdtree *t, const float *d_q, int64_t nq, float max_range,
                                             int32_t presort, const float *d_hint_xyz, uint32_t *d_leaf_io,
                                             uint64_t stats32[32]) {
  PCGX_API_LOCK();
  if (!t || !d_q || !stats32 || nq <= 0) return fail(PCGX_E_INVALID, "pcgx_debug_walk_stats: bad argument");
  PCGX_TRY(ensure_init());
  hipStream_t st = ctx().stream;
  Arena &ar = ctx().arena;
  PCGX_TRY(ar.begin(st));
  unsigned long long *d_stats = nullptr;
  int32_t *d_ids = nullptr, *perm = nullptr;
  float *d_dsq = nullptr;
  PCGX_TRY(ar.alloc_n((size_t)kStatWords * 4 * ((size_t)ctx().num_cu * 8 + 8), &d_stats));
  PCGX_TRY(ar.alloc_n((size_t)nq, &d_ids));
  PCGX_TRY(ar.alloc_n((size_t)nq, &d_dsq));
  if (presort) {
    PCGX_TRY(ar.alloc_n((size_t)nq, &perm));
    PCGX_TRY(morton_order(d_q, nq, t->bbox_lo, t->bbox_hi, perm, st));
  }
  const TreeView tv = t->view();
  int64_t blocks = (int64_t)ctx().num_cu * walk_blocks_per_cu(tv);
  const int64_t max_blocks = (nq + kKnnBlock - 1) / kKnnBlock;
  if (blocks > max_blocks) blocks = max_blocks;
  if (blocks >= 8) blocks &= ~(int64_t)7;
  hipEvent_t ev0, ev1;
  PCGX_HIP_TRY(hipEventCreate(&ev0));
  PCGX_HIP_TRY(hipEventCreate(&ev1));
  for (int rep = 0; rep < 2; rep++) {  // the first launch warms caches and TLBs: its timings are discarded
    PCGX_HIP_TRY(hipMemsetAsync(d_stats, 0, (size_t)blocks * (kKnnBlock / 64) * kStatWords * sizeof(unsigned long long), st));
    PCGX_HIP_TRY(hipEventRecord(ev0, st));
    hipLaunchKernelGGL((nearest_kernel<false, true>), dim3((unsigned)blocks), dim3(kKnnBlock),
                       walk_lds_bytes(tv, kKnnBlock), st, tv, d_q, perm, nq, max_range * max_range, 0.0f, d_ids,
                       d_dsq, d_stats, d_hint_xyz, d_leaf_io);
    PCGX_HIP_TRY(hipEventRecord(ev1, st));
  }
  PCGX_HIP_TRY(hipGetLastError());
  std::vector<uint64_t> rows((size_t)blocks * (kKnnBlock / 64) * kStatWords);
  PCGX_HIP_TRY(hipMemcpyAsync(rows.data(), d_stats, rows.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
  PCGX_HIP_TRY(hipStreamSynchronize(st));
  if (const char *dump = getenv("PCGX_DEBUG_WALK_ROWS")) {  // raw per-wave rows (uint64[waves][24]) for tools/
    if (FILE *f = fopen(dump, "wb")) {
      fwrite(rows.data(), sizeof(uint64_t), rows.size(), f);
      fclose(f);
    }
  }
  for (int k = 0; k < kStatWords; k++) stats32[k] = 0;
  for (size_t w = 0; w < rows.size() / kStatWords; w++)
    for (int k = 0; k < kStatWords; k++) {
      const uint64_t v = rows[w * kStatWords + k];
      if (k == 15 || k == 18) stats32[k] = v > stats32[k] ? v : stats32[k];
      else stats32[k] += v;
    }
  float ms = 0.0f;
  PCGX_HIP_TRY(hipEventElapsedTime(&ms, ev0, ev1));
  stats32[20] = (uint64_t)(ms * 1.0e6f);  // instrumented kernel, nanoseconds
  (void)hipEventDestroy(ev0);
  (void)hipEventDestroy(ev1);
  return PCGX_OK;
}
