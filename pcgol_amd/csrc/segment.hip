// segment.hip -- bucket voxel grid, voxel flood-fill segmentation and region growing on gfx950
// (the consumers of the spatial structures either side of the hot path).
//
// Reference:
//   pc/storage/voxelgrid/voxelgrid.go:7-122        VoxelGrid ([][]int buckets, Addr / Add / Get)
//   pc/segmentation/voxelgrid/voxelgrid.go:39-73   Segment: 26-neighbour flood fill from a seed
//   pc/segmentation/regiongrowing/regiongrowing.go:23-56  Segment: BFS over Range() neighbourhoods
//                                                          restricted to one property value
//
// The reference answers ONE seed per call with a sequential BFS.  Both searches are reachability
// in an undirected graph, so the device computes the connected components of the WHOLE graph once
// (lock-free union-find, every edge handled by one lane) and a seed query becomes a lookup:
//   flood fill:      vertices = occupied voxels, edges = 26-neighbourhood
//   region growing:  vertices = points, edges = {i, j} with |pi - pj|^2 < maxRange^2 and equal
//                    property value (the neighbourhoods are KDTree.Range's, range_walk.h)
// Component ids are canonical (smallest voxel address / smallest point id of the component).
// The reference returns ids in BFS discovery order and its own tests sort them before comparing
// (segmentation/voxelgrid/voxelgrid_test.go:36, regiongrowing_test.go:168); here the order is
// ascending voxel address (insertion order inside a voxel) resp. ascending point id.
#include <string.h>

#include <algorithm>
#include <deque>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "knn_grid.h"
#include "range_walk.h"

namespace pcgx {

// ------------------------------------------------------------------ union-find
// parent[x] <= x always: roots are hooked under smaller roots only and paths are compressed
// towards ancestors, so the root of a finished component is its smallest member.
__device__ __forceinline__ uint32_t uf_load(const uint32_t *p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }

__device__ __forceinline__ uint32_t uf_find(uint32_t *__restrict__ parent, uint32_t x) {
  for (;;) {
    const uint32_t p = uf_load(parent + x);
    if (p == x) return x;
    const uint32_t gp = uf_load(parent + p);
    // path halving with a plain store: whatever another thread writes here at the same time is an ancestor of x
    // too (parents only ever move towards the root), so either value keeps the structure -- and an atomic per hop
    // is what the finds used to cost
    if (gp != p) __atomic_store_n(parent + x, gp, __ATOMIC_RELAXED);
    x = p;
  }
}

__device__ __forceinline__ void uf_union(uint32_t *__restrict__ parent, uint32_t a, uint32_t b) {
  a = uf_find(parent, a);
  b = uf_find(parent, b);
  while (a != b) {
    if (a < b) {
      const uint32_t t = a;
      a = b;
      b = t;
    }
    const uint32_t old = atomicCAS(parent + a, a, b);  // hook the larger root under the smaller
    if (old == a) return;
    a = uf_find(parent, old);
    b = uf_find(parent, b);
  }
}

__global__ __launch_bounds__(256) void uf_init_kernel(uint32_t *__restrict__ parent, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) parent[i] = (uint32_t)i;
}

__global__ __launch_bounds__(256) void uf_flatten_kernel(uint32_t *__restrict__ parent, int64_t n,
                                                         uint32_t *__restrict__ root) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) root[i] = uf_find(parent, (uint32_t)i);
}

// ------------------------------------------------------------------ bucket grid
struct GridParams {
  float origin[3];
  float resolution_inv;
  int64_t size[3];
  int64_t len;
};

__device__ __forceinline__ float ld_f32_unaligned(const uint8_t *p) {
  float v;
  __builtin_memcpy(&v, p, 4);
  return v;
}

// VoxelGrid.Addr (storage/voxelgrid/voxelgrid.go:64-79): int(pos*resolutionInv + 0.5) per axis,
// Go's float->int truncation; outside the grid -> key = len (sorts behind every voxel).
__global__ __launch_bounds__(256) void grid_key_kernel(const uint8_t *__restrict__ data, int64_t n, int32_t stride,
                                                       int32_t off, GridParams gp, uint32_t *__restrict__ key,
                                                       uint32_t *__restrict__ key_orig, uint32_t *__restrict__ idx) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint8_t *rec = data + i * stride + off;
  int64_t v[3];
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float pos = ld_f32_unaligned(rec + 4 * k) - gp.origin[k];
    const float f = pos * gp.resolution_inv + 0.5f;
    // NaN or beyond int64: Go's conversion is implementation defined; such a point is outside
    ok = ok && (f == f) && f > -9.0e18f && f < 9.0e18f;
    v[k] = ok ? (int64_t)f : -1;
    ok = ok && v[k] >= 0 && v[k] < gp.size[k];
  }
  const uint32_t a = ok ? (uint32_t)(v[0] + (v[1] + v[2] * gp.size[1]) * gp.size[0]) : (uint32_t)gp.len;
  key[i] = a;
  key_orig[i] = a;
  idx[i] = (uint32_t)i;
}

// Run heads of the sorted keys: head_rank via block scan (one block per 2048 keys) in two launches.
constexpr int kRunTile = 2048;

__device__ __forceinline__ bool run_head(const uint32_t *__restrict__ k, int64_t j) { return j == 0 || k[j] != k[j - 1]; }

__global__ __launch_bounds__(256) void run_count_kernel(const uint32_t *__restrict__ k, int64_t n,
                                                        uint32_t *__restrict__ tile_count) {
  __shared__ uint32_t ws[4];
  const int64_t base = (int64_t)blockIdx.x * kRunTile;
  uint32_t c = 0;
  for (int r = 0; r < kRunTile / 256; r++) {
    const int64_t j = base + r * 256 + threadIdx.x;
    if (j < n && run_head(k, j)) c++;
  }
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) tile_count[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// exclusive scan of the tile counts by one block; total -> *total
__global__ __launch_bounds__(1024) void run_scan_kernel(uint32_t *__restrict__ tile_count, int ntiles,
                                                        uint32_t *__restrict__ total) {
  __shared__ uint32_t ws[16];
  __shared__ uint32_t carry_s;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int start = 0; start < ntiles; start += 1024) {
    const int i = start + threadIdx.x;
    const uint32_t v = i < ntiles ? tile_count[i] : 0u;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(inc, o);
      if (lane >= o) inc += t;
    }
    if (lane == 63) ws[wave] = inc;
    __syncthreads();
    uint32_t wbase = 0;
    for (int w = 0; w < wave; w++) wbase += ws[w];
    const uint32_t carry = carry_s;
    if (i < ntiles) tile_count[i] = carry + wbase + inc - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = carry + wbase + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry_s;
}

// run r: run_key[r] = its key, run_start[r] = first sorted position (run_start[runs] = n is set by the host)
__global__ __launch_bounds__(256) void run_write_kernel(const uint32_t *__restrict__ k, int64_t n,
                                                        const uint32_t *__restrict__ tile_offset,
                                                        uint32_t *__restrict__ run_key,
                                                        uint32_t *__restrict__ run_start) {
  __shared__ uint32_t ws[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t base = (int64_t)blockIdx.x * kRunTile;
  uint32_t running = tile_offset[blockIdx.x];
  const uint64_t lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  for (int r = 0; r < kRunTile / 256; r++) {
    const int64_t j = base + r * 256 + threadIdx.x;
    const bool head = j < n && run_head(k, j);
    const uint64_t bal = __ballot(head);
    if (lane == 0) ws[wave] = (uint32_t)__popcll(bal);
    __syncthreads();
    uint32_t wbase = 0, round_total = 0;
    for (int w = 0; w < 4; w++) {
      if (w < wave) wbase += ws[w];
      round_total += ws[w];
    }
    if (head) {
      const uint32_t slot = running + wbase + (uint32_t)__popcll(bal & lt_mask);
      run_key[slot] = k[j];
      run_start[slot] = (uint32_t)j;
    }
    running += round_total;
    __syncthreads();
  }
}

__device__ __forceinline__ int64_t lower_bound_u32(const uint32_t *__restrict__ a, int64_t n, uint32_t v) {
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (a[mid] < v) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// Flood-fill connectivity (segmentation/voxelgrid/voxelgrid.go:13-25,62-70): an occupied voxel is
// joined with each of its occupied 26-neighbours; every undirected edge is handled once, from
// the voxel with the larger address (13 of the 26 offsets).
__global__ __launch_bounds__(256) void grid_union_kernel(const uint32_t *__restrict__ cell_addr, int64_t m,
                                                         GridParams gp, uint32_t *__restrict__ parent) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= m) return;
  const int64_t a = cell_addr[c];
  const int64_t x = a % gp.size[0], y = (a / gp.size[0]) % gp.size[1], z = a / (gp.size[0] * gp.size[1]);
  for (int dz = -1; dz <= 0; dz++)
    for (int dy = -1; dy <= (dz < 0 ? 1 : 0); dy++)
      for (int dx = -1; dx <= ((dz < 0 || dy < 0) ? 1 : -1); dx++) {
        const int64_t nx = x + dx, ny = y + dy, nz = z + dz;
        if (nx < 0 || ny < 0 || nz < 0 || nx >= gp.size[0] || ny >= gp.size[1] || nz >= gp.size[2]) continue;
        const uint32_t na = (uint32_t)(nx + (ny + nz * gp.size[1]) * gp.size[0]);
        const int64_t j = lower_bound_u32(cell_addr, c, na);  // smaller address: before c
        if (j < c && cell_addr[j] == na) uf_union(parent, (uint32_t)c, (uint32_t)j);
      }
}

// cell_comp[c] = address of the smallest voxel of c's component
__global__ __launch_bounds__(256) void grid_comp_kernel(const uint32_t *__restrict__ root,
                                                        const uint32_t *__restrict__ cell_addr, int64_t m,
                                                        uint32_t *__restrict__ cell_comp) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c < m) cell_comp[c] = cell_addr[root[c]];
}

// ------------------------------------------------------------------ region growing
// One lane per tree node: the Range() neighbourhood of the node's point (regiongrowing.go:47),
// joined with every neighbour of the same property value (:43-45).  Each undirected edge is
// seen from both ends (the distance expression is symmetric bit for bit); the larger id unions.
__global__ __launch_bounds__(kRangeWalkBlock) void rg_union_kernel(TreeView tv, const uint32_t *__restrict__ labels,
                                                                   float bound, uint32_t *__restrict__ parent) {
  extern __shared__ uint32_t s_stack[];
  const uint32_t b = blockIdx.x * kRangeWalkBlock + threadIdx.x + 1u;
  if (b >= (1u << tv.depth)) return;
  const int depth = 31 - __clz((int)b);
  const uint32_t sz = node_size(b, depth, (uint32_t)tv.n + 1u);
  if (sz == 0u || sz > (uint32_t)tv.n) return;  // slot without a node
  const float4 nd = node_at(tv.nodes, b);
  const int32_t id = __float_as_int(nd.w);
  const uint32_t mine = labels[id];
  uint32_t my_root = (uint32_t)id;  // (a root this point was seen under: most neighbours are under it already)
  range_walk(tv, s_stack + threadIdx.x, kRangeWalkBlock, nd.x, nd.y, nd.z, bound, [&](int32_t j, float) {
    if (j < id && labels[j] == mine) {
      const uint32_t rj = uf_find(parent, (uint32_t)j);
      if (rj != my_root) {
        uf_union(parent, my_root, rj);
        my_root = uf_find(parent, my_root);
      }
    }
  });
}

// The same components on a handle with a uniform grid (knn_grid.h).  Which points lie within maxRange of a point does
// not depend on the walk (range.hip), and a union-find does not care in which order it is told the edges -- but it does
// care how many trees it has to merge and who merges them at the same time: the single pass above spends 370 of its
// 600 us at 300k points in finds and compare-and-swap retries.  Here the union-find works on the points' positions in
// CELL ORDER (the grid's own order), one lane per point:
//  (1) every point hooks itself under the same-valued neighbour with the SMALLEST position below its own (a store to
//      its own word: no contention, no find).  In cell order "smallest" means lowest z, then y, then x: the trees
//      are columns that run down through a region, and only points with nothing of their region below them are roots
//      -- hooked by point id, whose order has nothing to do with space, every sixth point was one;
//  (2) pointer jumping makes every parent a root;
//  (3) a second pass over the neighbourhoods unions what is still apart (few trees, roots one hop away);
//  (4) roots -> the component's smallest point id (what the walk's version returns).
template <bool kHookMin>
__global__ __launch_bounds__(256) void rg_grid_kernel(GridView g, int64_t n, const uint32_t *__restrict__ labels, float bound,
                                                      uint32_t *__restrict__ parent) {
  const int64_t f0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (f0 >= n) return;
  const float4 me = g.pts[f0];
  const uint32_t mine = labels[__float_as_uint(me.w)];
  uint32_t best = (uint32_t)f0;                                          // kHookMin: the lowest same-valued neighbour
  uint32_t my_root = kHookMin ? (uint32_t)f0 : uf_find(parent, (uint32_t)f0);  // else: a root this point was seen under
  const GridBox box = grid_cover(g, me.x, me.y, me.z, bound);
  for (int z = box.z0; z <= box.z1; z++) {
    for (int y = box.y0; y <= box.y1; y++) {
      const uint32_t row = ((uint32_t)z * (uint32_t)g.ny + (uint32_t)y) * (uint32_t)g.nx;
      uint32_t f = g.start[row + (uint32_t)box.x0];
      uint32_t e = g.start[row + (uint32_t)box.x1 + 1u];
      e = e < (uint32_t)f0 ? e : (uint32_t)f0;  // each edge from its higher end
      for (; f < e; f++) {
        const float4 p = g.pts[f];
        const float dx = p.x - me.x, dy = p.y - me.y, dz = p.z - me.z;
        const float d = (dx * dx + dy * dy) + dz * dz;  // (mat/vec3.go:18-20,38-40, as in the walk)
        if (!(d < bound) || labels[__float_as_uint(p.w)] != mine) continue;  // regiongrowing.go:43-47
        if (kHookMin) {
          best = f < best ? f : best;
        } else {
          const uint32_t rj = uf_find(parent, f);
          if (rj != my_root) {
            uf_union(parent, my_root, rj);
            my_root = uf_find(parent, my_root);
          }
        }
      }
    }
  }
  if (kHookMin) parent[f0] = best;
}

// (4): min_id[root position] = smallest point id under it, then root_of_point[id] = that id
__global__ __launch_bounds__(256) void rg_min_id_kernel(GridView g, int64_t n, const uint32_t *__restrict__ parent,
                                                        uint32_t *__restrict__ min_id) {
  // The lanes of a wave are neighbours in space and mostly of one region: one atomic per distinct root of the wave,
  // and none when the word already holds something smaller (150k atomicMin on the one word of a large region take
  // 2 ms: same-address atomics are served one after the other)
  const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool valid = f < n;
  const uint32_t root = valid ? parent[f] : 0xffffffffu;
  const uint32_t id = valid ? __float_as_uint(g.pts[f].w) : 0xffffffffu;
  const int lane = threadIdx.x & 63;
  unsigned long long todo = __ballot(valid);
  while (todo) {  // uniform
    const int leader = __ffsll((long long)todo) - 1;
    const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)root, leader);
    const unsigned long long same = __ballot(valid && root == r);
    uint32_t m = (valid && root == r) ? id : 0xffffffffu;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const uint32_t v = (uint32_t)__shfl_xor((int)m, o);
      m = v < m ? v : m;
    }
    if (lane == leader && m < __atomic_load_n(min_id + r, __ATOMIC_RELAXED)) atomicMin(min_id + r, m);
    todo &= ~same;
  }
}
__global__ __launch_bounds__(256) void rg_name_kernel(GridView g, int64_t n, const uint32_t *__restrict__ parent,
                                                      const uint32_t *__restrict__ min_id, uint32_t *__restrict__ root_of_point) {
  const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (f < n) root_of_point[__float_as_uint(g.pts[f].w)] = min_id[parent[f]];
}

// parent[i] := root of i (every chain descends: roots are their components' smallest members so far)
__global__ __launch_bounds__(256) void uf_jump_kernel(uint32_t *__restrict__ parent, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t r = uf_load(parent + i);
  for (;;) {
    const uint32_t p = uf_load(parent + r);
    if (p == r) break;
    r = p;
  }
  __atomic_store_n(parent + i, r, __ATOMIC_RELAXED);
}

}  // namespace pcgx

using namespace pcgx;

struct pcgx_bucket_grid {
  int64_t n = 0;  // points offered (Add(point i, i) for i in [0, n))
  float resolution = 0.0f;
  GridParams gp;
  // host copies of the CSR buckets (downloaded once): occupied voxels ascending
  std::vector<uint32_t> cell_addr, cell_start, idx_sorted, point_key;
  std::vector<int32_t> cell_of_addr;  // lazily (segment_bfs, grids up to 2^27 voxels): address -> voxel, -1 empty
  std::vector<uint32_t> cell_comp;  // lazily: smallest voxel address of each voxel's component
  bool have_comp = false;
  int64_t n_in = 0;
};

static pcgx_status grid_params(float resolution, const int64_t size[3], const float origin[3], GridParams &gp) {
  if (!(resolution > 0.0f)) return fail(PCGX_E_INVALID, "bucket grid: resolution must be > 0");
  double len = 1.0;
  for (int k = 0; k < 3; k++) {
    if (size[k] < 0) return fail(PCGX_E_INVALID, "bucket grid: negative size");
    gp.size[k] = size[k];
    gp.origin[k] = origin[k];
    len *= (double)size[k];
  }
  if (len >= 4294967295.0)
    return fail(PCGX_E_TOO_LARGE, "bucket grid: %lld x %lld x %lld voxels exceed 2^32-2", (long long)size[0],
                (long long)size[1], (long long)size[2]);
  gp.len = size[0] * size[1] * size[2];
  gp.resolution_inv = 1.0f / resolution;  // voxelgrid.go:21
  return PCGX_OK;
}

static int bits_for_count(int64_t count) {
  int b = 0;
  while (b < 63 && ((int64_t)1 << b) < count) b++;
  return b;
}

extern "C" pcgx_status pcgx_bucket_grid_build(const void *data, int64_t n, int32_t stride, int32_t xyz_off,
                                              float resolution, const int64_t size[3], const float origin[3],
                                              pcgx_bucket_grid **out) {
  PCGX_API_LOCK();
  if (!out) return fail(PCGX_E_INVALID, "pcgx_bucket_grid_build: out is NULL");
  *out = nullptr;
  if (n < 0 || !size || !origin || (n > 0 && !data)) return fail(PCGX_E_INVALID, "pcgx_bucket_grid_build: bad argument");
  if (n > 0 && (stride < 12 || xyz_off < 0 || xyz_off + 12 > stride))
    return fail(PCGX_E_BAD_FIELD, "pcgx_bucket_grid_build: stride %d / xyz offset %d do not hold an xyz triple", stride, xyz_off);
  if (n > 0x7fffffffll) return fail(PCGX_E_TOO_LARGE, "pcgx_bucket_grid_build: more than 2^31-1 points");
  GridParams gp;
  PCGX_TRY(grid_params(resolution, size, origin, gp));
  pcgx_bucket_grid *g = new pcgx_bucket_grid();
  g->n = n;
  g->resolution = resolution;
  g->gp = gp;
  g->cell_start.assign(1, 0u);
  if (n == 0) {
    *out = g;
    return PCGX_OK;
  }
  pcgx_status rc = ensure_init();
  if (rc != PCGX_OK) { delete g; return rc; }
  hipStream_t st = ctx().stream;
  Arena &ar = ctx().arena;
  uint8_t *d_data = nullptr;
  uint32_t *keys[2] = {nullptr, nullptr}, *vals[2] = {nullptr, nullptr}, *key_orig = nullptr, *tile_count = nullptr,
           *run_key = nullptr, *run_start = nullptr, *d_total = nullptr;
  void *ws = nullptr;
  const int ntiles = (int)((n + kRunTile - 1) / kRunTile);
  auto body = [&]() -> pcgx_status {
    PCGX_TRY(ar.begin(st));
    PCGX_TRY(ar.alloc_n((size_t)n * stride, &d_data));
    PCGX_TRY(ar.alloc_n((size_t)n, &keys[0]));
    PCGX_TRY(ar.alloc_n((size_t)n, &keys[1]));
    PCGX_TRY(ar.alloc_n((size_t)n, &vals[0]));
    PCGX_TRY(ar.alloc_n((size_t)n, &vals[1]));
    PCGX_TRY(ar.alloc_n((size_t)n, &key_orig));
    PCGX_TRY(ar.alloc_n((size_t)ntiles, &tile_count));
    PCGX_TRY(ar.alloc_n((size_t)n, &run_key));
    PCGX_TRY(ar.alloc_n((size_t)n, &run_start));
    PCGX_TRY(ar.alloc_n(1, &d_total));
    PCGX_TRY(ar.alloc(radix_sort_workspace_bytes(n), &ws));
    PCGX_HIP_TRY(hipMemcpyAsync(d_data, data, (size_t)n * stride, hipMemcpyHostToDevice, st));
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(grid_key_kernel, dim3(nb), dim3(256), 0, st, (const uint8_t *)d_data, n, stride, xyz_off, gp,
                       keys[0], key_orig, vals[0]);
    int res = 0;
    // stable: a voxel's points stay in insertion order (append, voxelgrid.go:43)
    PCGX_TRY(radix_sort_pairs(keys, vals, n, bits_for_count(gp.len + 1), ws, &res, st));
    hipLaunchKernelGGL(run_count_kernel, dim3(ntiles), dim3(256), 0, st, keys[res], n, tile_count);
    hipLaunchKernelGGL(run_scan_kernel, dim3(1), dim3(1024), 0, st, tile_count, ntiles, d_total);
    hipLaunchKernelGGL(run_write_kernel, dim3(ntiles), dim3(256), 0, st, keys[res], n, tile_count, run_key, run_start);
    PCGX_HIP_TRY(hipGetLastError());
    uint32_t runs = 0;
    PCGX_HIP_TRY(hipMemcpyAsync(&runs, d_total, 4, hipMemcpyDeviceToHost, st));
    PCGX_HIP_TRY(hipStreamSynchronize(st));
    g->cell_addr.resize(runs);
    g->cell_start.resize((size_t)runs + 1);
    g->idx_sorted.resize((size_t)n);
    g->point_key.resize((size_t)n);
    PCGX_HIP_TRY(hipMemcpyAsync(g->cell_addr.data(), run_key, (size_t)runs * 4, hipMemcpyDeviceToHost, st));
    PCGX_HIP_TRY(hipMemcpyAsync(g->cell_start.data(), run_start, (size_t)runs * 4, hipMemcpyDeviceToHost, st));
    PCGX_HIP_TRY(hipMemcpyAsync(g->idx_sorted.data(), vals[res], (size_t)n * 4, hipMemcpyDeviceToHost, st));
    PCGX_HIP_TRY(hipMemcpyAsync(g->point_key.data(), key_orig, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    PCGX_HIP_TRY(hipStreamSynchronize(st));
    g->cell_start[runs] = (uint32_t)n;
    // the last run holds the points outside the grid (key == len): not a voxel
    if (runs > 0 && g->cell_addr[runs - 1] == (uint32_t)gp.len) {
      g->cell_addr.pop_back();
      g->cell_start.pop_back();
    }
    g->n_in = g->cell_start.back();
    g->idx_sorted.resize((size_t)g->n_in);
    return PCGX_OK;
  };
  rc = body();
  if (rc != PCGX_OK) {
    delete g;
    return rc;
  }
  *out = g;
  return PCGX_OK;
}

extern "C" pcgx_status pcgx_bucket_grid_free(pcgx_bucket_grid *g) {
  PCGX_API_LOCK();
  delete g;
  return PCGX_OK;
}

// {Len() = voxels of the grid (voxelgrid.go:110-112), points accepted by Add, occupied voxels}
extern "C" pcgx_status pcgx_bucket_grid_counts(const pcgx_bucket_grid *g, int64_t *len, int64_t *n_added,
                                               int64_t *n_occupied) {
  PCGX_API_LOCK();
  if (!g) return fail(PCGX_E_INVALID, "pcgx_bucket_grid_counts: NULL grid");
  if (len) *len = g->gp.len;
  if (n_added) *n_added = g->n_in;
  if (n_occupied) *n_occupied = (int64_t)g->cell_addr.size();
  return PCGX_OK;
}

// VoxelGrid.Addr (voxelgrid.go:64-79), host arithmetic identical to grid_key_kernel
static bool grid_addr_host(const GridParams &gp, const float p[3], int64_t *addr, int64_t xyz[3]) {
  for (int k = 0; k < 3; k++) {
    const float pos = p[k] - gp.origin[k];
    const float f = pos * gp.resolution_inv + 0.5f;
    if (!(f == f) || !(f > -9.0e18f) || !(f < 9.0e18f)) return false;
    const int64_t v = (int64_t)f;
    if (v < 0 || v >= gp.size[k]) return false;
    xyz[k] = v;
  }
  *addr = xyz[0] + (xyz[1] + xyz[2] * gp.size[1]) * gp.size[0];
  return true;
}

extern "C" pcgx_status pcgx_bucket_grid_addr(const pcgx_bucket_grid *g, const float p[3], int64_t *addr, int32_t *ok) {
  PCGX_API_LOCK();
  if (!g || !p || !addr || !ok) return fail(PCGX_E_INVALID, "pcgx_bucket_grid_addr: NULL argument");
  int64_t xyz[3];
  *addr = 0;
  *ok = grid_addr_host(g->gp, p, addr, xyz) ? 1 : 0;
  return PCGX_OK;
}

// addr of every offered point, -1 where Add returned false (voxelgrid.go:37-41)
extern "C" pcgx_status pcgx_bucket_grid_point_addrs(const pcgx_bucket_grid *g, int64_t *addrs) {
  PCGX_API_LOCK();
  if (!g || (g->n > 0 && !addrs)) return fail(PCGX_E_INVALID, "pcgx_bucket_grid_point_addrs: NULL argument");
  for (int64_t i = 0; i < g->n; i++)
    addrs[i] = g->point_key[(size_t)i] == (uint32_t)g->gp.len ? -1 : (int64_t)g->point_key[(size_t)i];
  return PCGX_OK;
}

static int64_t find_cell(const pcgx_bucket_grid *g, int64_t addr) {
  auto it = std::lower_bound(g->cell_addr.begin(), g->cell_addr.end(), (uint32_t)addr);
  if (it == g->cell_addr.end() || *it != (uint32_t)addr) return -1;
  return it - g->cell_addr.begin();
}

// GetByAddr (voxelgrid.go:60-62): *count = bucket length; the first min(count, cap) ids are written
extern "C" pcgx_status pcgx_bucket_grid_get_by_addr(const pcgx_bucket_grid *g, int64_t addr, int64_t *out, int64_t cap,
                                                    int64_t *count) {
  PCGX_API_LOCK();
  if (!g || !count || cap < 0 || (cap > 0 && !out)) return fail(PCGX_E_INVALID, "pcgx_bucket_grid_get_by_addr: bad argument");
  if (addr < 0 || addr >= g->gp.len)
    return fail(PCGX_E_OUT_OF_RANGE, "voxel address %lld outside the grid (the reference panics: index out of range)", (long long)addr);
  *count = 0;
  const int64_t c = find_cell(g, addr);
  if (c < 0) return PCGX_OK;
  const int64_t s = g->cell_start[(size_t)c], e = g->cell_start[(size_t)c + 1];
  *count = e - s;
  for (int64_t k = 0; k < e - s && k < cap; k++) out[k] = g->idx_sorted[(size_t)(s + k)];
  return PCGX_OK;
}

// Get (voxelgrid.go:52-58): *count = -1 for nil (p outside the grid)
extern "C" pcgx_status pcgx_bucket_grid_get(const pcgx_bucket_grid *g, const float p[3], int64_t *out, int64_t cap,
                                            int64_t *count) {
  PCGX_API_LOCK();
  if (!g || !p || !count) return fail(PCGX_E_INVALID, "pcgx_bucket_grid_get: NULL argument");
  int64_t addr, xyz[3];
  if (!grid_addr_host(g->gp, p, &addr, xyz)) {
    *count = -1;
    return PCGX_OK;
  }
  return pcgx_bucket_grid_get_by_addr(g, addr, out, cap, count);
}

// Indice (voxelgrid.go:114-120): all ids, voxels ascending, insertion order inside a voxel
extern "C" pcgx_status pcgx_bucket_grid_indice(const pcgx_bucket_grid *g, int64_t *out) {
  PCGX_API_LOCK();
  if (!g || (g->n_in > 0 && !out)) return fail(PCGX_E_INVALID, "pcgx_bucket_grid_indice: NULL argument");
  for (int64_t k = 0; k < g->n_in; k++) out[k] = g->idx_sorted[(size_t)k];
  return PCGX_OK;
}

static pcgx_status ensure_components(pcgx_bucket_grid *g) {
  if (g->have_comp) return PCGX_OK;
  const int64_t m = (int64_t)g->cell_addr.size();
  g->cell_comp.assign((size_t)m, 0u);
  if (m > 0) {
    PCGX_TRY(ensure_init());
    hipStream_t st = ctx().stream;
    Arena &ar = ctx().arena;
    PCGX_TRY(ar.begin(st));
    uint32_t *d_addr = nullptr, *d_parent = nullptr, *d_root = nullptr, *d_comp = nullptr;
    PCGX_TRY(ar.alloc_n((size_t)m, &d_addr));
    PCGX_TRY(ar.alloc_n((size_t)m, &d_parent));
    PCGX_TRY(ar.alloc_n((size_t)m, &d_root));
    PCGX_TRY(ar.alloc_n((size_t)m, &d_comp));
    PCGX_HIP_TRY(hipMemcpyAsync(d_addr, g->cell_addr.data(), (size_t)m * 4, hipMemcpyHostToDevice, st));
    const unsigned nb = (unsigned)((m + 255) / 256);
    hipLaunchKernelGGL(uf_init_kernel, dim3(nb), dim3(256), 0, st, d_parent, m);
    hipLaunchKernelGGL(grid_union_kernel, dim3(nb), dim3(256), 0, st, (const uint32_t *)d_addr, m, g->gp, d_parent);
    hipLaunchKernelGGL(uf_flatten_kernel, dim3(nb), dim3(256), 0, st, d_parent, m, d_root);
    hipLaunchKernelGGL(grid_comp_kernel, dim3(nb), dim3(256), 0, st, (const uint32_t *)d_root, (const uint32_t *)d_addr,
                       m, d_comp);
    PCGX_HIP_TRY(hipGetLastError());
    PCGX_HIP_TRY(hipMemcpyAsync(g->cell_comp.data(), d_comp, (size_t)m * 4, hipMemcpyDeviceToHost, st));
    PCGX_HIP_TRY(hipStreamSynchronize(st));
  }
  g->have_comp = true;
  return PCGX_OK;
}

// Flood-fill component of every offered point: the smallest voxel address of the 26-connected
// set of occupied voxels its voxel belongs to; -1 for points outside the grid.  Segment(p) for
// every seed at once.
extern "C" pcgx_status pcgx_bucket_grid_components(pcgx_bucket_grid *g, int64_t *point_comp) {
  PCGX_API_LOCK();
  if (!g || (g->n > 0 && !point_comp)) return fail(PCGX_E_INVALID, "pcgx_bucket_grid_components: NULL argument");
  PCGX_TRY(ensure_components(g));
  for (int64_t i = 0; i < g->n; i++) point_comp[i] = -1;
  for (size_t c = 0; c < g->cell_addr.size(); c++)
    for (uint32_t e = g->cell_start[c]; e < g->cell_start[c + 1]; e++) point_comp[g->idx_sorted[e]] = g->cell_comp[c];
  return PCGX_OK;
}

// VoxelGrid.Segment(p) (segmentation/voxelgrid/voxelgrid.go:39-73): the ids of every point in the
// 26-connected set of occupied voxels around p's voxel; empty if p is outside the grid or its
// voxel is empty.  *count = result length; the first min(count, cap) ids are written.
extern "C" pcgx_status pcgx_bucket_grid_segment(pcgx_bucket_grid *g, const float p[3], int64_t *out, int64_t cap,
                                                int64_t *count) {
  PCGX_API_LOCK();
  if (!g || !p || !count || cap < 0 || (cap > 0 && !out)) return fail(PCGX_E_INVALID, "pcgx_bucket_grid_segment: bad argument");
  *count = 0;
  int64_t addr, xyz[3];
  if (!grid_addr_host(g->gp, p, &addr, xyz)) return PCGX_OK;
  const int64_t c0 = find_cell(g, addr);
  if (c0 < 0) return PCGX_OK;
  PCGX_TRY(ensure_components(g));
  const uint32_t comp = g->cell_comp[(size_t)c0];
  int64_t k = 0;
  for (size_t c = 0; c < g->cell_addr.size(); c++) {
    if (g->cell_comp[c] != comp) continue;
    for (uint32_t e = g->cell_start[c]; e < g->cell_start[c + 1]; e++) {
      if (k < cap) out[k] = g->idx_sorted[e];
      k++;
    }
  }
  *count = k;
  return PCGX_OK;
}

// The same Segment(p) in the reference's own order: its FIFO flood fill (voxelgrid.go:39-73, cursor
// order x, y, z in {-1, 0, 1}, :13-25) run on the host over the device-built sparse buckets, so the
// ids come out exactly as the Go code appends them.  Costs 26 voxel look-ups per voxel of the
// component (dense address -> voxel map up to 2^27 voxels, hashing beyond); use
// pcgx_bucket_grid_segment / _components when the order does not matter.
extern "C" pcgx_status pcgx_bucket_grid_segment_bfs(pcgx_bucket_grid *g, const float p[3], int64_t *out, int64_t cap,
                                                    int64_t *count) {
  PCGX_API_LOCK();
  if (!g || !p || !count || cap < 0 || (cap > 0 && !out)) return fail(PCGX_E_INVALID, "pcgx_bucket_grid_segment_bfs: bad argument");
  *count = 0;
  int64_t addr0, pos[3];
  if (!grid_addr_host(g->gp, p, &addr0, pos)) return PCGX_OK;  // PosInt failed (voxelgrid.go:41-44)
  const GridParams &gp = g->gp;
  const bool dense = gp.len <= ((int64_t)1 << 27);
  if (dense && g->cell_of_addr.empty() && gp.len > 0) {
    g->cell_of_addr.assign((size_t)gp.len, -1);
    for (size_t c = 0; c < g->cell_addr.size(); c++) g->cell_of_addr[g->cell_addr[c]] = (int32_t)c;
  }
  std::vector<uint8_t> searched_dense;
  std::unordered_set<int64_t> searched_sparse;
  if (dense) searched_dense.assign((size_t)gp.len, 0);
  auto test_and_set = [&](int64_t a) {  // returns true if already searched
    if (dense) {
      const bool was = searched_dense[(size_t)a] != 0;
      searched_dense[(size_t)a] = 1;
      return was;
    }
    return !searched_sparse.insert(a).second;
  };
  auto is_searched = [&](int64_t a) { return dense ? searched_dense[(size_t)a] != 0 : searched_sparse.count(a) != 0; };
  auto addr_of = [&](const int64_t v[3], int64_t *a) {  // AddrByPosInt (voxelgrid.go:81-92)
    if (v[0] < 0 || v[1] < 0 || v[2] < 0 || v[0] >= gp.size[0] || v[1] >= gp.size[1] || v[2] >= gp.size[2]) return false;
    *a = v[0] + (v[1] + v[2] * gp.size[1]) * gp.size[0];
    return true;
  };
  struct P3 { int64_t v[3]; };
  std::deque<P3> next;
  next.push_back(P3{{pos[0], pos[1], pos[2]}});
  int64_t k = 0;
  while (!next.empty()) {
    const P3 cur = next.front();
    next.pop_front();
    int64_t a;
    if (!addr_of(cur.v, &a) || test_and_set(a)) continue;
    const int64_t c = dense ? g->cell_of_addr[(size_t)a] : find_cell(g, a);
    if (c < 0) continue;  // empty voxel: not expanded (voxelgrid.go:57-60)
    for (uint32_t e = g->cell_start[(size_t)c]; e < g->cell_start[(size_t)c + 1]; e++) {
      if (k < cap) out[k] = g->idx_sorted[e];
      k++;
    }
    for (int dx = -1; dx <= 1; dx++)
      for (int dy = -1; dy <= 1; dy++)
        for (int dz = -1; dz <= 1; dz++) {
          if (dx == 0 && dy == 0 && dz == 0) continue;
          const int64_t n[3] = {cur.v[0] + dx, cur.v[1] + dy, cur.v[2] + dz};
          int64_t a2;
          if (!addr_of(n, &a2) || is_searched(a2)) continue;
          next.push_back(P3{{n[0], n[1], n[2]}});
        }
  }
  *count = k;
  return PCGX_OK;
}

// Region-growing components (regiongrowing.go:23-56 for every seed at once): comp[i] = smallest id
// of the set of points reachable from i through steps shorter than max_range between points of
// i's property value.  labels: Uint32At(id) for id in [0, Len()).
extern "C" pcgx_status pcgx_region_growing_components(const pcgx_kdtree *t, const uint32_t *labels, float max_range,
                                                      int64_t *comp) {
  PCGX_API_LOCK();
  if (!t || !labels || !comp) return fail(PCGX_E_INVALID, "pcgx_region_growing_components: NULL argument");
  PCGX_TRY(ensure_init());
  const int64_t n = t->n;  // ids of the accessor, also after DeletePoint
  bool empty = false;
  PCGX_TRY(resolve_tree(t, &t, &empty));
  if (empty) {
    for (int64_t i = 0; i < n; i++) comp[i] = i;
    return PCGX_OK;
  }
  hipStream_t st = ctx().stream;
  Arena &ar = ctx().arena;
  PCGX_TRY(ar.begin(st));
  uint32_t *d_labels = nullptr, *d_parent = nullptr, *d_root = nullptr;
  PCGX_TRY(ar.alloc_n((size_t)n, &d_labels));
  PCGX_TRY(ar.alloc_n((size_t)n, &d_parent));
  PCGX_TRY(ar.alloc_n((size_t)n, &d_root));
  PCGX_HIP_TRY(hipMemcpyAsync(d_labels, labels, (size_t)n * 4, hipMemcpyHostToDevice, st));
  const unsigned nb = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(uf_init_kernel, dim3(nb), dim3(256), 0, st, d_parent, n);
  const TreeView tv = t->view();
  const uint32_t slots = 1u << tv.depth;
  const char *walk_env = getenv("PCGX_RANGE_WALK");  // (as in range.hip: the walk although the handle has a grid)
  const bool on_grid = t->grid_ok && t->n == n && !(walk_env && *walk_env && *walk_env != '0');
  if (on_grid) {
    uint32_t *d_min = nullptr;
    PCGX_TRY(ar.alloc_n((size_t)n, &d_min));
    PCGX_HIP_TRY(hipMemsetAsync(d_min, 0xFF, (size_t)n * 4, st));
    hipLaunchKernelGGL(rg_grid_kernel<true>, dim3(nb), dim3(256), 0, st, t->grid, n, (const uint32_t *)d_labels,
                       max_range * max_range, d_parent);
    hipLaunchKernelGGL(uf_jump_kernel, dim3(nb), dim3(256), 0, st, d_parent, n);
    hipLaunchKernelGGL(rg_grid_kernel<false>, dim3(nb), dim3(256), 0, st, t->grid, n, (const uint32_t *)d_labels,
                       max_range * max_range, d_parent);
    hipLaunchKernelGGL(uf_jump_kernel, dim3(nb), dim3(256), 0, st, d_parent, n);
    hipLaunchKernelGGL(rg_min_id_kernel, dim3(nb), dim3(256), 0, st, t->grid, n, (const uint32_t *)d_parent, d_min);
    hipLaunchKernelGGL(rg_name_kernel, dim3(nb), dim3(256), 0, st, t->grid, n, (const uint32_t *)d_parent,
                       (const uint32_t *)d_min, d_root);
  } else {
    hipLaunchKernelGGL(rg_union_kernel, dim3((slots + kRangeWalkBlock - 1) / kRangeWalkBlock), dim3(kRangeWalkBlock),
                       walk_stack_bytes(tv, kRangeWalkBlock), st, tv, (const uint32_t *)d_labels, max_range * max_range,
                       d_parent);
  }
  if (!on_grid) hipLaunchKernelGGL(uf_flatten_kernel, dim3(nb), dim3(256), 0, st, d_parent, n, d_root);
  PCGX_HIP_TRY(hipGetLastError());
  std::vector<uint32_t> h((size_t)n);
  PCGX_HIP_TRY(hipMemcpyAsync(h.data(), d_root, (size_t)n * 4, hipMemcpyDeviceToHost, st));
  PCGX_HIP_TRY(hipStreamSynchronize(st));
  for (int64_t i = 0; i < n; i++) comp[i] = h[(size_t)i];
  return PCGX_OK;
}

// RegionGrowing.Segment(p, maxRange) from the components of the same max_range:
// neighbours = Range(p, maxRange); targetVal = property of the nearest (regiongrowing.go:26-31);
// result = every point of the components of the neighbours that carry targetVal, ascending id.
extern "C" pcgx_status pcgx_region_growing_segment(const pcgx_kdtree *t, const uint32_t *labels, const int64_t *comp,
                                                   const float p[3], float max_range, int64_t *out, int64_t cap,
                                                   int64_t *count) {
  PCGX_API_LOCK();
  if (!t || !labels || !comp || !p || !count || cap < 0 || (cap > 0 && !out))
    return fail(PCGX_E_INVALID, "pcgx_region_growing_segment: bad argument");
  *count = 0;
  int64_t nn = 0;
  PCGX_TRY(pcgx_kdtree_range_count(t, p, 1, max_range, &nn));
  if (nn == 0) return PCGX_OK;
  std::vector<int64_t> ids((size_t)nn);
  std::vector<float> dsq((size_t)nn);
  const int64_t offs[2] = {0, nn};
  PCGX_TRY(pcgx_kdtree_range_fill(t, p, 1, max_range, offs, ids.data(), dsq.data()));
  const uint32_t target = labels[ids[0]];
  std::vector<int64_t> roots;
  for (int64_t j : ids)
    if (labels[j] == target) roots.push_back(comp[j]);
  std::sort(roots.begin(), roots.end());
  roots.erase(std::unique(roots.begin(), roots.end()), roots.end());
  int64_t k = 0;
  for (int64_t i = 0; i < t->n; i++) {
    if (labels[i] != target || !std::binary_search(roots.begin(), roots.end(), comp[i])) continue;
    if (k < cap) out[k] = i;
    k++;
  }
  *count = k;
  return PCGX_OK;
}

// RegionGrowing.Segment(p, maxRange) with the ids in the reference's own order: its FIFO search
// (regiongrowing.go:23-56) level by level -- the Range() calls of one BFS level are ONE batch on
// the device (each list sorted by DistSq like KDTree.Range), the queue bookkeeping (toVisit, append
// order) runs on the host exactly as the Go code does it.  Same set as pcgx_region_growing_segment.
extern "C" pcgx_status pcgx_region_growing_segment_bfs(const pcgx_kdtree *t, const uint32_t *labels, const float p[3],
                                                       float max_range, int64_t *out, int64_t cap, int64_t *count) {
  PCGX_API_LOCK();
  if (!t || !labels || !p || !count || cap < 0 || (cap > 0 && !out))
    return fail(PCGX_E_INVALID, "pcgx_region_growing_segment_bfs: bad argument");
  *count = 0;
  auto range_batch = [&](const std::vector<float> &q, std::vector<int64_t> &offs, std::vector<int64_t> &ids) -> pcgx_status {
    const int64_t nq = (int64_t)q.size() / 3;
    std::vector<int64_t> counts((size_t)nq);
    PCGX_TRY(pcgx_kdtree_range_count(t, q.data(), nq, max_range, counts.data()));
    offs.assign((size_t)nq + 1, 0);
    for (int64_t i = 0; i < nq; i++) offs[(size_t)i + 1] = offs[(size_t)i] + counts[(size_t)i];
    ids.assign((size_t)offs[(size_t)nq], 0);
    std::vector<float> dsq((size_t)offs[(size_t)nq]);
    if (offs[(size_t)nq] > 0)
      PCGX_TRY(pcgx_kdtree_range_fill(t, q.data(), nq, max_range, offs.data(), ids.data(), dsq.data()));
    return PCGX_OK;
  };
  std::vector<float> q(p, p + 3);
  std::vector<int64_t> offs, ids;
  PCGX_TRY(range_batch(q, offs, ids));
  if (ids.empty()) return PCGX_OK;  // regiongrowing.go:27-29
  const uint32_t target = labels[ids[0]];  // :31
  std::vector<uint8_t> to_visit((size_t)t->n, 0);
  std::vector<int64_t> frontier(ids.begin(), ids.end());  // `next`, in append order
  for (int64_t id : frontier) to_visit[(size_t)id] = 1;
  int64_t k = 0;
  while (!frontier.empty()) {
    // the ids of this level that are kept (:41-45), in queue order; their Range() lists in one batch
    std::vector<int64_t> kept;
    for (int64_t id : frontier)
      if (labels[id] == target) kept.push_back(id);
    q.resize(kept.size() * 3);
    for (size_t j = 0; j < kept.size(); j++) memcpy(&q[3 * j], &t->points[3 * (size_t)kept[j]], 12);  // Vec3At(id), :46
    std::vector<int64_t> next;
    if (!kept.empty()) {
      PCGX_TRY(range_batch(q, offs, ids));
      for (size_t j = 0; j < kept.size(); j++) {
        if (k < cap) out[k] = kept[j];
        k++;
        for (int64_t e = offs[j]; e < offs[j + 1]; e++) {
          const int64_t nb = ids[(size_t)e];
          if (!to_visit[(size_t)nb]) {  // :48-51
            to_visit[(size_t)nb] = 1;
            next.push_back(nb);
          }
        }
      }
    }
    frontier.swap(next);
  }
  *count = k;
  return PCGX_OK;
}
