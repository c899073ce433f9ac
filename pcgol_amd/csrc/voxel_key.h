// voxel_key.h -- the VoxelGrid filter's grid and the cell of a point, shared by the radix path (voxel.hip) and the
// bucket path (voxel_bucket.hip).  Reference: pc/filter/voxelgrid/voxelgrid.go:45-79,137-151.
#pragma once
#include <math.h>

#include "pcgx_internal.h"

namespace pcgx {

struct VoxelParams {
  float vmin[3];
  float leaf[3];
  int64_t xs, ys;      // strides of the dense index (voxelgrid.go:137,151)
  int64_t n_voxels;    // (xs+1)(ys+1)(zs+1) (voxelgrid.go:138)
  // chunked mode (voxelgrid.go:49-79)
  int32_t chunked;
  float cs[3];         // clamped chunk size in metres
  int64_t nx, ny, n_chunks;
  // chunked mode with (chunk id, cell) fitting 32 bits: ONE sort key = cid * n_voxels + cell
  int32_t combined, pad_;
};

__device__ __forceinline__ float ld_f32(const uint8_t *p) {
  float v;
  __builtin_memcpy(&v, p, 4);  // records may be 1-byte aligned (pc/iterator.go:71-76)
  return v;
}

__device__ __forceinline__ void chunk_origin(const VoxelParams &vp, uint32_t cid, float o[3]) {
  // cid2xyz + vMin.Add(cp.ElementMul(chunkSize))  (voxelgrid.go:69-75,109-110)
  // (cid, nx, ny < 2^32, voxel_grid_params: the reference's int arithmetic in 32-bit registers)
  const uint32_t nx = (uint32_t)vp.nx, ny = (uint32_t)vp.ny;
  const uint32_t c2 = cid / nx;
  const uint32_t x = cid - c2 * nx;
  const uint32_t z = c2 / ny;
  const uint32_t y = c2 - z * ny;
  o[0] = vp.vmin[0] + (float)x * vp.cs[0];
  o[1] = vp.vmin[1] + (float)y * vp.cs[1];
  o[2] = vp.vmin[2] + (float)z * vp.cs[2];
}

// (int64_t)q of a float32 quotient that lies in (-1, 2^31): the truncation is 0 for the negative ones, the value fits a
// register -- every point of a cloud whose min / max the grid was made from, NaN and a negative vMin (non-chunked)
// aside.  The general conversion (float -> int64: a dozen instructions) and the 64-bit index arithmetic behind it
// were two thirds of the key kernels' instructions: those kernels were bound by vector issue, not by memory.
__device__ __forceinline__ bool cell_u32(float q, uint32_t &v) {
  const bool ok = q > -1.0f && q < 2147483648.0f;
  v = (uint32_t)(int32_t)(ok ? q : 0.0f);
  return ok;
}

// The cell of a point in the reference's arithmetic (float32 subtract and divide, truncation, the xs / ys strides of
// voxelgrid.go:137-151; chunked mode: the chunk first, voxelgrid.go:76-79): the sort key (cell, or chunk id and cell
// in one word), the chunk id and the cell by themselves.  bad: the Go code would panic (index out of range); the
// key is then 0 (an out-of-range chunk) or carries cell 0.  (The general form: any float, 64-bit indices.)
__device__ inline uint32_t voxel_key_xyz_any(const float pt[3], const VoxelParams &vp, uint32_t &cid_out, uint32_t &ka_out, bool &bad) {
  float origin[3] = {vp.vmin[0], vp.vmin[1], vp.vmin[2]};
  uint32_t cid = 0;
  bad = false;
  cid_out = 0;
  ka_out = 0;
  if (vp.chunked) {
    const float q0 = pt[0] - vp.vmin[0], q1 = pt[1] - vp.vmin[1], q2 = pt[2] - vp.vmin[2];
    const int64_t cx = (int64_t)(q0 / vp.cs[0]), cy = (int64_t)(q1 / vp.cs[1]), cz = (int64_t)(q2 / vp.cs[2]);
    const int64_t c = ((cz * vp.ny) + cy) * vp.nx + cx;
    if (c < 0 || c >= vp.n_chunks) {  // nIndices[cid] would panic
      bad = true;
      return 0u;
    }
    cid = (uint32_t)c;
    chunk_origin(vp, cid, origin);
    cid_out = cid;
  }
  const float p0 = pt[0] - origin[0], p1 = pt[1] - origin[1], p2 = pt[2] - origin[2];
  const int64_t x = (int64_t)(p0 / vp.leaf[0]), y = (int64_t)(p1 / vp.leaf[1]), z = (int64_t)(p2 / vp.leaf[2]);
  const int64_t a = x + vp.xs * (y + vp.ys * z);
  uint32_t ka = 0;
  if (a < 0 || a >= vp.n_voxels) bad = true;  // f.voxels[a] would panic
  else ka = (uint32_t)a;
  ka_out = ka;
  return vp.combined ? cid * (uint32_t)vp.n_voxels + ka : ka;
}

// The same result through 32-bit registers where every quotient allows it (cell_u32) and no product leaves 32 bits on
// its way; the general form for the lanes where one does.  xs, ys, nx, ny, n_chunks, n_voxels are all below 2^32
// (voxel_grid_params).
__device__ __forceinline__ uint32_t voxel_key_xyz(const float pt[3], const VoxelParams &vp, uint32_t &cid_out, uint32_t &ka_out,
                                                  bool &bad) {
  float origin[3] = {vp.vmin[0], vp.vmin[1], vp.vmin[2]};
  uint32_t cid = 0;
  bool fast = true;
  if (vp.chunked) {
    const float q0 = pt[0] - vp.vmin[0], q1 = pt[1] - vp.vmin[1], q2 = pt[2] - vp.vmin[2];
    uint32_t cx, cy, cz;
    fast = cell_u32(q0 / vp.cs[0], cx) & cell_u32(q1 / vp.cs[1], cy) & cell_u32(q2 / vp.cs[2], cz);
    // cid2xyz(cid) is (cx, cy, cz) again when cx < nx and cy < ny; cz < 2^31 and ny, nx < 2^32: no product leaves 64 bits
    const uint64_t c = ((uint64_t)cz * (uint32_t)vp.ny + cy) * (uint32_t)vp.nx + cx;
    fast = fast && cx < (uint32_t)vp.nx && cy < (uint32_t)vp.ny && (c >> 32) == 0ull;
    if (fast) {
      if (c >= (uint64_t)vp.n_chunks) {  // nIndices[cid] would panic
        bad = true;
        cid_out = 0;
        ka_out = 0;
        return 0u;
      }
      cid = (uint32_t)c;
      origin[0] = vp.vmin[0] + (float)cx * vp.cs[0];  // vMin.Add(cp.ElementMul(chunkSize))  (voxelgrid.go:69-75,109-110)
      origin[1] = vp.vmin[1] + (float)cy * vp.cs[1];
      origin[2] = vp.vmin[2] + (float)cz * vp.cs[2];
    }
  }
  if (fast) {
    const float p0 = pt[0] - origin[0], p1 = pt[1] - origin[1], p2 = pt[2] - origin[2];
    uint32_t x, y, z;
    fast = cell_u32(p0 / vp.leaf[0], x) & cell_u32(p1 / vp.leaf[1], y) & cell_u32(p2 / vp.leaf[2], z);
    const uint64_t t = (uint64_t)(uint32_t)vp.ys * z + y;                 // < 2^63
    fast = fast && (t >> 32) == 0ull;
    if (fast) {
      const uint64_t a = (uint64_t)(uint32_t)vp.xs * (uint32_t)t + x;     // < 2^64
      bad = a >= (uint64_t)vp.n_voxels;                                    // f.voxels[a] would panic
      const uint32_t ka = bad ? 0u : (uint32_t)a;
      cid_out = cid;
      ka_out = ka;
      return vp.combined ? cid * (uint32_t)vp.n_voxels + ka : ka;
    }
  }
  return voxel_key_xyz_any(pt, vp, cid_out, ka_out, bad);
}

// ---- the grid of a call, from the cloud's min / max (voxelgrid.go:45-62,137-138), in the reference's float32
// arithmetic.  Host and device run the SAME code (the library is built with -ffp-contract=off, float32 divides are
// correctly rounded on both): the bucket path makes its plan on the device, behind the min/max pass, without the host
// in between; the radix path and the error messages use the host's.  Returns 0, or 1 (the chunk grid is not
// addressable) / 2 (the dense grid is empty or exceeds 2^32 cells).
__host__ __device__ inline int voxel_grid_params(const float mm6[6], const float leaf[3], const int32_t chunk[3], VoxelParams &vp) {
  vp = VoxelParams{};
  const float *vmin = mm6, *vmax = mm6 + 3;
  for (int k = 0; k < 3; k++) {
    vp.vmin[k] = vmin[k];
    vp.leaf[k] = leaf[k];
  }
  float size[3];
  if ((int64_t)chunk[0] * chunk[1] * chunk[2] == 0) {
    for (int k = 0; k < 3; k++) size[k] = vmax[k];  // sic (voxelgrid.go:46)
    vp.chunked = 0;
    vp.nx = vp.ny = vp.n_chunks = 1;
  } else {
    float ext[3];
    for (int k = 0; k < 3; k++) {
      ext[k] = vmax[k] - vmin[k];
      vp.cs[k] = leaf[k] * (float)chunk[k];
    }
    for (int k = 0; k < 3; k++)
      if (vp.cs[k] > ext[k] + leaf[k]) vp.cs[k] = ext[k] + leaf[k];
    vp.chunked = 1;
    vp.nx = (int64_t)(ext[0] / vp.cs[0]) + 1;
    vp.ny = (int64_t)(ext[1] / vp.cs[1]) + 1;
    const int64_t nz = (int64_t)(ext[2] / vp.cs[2]) + 1;
    vp.n_chunks = nz;  // (for the message, should the grid not be addressable)
    if (vp.nx <= 0 || vp.ny <= 0 || nz <= 0 || (double)vp.nx * (double)vp.ny * (double)nz >= 4294967296.0) return 1;
    vp.n_chunks = vp.nx * vp.ny * nz;
    for (int k = 0; k < 3; k++) size[k] = vp.cs[k];
  }
  const int64_t xs = (int64_t)(size[0] / leaf[0]), ys = (int64_t)(size[1] / leaf[1]), zs = (int64_t)(size[2] / leaf[2]);
  const double nv = ((double)xs + 1) * ((double)ys + 1) * ((double)zs + 1);
  vp.xs = xs;
  vp.ys = ys;
  vp.n_voxels = zs;  // (for the message)
  if (xs < 0 || ys < 0 || zs < 0 || !(nv >= 1.0) || nv >= 4294967296.0) return 2;
  vp.n_voxels = (xs + 1) * (ys + 1) * (zs + 1);
  return 0;
}

// the host's form: the reference's panics as errors, in words
inline pcgx_status voxel_grid_params_or_fail(const float mm6[6], const float leaf[3], const int32_t chunk[3], VoxelParams &vp) {
  const int rc = voxel_grid_params(mm6, leaf, chunk, vp);
  if (rc == 1)
    return fail(PCGX_E_OUT_OF_RANGE, "voxel filter: chunk grid %lld x %lld x %lld is not addressable",
                (long long)vp.nx, (long long)vp.ny, (long long)vp.n_chunks);
  if (rc == 2)
    return fail(PCGX_E_OUT_OF_RANGE,
                "voxel filter: dense grid (%lld+1)(%lld+1)(%lld+1) is empty or exceeds 2^32 cells "
                "(the reference would panic or need >128 GiB)",
                (long long)vp.xs, (long long)vp.ys, (long long)vp.n_voxels);
  return PCGX_OK;
}

__host__ __device__ inline int voxel_bits_for(int64_t count) {  // bits needed for values in [0, count)
  int b = 0;
  while (b < 63 && ((int64_t)1 << b) < count) b++;
  return b;
}

// (chunk id, cell) in one 32-bit key, cid * n_voxels + cell -- dense: no unused values between the chunks' cell ranges,
// so the key is as short as it can be and the bucket path's buckets are evenly filled.  One stable sort gives the
// reference's output order (chunks ascending, cells ascending inside a chunk) without a second sort and its gathers.
// Sets vp.combined where that fits; returns the sort key's bits, *two_level: cell and chunk id are sorted one after
// the other (radix path only).
__host__ __device__ inline int voxel_key_layout(VoxelParams &vp, bool force_two_sorts, bool *two_level) {
  *two_level = vp.chunked && vp.n_chunks > 1;
  int key_bits = voxel_bits_for(vp.n_voxels);
  if (*two_level && (double)vp.n_chunks * (double)vp.n_voxels <= 4294967296.0 && !force_two_sorts) {
    vp.combined = 1;
    key_bits = voxel_bits_for(vp.n_chunks * vp.n_voxels);
    *two_level = false;
  }
  return key_bits;
}

// ---- the bucket path's plan (voxel_bucket.hip) ----------------------------------------------------------------------
constexpr int kVbMaxLowBits = 10;      // cells per bucket <= 1024
constexpr int kVbMaxBucketBits = 16;   // two digits of <= 8 bits
constexpr int kVbMaxBuckets = 1 << kVbMaxBucketBits;
#ifndef PCGX_VB_CAP
#define PCGX_VB_CAP 1280
#endif
constexpr int kVbCap = PCGX_VB_CAP;           // points per bucket the bucket kernel holds in LDS (26 KB: six workgroups per CU -- a
                                       // workgroup is a chain of short dependent phases, what hides them is the other workgroups)

struct VbPlan {
  int32_t low_bits;      // s
  int32_t nbuckets;
  int32_t d_bits[2];     // digits of the bucket number, low digit first; d_bits[1] == 0: one pass
  int32_t ntiles;
  int32_t dbg;           // measurement aid (PCGX_VOXEL_BUCKET_DBG): 1 no wait for the earlier buckets, 2 no cell phase (wrong output)
};

// What the device makes of the six floats, read by every kernel of the call and by the host at its end.
struct VoxelDevPlan {
  VoxelParams vp;
  VbPlan plan;
  int32_t status;        // 0 the bucket path runs; 1 / 2 voxel_grid_params' errors; 3 not a call for the bucket path
  int32_t key_bits;
  float mm6[6];
};

// bucket = key >> s: as many cells per bucket as keep an evenly filled bucket at 0.6 of the LDS tile (C3: the fullest
// of 6505 holds 1.35x the mean).  The keys that can occur are not more than the cells the cloud's own extent spans (the
// non-chunked grid is sized by vMax, voxelgrid.go:46; a chunked grid's last chunks are partly empty, and every chunk
// seam splits a cell).  false: not a call for the bucket path (two sorts, keys too wide, too few keys).
__host__ __device__ inline bool vb_choose_plan(int64_t n, const VoxelParams &vp, const float mm6[6], int key_bits, bool two_level,
                                               int ntiles, int dbg, VbPlan &plan) {
  plan = VbPlan{};
  const uint64_t key_range = vp.combined ? (uint64_t)vp.n_chunks * (uint64_t)vp.n_voxels : (uint64_t)vp.n_voxels;
  if (two_level || key_bits < 1 || key_bits > kVbMaxLowBits + kVbMaxBucketBits || key_range < 2) return false;
  uint64_t key_population = key_range;
  {
    double cells = 1.0;
    const int64_t per_axis_chunks[3] = {vp.chunked ? vp.nx : 0, vp.chunked ? vp.ny : 0, vp.chunked ? vp.n_chunks / (vp.nx * vp.ny) : 0};
    for (int k = 0; k < 3; k++) {
      const double ext = (double)mm6[3 + k] - (double)mm6[k];
      cells *= (ext > 0.0 ? floor(ext / (double)vp.leaf[k]) : 0.0) + 1.0 + (double)per_axis_chunks[k];
    }
    if (cells >= 1.0 && cells < (double)key_population) key_population = (uint64_t)cells;
  }
  int s = key_bits < kVbMaxLowBits ? key_bits : kVbMaxLowBits;
  while (s > 0 && (double)n * (double)((uint64_t)1 << s) / (double)key_population > 0.6 * kVbCap) s--;
  const uint64_t nb = ((key_range - 1) >> s) + 1;
  int bb = 0;
  while (((uint64_t)1 << bb) < nb) bb++;
  if (s == 0 || bb < 1 || bb > kVbMaxBucketBits) return false;
  plan.low_bits = s;
  plan.nbuckets = (int32_t)nb;
  plan.d_bits[0] = bb <= 8 ? bb : bb - bb / 2;
  plan.d_bits[1] = bb - plan.d_bits[0];
  plan.ntiles = ntiles;
  plan.dbg = dbg;
  return true;
}

// The plan is made by the min/max launch's last workgroup (sort.hip, minmax_block_fold), right behind the six floats:
// what it needs of the call rides along in the launch's arguments.  dp == nullptr: no plan asked for.
struct VoxelPlanHook {
  float leaf[3];
  int32_t chunk[3];
  int32_t force_two_sorts, dbg, ntiles, grid;   // grid: the bucket kernel's; a plan with more buckets is not for this path
  int64_t n;
  VoxelDevPlan *dp;
  int32_t *head;        // flags, key-out-of-range, total (two words): zero, but flags = 8 when there is no plan
  uint32_t *zero;       // cleared by the launch's workgroups, a slice each (sample counts, the buckets' cell counts and starts)
  uint32_t zero_words;  // (a multiple of 4)
};

__device__ inline void voxel_plan_on_device(const float mm6[6], const VoxelPlanHook &h) {
  VoxelDevPlan d;
  for (int k = 0; k < 6; k++) d.mm6[k] = mm6[k];
  d.plan = VbPlan{};
  d.key_bits = 0;
  d.status = voxel_grid_params(d.mm6, h.leaf, h.chunk, d.vp);
  if (d.status == 0) {
    bool two_level = false;
    d.key_bits = voxel_key_layout(d.vp, h.force_two_sorts != 0, &two_level);
    if (!vb_choose_plan(h.n, d.vp, d.mm6, d.key_bits, two_level, h.ntiles, h.dbg, d.plan) || d.plan.nbuckets > h.grid) d.status = 3;
  }
  *h.dp = d;
  h.head[0] = d.status ? 8 : 0;  // 8: every kernel behind this one returns at once
  h.head[1] = h.head[2] = h.head[3] = 0;
}

// sort.hip: the min/max pass with the plan behind it
pcgx_status launch_minmax_with_plan(const void *d_data, int64_t n, int32_t stride, int32_t off, float *d_out6,
                                    const VoxelPlanHook &hook, hipStream_t st);

// voxel_bucket.hip: the bucket path of a filter call on one GPU (see there): min/max, the plan and every kernel behind
// it enqueued without the host in between.  *attempted false: not a call for this path by what the host knows (too
// small, PCGX_VOXEL_BUCKET=0) -- nothing was launched.  Otherwise dp_host holds what the device made of the cloud (its
// min / max among it) and *taken says whether the output is there; if not, the radix path does the call from
// dp_host->mm6.
pcgx_status voxel_bucket_filter(const void *d_data, int64_t n, int32_t stride, int32_t xyz_off, const float leaf[3],
                                const int32_t chunk[3], void *d_out, int64_t *out_n, bool *attempted, bool *taken,
                                VoxelDevPlan *dp_host, hipStream_t st);

}  // namespace pcgx
