// voxel_key.h -- the VoxelGrid filter's grid and the cell of a point, shared by the radix path (voxel.hip) and the
// bucket path (voxel_bucket.hip).  Reference: pc/filter/voxelgrid/voxelgrid.go:45-79,137-151.
#pragma once
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
  int64_t c = cid;
  const int64_t x = c % vp.nx;
  c = c / vp.nx;
  const int64_t y = c % vp.ny;
  const int64_t z = c / vp.ny;
  o[0] = vp.vmin[0] + (float)x * vp.cs[0];
  o[1] = vp.vmin[1] + (float)y * vp.cs[1];
  o[2] = vp.vmin[2] + (float)z * vp.cs[2];
}

// The cell of a point in the reference's arithmetic (float32 subtract and divide, truncation, the xs / ys strides of
// voxelgrid.go:137-151; chunked mode: the chunk first, voxelgrid.go:76-79): the sort key (cell, or chunk id and cell
// in one word), the chunk id and the cell by themselves.  bad: the Go code would panic (index out of range); the
// key is then 0 (an out-of-range chunk) or carries cell 0.
__device__ __forceinline__ uint32_t voxel_key_xyz(const float pt[3], const VoxelParams &vp, uint32_t &cid_out, uint32_t &ka_out,
                                                  bool &bad) {
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

// voxel_bucket.hip: the bucket path of a filter call on one GPU (see there); *taken false: the radix path does the call
pcgx_status voxel_bucket_filter(const void *d_data, int64_t n, int32_t stride, int32_t xyz_off, const VoxelParams &vp,
                                int key_bits, uint64_t key_range, uint64_t key_population, void *d_out, int64_t *out_n, bool *taken, hipStream_t st);

}  // namespace pcgx
