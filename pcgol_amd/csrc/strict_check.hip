// strict_check.hip -- the reference's sequential float32 sums by ONE wave, term after term: the plain dependent chain.
// Milliseconds per iteration at 1M pairs; kept as the on-device cross-check of the parallel evaluation (strict.hip:
// tests compare the two sum by sum, PCGX_SUMS_REFERENCE_CHAIN / pcgx_icp_session_set_strict(2)) and off the hot
// translation units.
#include "pcgx_internal.h"

namespace pcgx {

// STRICT mode: the evaluator's sums exactly as the reference forms them -- sequential float32
// additions over the pairs in target order (evaluator.go:122-145; Go evaluates them one pair after
// the other in a single goroutine).  A parallel reduction cannot reproduce those bits (float
// addition is not associative; at 1M pairs the reference's own rounding noise is ~1.6e-5 on the
// final transform), so: (1) icp_strict_terms_kernel forms the nine float32 terms of every target
// in parallel and stores them in the CALLER's target order (through pos_of), one row per
// component, plus a valid bit per target; (2) icp_strict_sums_kernel is ONE wave whose lane k
// streams row k and adds its terms one after the other (next block's 16-byte loads in flight
// while the dependent chain of 64 additions runs).  Milliseconds per iteration at 1M pairs
// against 0.07 ms for the float64 tree, hence opt-in (pcgx_icp_session_set_strict /
// PCGX_ICP_STRICT=1): bit-identical Evaluated and pose at any size.
__global__ __launch_bounds__(256) void icp_strict_terms_kernel(const float *__restrict__ tx, const float *__restrict__ ty,
                                                               const float *__restrict__ tz, int64_t nt, int64_t nt_pad,
                                                               const float4 *__restrict__ match,
                                                               const uint32_t *__restrict__ pos_of,
                                                               const IcpState *__restrict__ state, IcpKernelParams kp,
                                                               float *__restrict__ terms,
                                                               unsigned long long *__restrict__ valid_bits) {
  if (state->done) return;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // grid covers nt_pad (a multiple of 64)
  float m[16];
#pragma unroll
  for (int k = 0; k < 16; k++) m[k] = state->trans[k];
  const bool project = state->iter > 0;  // icp.go:27-30: the first Evaluate sees the raw target
  bool valid = false;
  // unmatched targets and the padding behind nt carry -0.0f: x + (-0.0f) == x for EVERY float x
  // (including both zeros), so the sequential sum needs no branch or select for them
  float t[9];
#pragma unroll
  for (int k = 0; k < 9; k++) t[k] = -0.0f;
  if (i < nt) {
    const uint32_t pos = pos_of[i];
    const float4 bp = match[pos];
    if (bp.w >= 0.0f) {  // correspondence.go:27-29
      valid = true;
      float x0 = tx[pos], y0 = ty[pos], z0 = tz[pos];
      if (project) {
        float px, py, pz;
        mat4_transform(m, x0, y0, z0, px, py, pz);
        x0 = px; y0 = py; z0 = pz;
      }
      const float x1 = bp.x, y1 = bp.y, z1 = bp.z;
      const float w = eval_weight_fn(kp.weight_fn, kp.weight_a, bp.w);  // evaluator.go:130
      t[0] = w * bp.w;
      t[1] = w * (x0 - x1);
      t[2] = w * (y0 - y1);
      t[3] = w * (z0 - z1);
      t[4] = w * (z0 * y1 - y0 * z1);
      t[5] = w * (x0 * z1 - z0 * x1);
      t[6] = w * (y0 * x1 - x0 * y1);
      t[7] = w * norm_sq3(x0, y0, z0);
      t[8] = w;
    }
  }
  if (i < nt_pad) {
#pragma unroll
    for (int k = 0; k < 9; k++) terms[(int64_t)k * nt_pad + i] = t[k];
  }
  const unsigned long long bal = __ballot(valid);
  if ((threadIdx.x & 63) == 0 && i < nt_pad) valid_bits[i >> 6] = bal;
}

// kFuseUpdate: thread 0 then runs the evaluate tail + pose update (single GPU).
template <bool kFuseUpdate>
__global__ __launch_bounds__(64) void icp_strict_sums_kernel(const float *__restrict__ terms,
                                                             const unsigned long long *__restrict__ valid_bits,
                                                             int64_t nt_pad, IcpState *__restrict__ state,
                                                             double *__restrict__ sums10, IcpKernelParams kp) {
  __shared__ double s_sums[S_COUNT];
  if (state->done) return;
  const int lane = threadIdx.x;
  const int64_t nblk = nt_pad >> 6;
  float acc = 0.0f;   // lanes 0..8: Value, G0..G5, DistRMS, sum of weights (float32, sequential)
  int64_t pairs = 0;  // lane 9
  if (lane < 9 && nblk > 0) {
    const float4 *row = reinterpret_cast<const float4 *>(terms + (int64_t)lane * nt_pad);
    // two register buffers used alternately: while one block's 64 additions run (a dependent
    // chain), the 16 loads of the block after the next are already in flight
    float4 a[16], b[16];
#pragma unroll
    for (int v = 0; v < 16; v++) a[v] = row[v];
    for (int64_t blk = 0; blk < nblk; blk += 2) {
      const int64_t b1 = blk + 1 < nblk ? blk + 1 : blk, b2 = blk + 2 < nblk ? blk + 2 : blk;
#pragma unroll
      for (int v = 0; v < 16; v++) b[v] = row[b1 * 16 + v];
#pragma unroll
      for (int v = 0; v < 16; v++) acc = (((acc + a[v].x) + a[v].y) + a[v].z) + a[v].w;
#pragma unroll
      for (int v = 0; v < 16; v++) a[v] = row[b2 * 16 + v];
      if (blk + 1 < nblk) {
#pragma unroll
        for (int v = 0; v < 16; v++) acc = (((acc + b[v].x) + b[v].y) + b[v].z) + b[v].w;
      }
    }
  } else if (lane == 9) {
    for (int64_t blk = 0; blk < nblk; blk++) pairs += (int64_t)__popcll(valid_bits[blk]);
  }
  // component order of sums10: Value, G0..G5, DistRMS, Weight, Pairs
  if (lane < 9) {
    const int slot = lane == 0 ? S_VALUE : (lane <= 6 ? S_G0 + lane - 1 : (lane == 7 ? S_DIST_RMS : S_WEIGHT));
    sums10[slot] = (double)acc;
    s_sums[slot] = (double)acc;
  } else if (lane == 9) {
    sums10[S_PAIRS] = (double)pairs;
    s_sums[S_PAIRS] = (double)pairs;
  }
  if (kFuseUpdate) {
    __syncthreads();
    if (lane == 0) icp_update_step(state, s_sums, kp);
  }
}


pcgx_status strict_check_enqueue(const float *d_xyz, int64_t nt, int64_t nt_pad, const float4 *match, const uint32_t *pos_of,
                                 IcpState *state, const IcpKernelParams &kp, float *d_terms, unsigned long long *d_valid,
                                 double *d_sums, bool fuse_update, hipStream_t st) {
  if (nt_pad > 0)
    hipLaunchKernelGGL(icp_strict_terms_kernel, dim3((unsigned)(nt_pad / 256 + 1)), dim3(256), 0, st, d_xyz, d_xyz + nt, d_xyz + 2 * nt,
                       nt, nt_pad, match, pos_of, (const IcpState *)state, kp, d_terms, d_valid);
  if (fuse_update)
    hipLaunchKernelGGL(icp_strict_sums_kernel<true>, dim3(1), dim3(64), 0, st, (const float *)d_terms,
                       (const unsigned long long *)d_valid, nt_pad, state, d_sums, kp);
  else
    hipLaunchKernelGGL(icp_strict_sums_kernel<false>, dim3(1), dim3(64), 0, st, (const float *)d_terms,
                       (const unsigned long long *)d_valid, nt_pad, state, d_sums, kp);
  PCGX_HIP_TRY(hipGetLastError());
  return PCGX_OK;
}

}  // namespace pcgx
