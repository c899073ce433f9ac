// pcgx_math.h -- float32 pose arithmetic shared by host code and device kernels.
//
// Restates (product side; the test oracle has its own, separate restatement)
// the O(1) per-iteration arithmetic of the reference's ICP loop exactly as Go
// evaluates it on amd64: float32, left to right, no FMA (build with
// -ffp-contract=off), float64 only where the Go code converts
// (math.Sqrt/Sin/Cos).  Paths are relative to the reference repository root.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define PCGX_HD __host__ __device__ inline
#else
#define PCGX_HD inline
#endif

namespace pcgx {

struct Mat4 {
  float m[16];  // column-major: m[4*col + row]   (mat/mat4.go:8-10)
};

// mat/transform.go:7-14
PCGX_HD Mat4 mat4_translate(float x, float y, float z) {
  Mat4 r;
  for (int i = 0; i < 16; i++) r.m[i] = (i % 5 == 0) ? 1.0f : 0.0f;
  r.m[12] = x;
  r.m[13] = y;
  r.m[14] = z;
  return r;
}

// mat/mat4.go:16-28: out[4j+i] = sum_k m[4k+i]*a[4j+k], accumulated from 0 in k order
PCGX_HD Mat4 mat4_mul(const Mat4 &m, const Mat4 &a) {
  Mat4 o;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      float s = 0.0f;
      for (int k = 0; k < 4; k++) s = s + m.m[4 * k + i] * a.m[4 * j + k];
      o.m[4 * j + i] = s;
    }
  return o;
}

// mat/mat4.go:30-36 and :38-44
PCGX_HD Mat4 mat4_factor(const Mat4 &m, float f) {
  Mat4 o;
  for (int i = 0; i < 16; i++) o.m[i] = m.m[i] * f;
  return o;
}
PCGX_HD Mat4 mat4_add(const Mat4 &m, const Mat4 &a) {
  Mat4 o;
  for (int i = 0; i < 16; i++) o.m[i] = m.m[i] + a.m[i];
  return o;
}

// mat/mat4.go:130-137 Mat4.Transform: projective, w = 1/(m3 x + m7 y + m11 z + m15)
PCGX_HD void mat4_transform(const float *m, float x, float y, float z, float &ox, float &oy,
                            float &oz) {
  float w = 1.0f / (((m[3] * x + m[7] * y) + m[11] * z) + m[15]);
  ox = (((m[0] * x + m[4] * y) + m[8] * z) + m[12]) * w;
  oy = (((m[1] * x + m[5] * y) + m[9] * z) + m[13]) * w;
  oz = (((m[2] * x + m[6] * y) + m[10] * z) + m[14]) * w;
}

// mat/vec3.go:18-20
PCGX_HD float norm_sq3(float a, float b, float c) { return (a * a + b * b) + c * c; }

// icp/rodrigues.go:11-33
PCGX_HD Mat4 rodrigues_to_rotation(float v0, float v1, float v2) {
  float ang = (float)sqrt((double)norm_sq3(v0, v1, v2));  // Vec3.Norm, mat/vec3.go:22-24
  Mat4 r;
  for (int i = 0; i < 16; i++) r.m[i] = 0.0f;
  r.m[1] = v2;  r.m[2] = -v1;
  r.m[4] = -v2; r.m[6] = v0;
  r.m[8] = v1;  r.m[9] = -v0;
  Mat4 id = mat4_translate(0.0f, 0.0f, 0.0f);
  float f0, f1;
  if (ang < 0.1f) {
    f0 = 1.0f;
    f1 = 0.5f;
  } else {
    f0 = (float)sin((double)ang) / ang;
    f1 = (float)(1.0 - cos((double)ang)) / (ang * ang);
  }
  return mat4_add(mat4_add(id, mat4_factor(r, f0)), mat4_factor(mat4_mul(r, r), f1));
}

// The 10 per-iteration sums: order fixed across the library.
enum { S_VALUE = 0, S_G0 = 1, S_DIST_RMS = 7, S_WEIGHT = 8, S_PAIRS = 9, S_COUNT = 10 };

struct Evaluated {
  float value;
  float gradient[6];
  float dist_rms;
  int64_t num_pairs;
};

// icp/evaluator.go:156-186: normalise by 1/sum(w) (only if > 1), gradient by
// 2f, DistRMS = sqrt(.), rotation limiter.  The sums arrive as float64
// (device reduction) and are rounded to float32 first: from here on the
// arithmetic is the reference's float32 sequence.
PCGX_HD void finish_evaluate(const double *sums, Evaluated &ev) {
  float value = (float)sums[S_VALUE];
  float sum_weight = (float)sums[S_WEIGHT];
  float dist_rms = (float)sums[S_DIST_RMS];
  float g[6];
  for (int i = 0; i < 6; i++) g[i] = (float)sums[S_G0 + i];
  float f = 1.0f;
  if (sum_weight > 1.0f) f = 1.0f / sum_weight;
  value = value * f;
  float f2 = 2.0f * f;
  for (int i = 0; i < 6; i++) g[i] = g[i] * f2;
  dist_rms = (float)sqrt((double)(dist_rms * f));
  float rot_limit = 1.0f;
  float dist = (float)sqrt((double)value);
  for (int i = 3; i < 6; i++) {
    float d = g[i] * dist_rms;
    if (d < 0.0f) d = -d;
    if (dist < d) {
      float l = dist / d;
      if (rot_limit > l) rot_limit = l;
    }
  }
  for (int i = 3; i < 6; i++) g[i] = g[i] * rot_limit;
  ev.value = value;
  for (int i = 0; i < 6; i++) ev.gradient[i] = g[i];
  ev.dist_rms = dist_rms;
  ev.num_pairs = (int64_t)sums[S_PAIRS];
}

// PointToPointEvaluator.WeightFn built-ins (include/pcgx.h PCGX_WEIGHT_*): float32, the Go
// expression's order of operations, math.Sqrt in float64 where Go would call it.
PCGX_HD float eval_weight_fn(int32_t kind, float a, float d) {
  switch (kind) {
    case 1: return a;
    case 2: return 1.0f / (a + d);
    case 3: {
      if (d <= a) return 1.0f;
      const float q = a / d;
      return (float)sqrt((double)q);
    }
    case 4: {
      if (!(d < a)) return 0.0f;
      const float u = 1.0f - d / a;
      return u * u;
    }
    default: return 1.0f;  // DefaultEvaluateWeightFn (evaluator.go:21-23)
  }
}

struct UpdaterParams {
  float weight[6];
  float threshold[6];
  int32_t max_iteration;
};

// icp/updater.go:15-37 factory defaults: all-zero vectors / zero count select
// 0.3 / 0.01 / 20.
PCGX_HD UpdaterParams resolve_updater(const float *weight, const float *threshold,
                                      int32_t max_iteration) {
  UpdaterParams u;
  bool wz = true, tz = true;
  for (int i = 0; i < 6; i++) {
    if (weight[i] != 0.0f) wz = false;
    if (threshold[i] != 0.0f) tz = false;
  }
  for (int i = 0; i < 6; i++) {
    u.weight[i] = wz ? 0.3f : weight[i];
    u.threshold[i] = tz ? 0.01f : threshold[i];
  }
  u.max_iteration = max_iteration == 0 ? 20 : max_iteration;
  return u;
}

// icp/updater.go:44-71 gradientDescentUpdater.Update.  Returns converged.
PCGX_HD bool gradient_descent_update(const UpdaterParams &u, int32_t &iter, const float *g,
                                     Mat4 &trans) {
  bool flat = true;
  for (int j = 0; j < 6; j++) {
    if (g[j] < -u.threshold[j] || u.threshold[j] < g[j]) {
      flat = false;
      break;
    }
  }
  if (flat) return true;
  float factor_iter = -(1.0f - ((float)iter / (float)u.max_iteration));
  float d[6];
  for (int j = 0; j < 6; j++) d[j] = (factor_iter * u.weight[j]) * g[j];
  Mat4 delta_trans = mat4_translate(d[0], d[1], d[2]);
  Mat4 delta_rot = rodrigues_to_rotation(d[3], d[4], d[5]);
  trans = mat4_mul(delta_trans, mat4_mul(delta_rot, trans));
  iter = iter + 1;
  return iter >= u.max_iteration;
}

// ---------------------------------------------------------------------------
// Point-to-plane / Gauss-Newton extension.  NOT in the reference:
// pcgol only declares the slots (Evaluated.Hessian mat.Mat6, Evaluator.HasHessian,
// icp/evaluator.go:28,35,76; mat/mat6.go:3).  Conventions follow the reference's
// point-to-point evaluator so the two are interchangeable behind icp.Evaluator:
// the pose increment is applied on the left, p' = p + t + w x p (updater.go:65-68),
// parameters ordered {t0,t1,t2,w0,w1,w2} like Evaluated.Gradient (evaluator.go:135-142).
//   residual  r = n . (pt - pb)            (n: unit normal of the matched base point)
//   Jacobian  J = {n, pt x n}
//   30 sums   {sum r^2, sum J r [6], sum J J^T upper triangle row-major [21], sum w, pairs}
enum { P_VALUE = 0, P_G0 = 1, P_H0 = 7, P_WEIGHT = 28, P_PAIRS = 29, P_COUNT = 30 };

struct EvaluatedPlane {
  float value;         // mean squared point-to-plane distance
  float gradient[6];   // d value / d params = 2/sum(w) * sum J r
  float hessian[36];   // Gauss-Newton Hessian 2/sum(w) * sum J J^T, symmetric (Mat6 slot)
  int64_t num_pairs;
};

// One pair's terms, every product formed in float32 in this fixed expression order
// (the oracle restates the same sequence); the caller accumulates them in float64.
PCGX_HD void plane_terms(float x0, float y0, float z0, float x1, float y1, float z1, float nx, float ny,
                         float nz, float J[6], float &r) {
  const float dx = x0 - x1, dy = y0 - y1, dz = z0 - z1;
  r = (nx * dx + ny * dy) + nz * dz;
  J[0] = nx;
  J[1] = ny;
  J[2] = nz;
  J[3] = y0 * nz - z0 * ny;
  J[4] = z0 * nx - x0 * nz;
  J[5] = x0 * ny - y0 * nx;
}

// Normalisation with the reference's convention (evaluator.go:156-163): f = 1/sum(w) only if > 1.
PCGX_HD void finish_evaluate_plane(const double *sums, EvaluatedPlane &ev) {
  const double sw = sums[P_WEIGHT];
  const double f = sw > 1.0 ? 1.0 / sw : 1.0;
  ev.value = (float)(sums[P_VALUE] * f);
  for (int i = 0; i < 6; i++) ev.gradient[i] = (float)(sums[P_G0 + i] * (2.0 * f));
  int k = 0;
  for (int a = 0; a < 6; a++)
    for (int b = a; b < 6; b++) {
      const float h = (float)(sums[P_H0 + k] * (2.0 * f));
      ev.hessian[6 * a + b] = h;
      ev.hessian[6 * b + a] = h;
      k++;
    }
  ev.num_pairs = (int64_t)sums[P_PAIRS];
}

struct GaussNewtonParams {
  float threshold[6];   // flat test on the gradient, as updater.go:45-54 (all-zero -> 0.01)
  float damping;        // Levenberg-Marquardt: H + damping * diag(H)
  int32_t max_iteration;  // 0 -> 20 (updater.go:33)
};

PCGX_HD GaussNewtonParams resolve_gauss_newton(const float *threshold, float damping, int32_t max_iteration) {
  GaussNewtonParams u;
  bool tz = true;
  for (int i = 0; i < 6; i++)
    if (threshold[i] != 0.0f) tz = false;
  for (int i = 0; i < 6; i++) u.threshold[i] = tz ? 0.01f : threshold[i];
  u.damping = damping;
  u.max_iteration = max_iteration == 0 ? 20 : max_iteration;
  return u;
}

// Solves (H + damping diag H) d = -g by Cholesky in float64.  false: not positive definite.
PCGX_HD bool gauss_newton_solve(const float *h36, const float *g6, float damping, float d6[6]) {
  double a[6][6], y[6];
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) a[i][j] = (double)h36[6 * i + j];
  double tr = 0.0;
  for (int i = 0; i < 6; i++) {
    a[i][i] = a[i][i] + (double)damping * a[i][i];
    tr += a[i][i];
  }
  if (!(tr > 0.0)) return false;
  const double tiny = tr * 1e-12;
  for (int j = 0; j < 6; j++) {  // a = L L^T, L stored in the lower triangle
    double s = a[j][j];
    for (int k = 0; k < j; k++) s -= a[j][k] * a[j][k];
    if (!(s > tiny)) return false;
    const double l = sqrt(s);
    a[j][j] = l;
    for (int i = j + 1; i < 6; i++) {
      double t = a[i][j];
      for (int k = 0; k < j; k++) t -= a[i][k] * a[j][k];
      a[i][j] = t / l;
    }
  }
  for (int i = 0; i < 6; i++) {  // L y = -g
    double t = -(double)g6[i];
    for (int k = 0; k < i; k++) t -= a[i][k] * y[k];
    y[i] = t / a[i][i];
  }
  for (int i = 5; i >= 0; i--) {  // L^T d = y
    double t = y[i];
    for (int k = i + 1; k < 6; k++) t -= a[k][i] * y[k];
    y[i] = t / a[i][i];
  }
  for (int i = 0; i < 6; i++) d6[i] = (float)y[i];
  return true;
}

// Gauss-Newton counterpart of gradientDescentUpdater.Update: same flat test, same pose
// composition trans = Translate(d0..2) * (Rodrigues(d3..5) * trans), same iteration cap.
// Returns 1 converged, 0 continue, -1 singular normal equations (trans unchanged).
PCGX_HD int gauss_newton_update(const GaussNewtonParams &u, int32_t &iter, const EvaluatedPlane &ev, Mat4 &trans) {
  bool flat = true;
  for (int j = 0; j < 6; j++) {
    if (ev.gradient[j] < -u.threshold[j] || u.threshold[j] < ev.gradient[j]) {
      flat = false;
      break;
    }
  }
  if (flat) return 1;
  float d[6];
  if (!gauss_newton_solve(ev.hessian, ev.gradient, u.damping, d)) return -1;
  Mat4 delta_trans = mat4_translate(d[0], d[1], d[2]);
  Mat4 delta_rot = rodrigues_to_rotation(d[3], d[4], d[5]);
  trans = mat4_mul(delta_trans, mat4_mul(delta_rot, trans));
  iter = iter + 1;
  return iter >= u.max_iteration ? 1 : 0;
}

}  // namespace pcgx
