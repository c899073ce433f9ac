/*
 * plane_oracle.c -- CPU ORACLE (test infrastructure, NOT product code) for the
 * point-to-plane / Gauss-Newton ICP EXTENSION.
 *
 * PARITY UNPINNED: the reference (seqsense/pcgol) has no point-to-plane evaluator and
 * never writes Evaluated.Hessian (pc/registration/icp/evaluator.go:28,76; mat/mat6.go:3
 * is a bare type).  There is nothing in the reference to pin this file against; it is an
 * independent plain-C statement of the extension's definition (include/pcgx.h, "point-to-plane
 * ICP (extension)"), written separately from the product's pcgx_math.h, against which the HIP
 * path is checked.  The correspondence step IS the reference's (correspondence.go:22-37 via
 * orc_kdtree_nearest, pinned in pcgol_oracle.c).
 *
 * Definition (conventions of the reference's point-to-point evaluator, evaluator.go:122-145,
 * and updater, updater.go:44-71):
 *   pair (pt target, pb base, n unit normal of pb):   r = n . (pt - pb),  J = {n, pt x n}
 *   sums30 = {sum r^2, sum J r [6], upper triangle of sum J J^T row-major [21], sum w, pairs}
 *   every product is formed in float32 (left to right, no FMA), accumulated in float64 in
 *   target order;
 *   f = 1/sum(w) if sum(w) > 1;  Value = f sum r^2;  Gradient = 2 f sum J r;  Hessian = 2 f sum J J^T
 *   update: flat test on the gradient (updater.go:45-54); solve (H + damping diag H) d = -g
 *   (float64 Cholesky); trans = Translate(d0..2) * (Rodrigues(d3..5) * trans); i++.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_OK 0
#define ORC_E_NOT_ENOUGH_PAIRS 2
#define ORC_E_OOM 4
#define ORC_E_SINGULAR 5

typedef struct okdtree okdtree;
void orc_kdtree_nearest(okdtree *t, const float *p, float max_range, float min_dist_sq, int64_t *id, float *dsq);
const float *orc_kdtree_point(const okdtree *t, int64_t id);
void orc_translate(float x, float y, float z, float *out);
void orc_rodrigues(const float *v, float *out);
void orc_mat4_mul(const float *m, const float *a, float *out);
void orc_mat4_transform(const float *m, const float *a, float *out);

/* The 30 sums of one evaluation at the given (already transformed) target. */
int orc_plane_sums(okdtree *t, const float *normals, const float *target, int64_t nt, float max_dist,
                   double *sums30) {
  for (int k = 0; k < 30; k++) sums30[k] = 0.0;
  for (int64_t i = 0; i < nt; i++) {
    int64_t id;
    float dsq;
    orc_kdtree_nearest(t, target + 3 * i, max_dist, 0.0f, &id, &dsq);
    if (id < 0) continue; /* correspondence.go:27-29 */
    const float *pb = orc_kdtree_point(t, id);
    const float *n = normals + 3 * id;
    const float x0 = target[3 * i], y0 = target[3 * i + 1], z0 = target[3 * i + 2];
    const float dx = x0 - pb[0], dy = y0 - pb[1], dz = z0 - pb[2];
    float r = n[0] * dx;
    r = r + n[1] * dy;
    r = r + n[2] * dz;
    float J[6];
    J[0] = n[0];
    J[1] = n[1];
    J[2] = n[2];
    J[3] = y0 * n[2] - z0 * n[1];
    J[4] = z0 * n[0] - x0 * n[2];
    J[5] = x0 * n[1] - y0 * n[0];
    sums30[0] += (double)(r * r);
    for (int a = 0; a < 6; a++) sums30[1 + a] += (double)(J[a] * r);
    int k = 7;
    for (int a = 0; a < 6; a++)
      for (int b = a; b < 6; b++) sums30[k++] += (double)(J[a] * J[b]);
    sums30[28] += 1.0;
    sums30[29] += 1.0;
  }
  return ORC_OK;
}

int orc_plane_finish(const double *sums30, int32_t min_pairs, float *value, float *grad6, float *hess36,
                     int64_t *npairs) {
  if (min_pairs == 0) min_pairs = 6;
  *npairs = (int64_t)sums30[29];
  if (*npairs < min_pairs) return ORC_E_NOT_ENOUGH_PAIRS;
  double f = 1.0;
  if (sums30[28] > 1.0) f = 1.0 / sums30[28];
  *value = (float)(sums30[0] * f);
  for (int a = 0; a < 6; a++) grad6[a] = (float)(sums30[1 + a] * (2.0 * f));
  int k = 7;
  for (int a = 0; a < 6; a++)
    for (int b = a; b < 6; b++) {
      float h = (float)(sums30[k++] * (2.0 * f));
      hess36[6 * a + b] = h;
      hess36[6 * b + a] = h;
    }
  return ORC_OK;
}

/* Returns 1 converged, 0 continue, -1 singular. */
int orc_gauss_newton_update(const float *thresh_in, float damping, int32_t max_iter, int32_t *iter,
                            const float *grad6, const float *hess36, float *trans) {
  float thresh[6];
  int tz = 1;
  for (int k = 0; k < 6; k++)
    if (thresh_in[k] != 0) tz = 0;
  for (int k = 0; k < 6; k++) thresh[k] = tz ? 0.01f : thresh_in[k];
  if (max_iter == 0) max_iter = 20;
  int flat = 1;
  for (int j = 0; j < 6; j++)
    if (grad6[j] < -thresh[j] || thresh[j] < grad6[j]) { flat = 0; break; }
  if (flat) return 1;
  /* Gaussian elimination with the Cholesky recurrences written out on a dense copy */
  double L[6][6], y[6], d[6], tr = 0.0;
  memset(L, 0, sizeof L);
  double A[6][6];
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) A[i][j] = (double)hess36[6 * i + j];
  for (int i = 0; i < 6; i++) { A[i][i] += (double)damping * A[i][i]; tr += A[i][i]; }
  if (!(tr > 0.0)) return -1;
  for (int j = 0; j < 6; j++) {
    double s = A[j][j];
    for (int k = 0; k < j; k++) s -= L[j][k] * L[j][k];
    if (!(s > tr * 1e-12)) return -1;
    L[j][j] = sqrt(s);
    for (int i = j + 1; i < 6; i++) {
      double v = A[i][j];
      for (int k = 0; k < j; k++) v -= L[i][k] * L[j][k];
      L[i][j] = v / L[j][j];
    }
  }
  for (int i = 0; i < 6; i++) {
    double v = -(double)grad6[i];
    for (int k = 0; k < i; k++) v -= L[i][k] * y[k];
    y[i] = v / L[i][i];
  }
  for (int i = 5; i >= 0; i--) {
    double v = y[i];
    for (int k = i + 1; k < 6; k++) v -= L[k][i] * d[k];
    d[i] = v / L[i][i];
  }
  float delta[6];
  for (int k = 0; k < 6; k++) delta[k] = (float)d[k];
  float dt[16], drot[16], tmp[16], res[16];
  orc_translate(delta[0], delta[1], delta[2], dt);
  orc_rodrigues(delta + 3, drot);
  orc_mat4_mul(drot, trans, tmp);
  orc_mat4_mul(dt, tmp, res);
  memcpy(trans, res, sizeof res);
  (*iter)++;
  return *iter >= max_iter ? 1 : 0;
}

/* The Fit loop of icp.go:23-67 with the plane evaluator / Gauss-Newton updater. */
int orc_plane_fit(okdtree *t, const float *normals, const float *target, int64_t nt, float max_dist,
                  int32_t min_pairs, const float *thresh6, float damping, int32_t max_iter, float *trans16,
                  float *out_value, float *out_grad6, float *out_hess36, int32_t *num_iteration) {
  float *tt = malloc((size_t)(nt ? nt : 1) * 12);
  if (!tt) return ORC_E_OOM;
  memcpy(tt, target, (size_t)nt * 12);
  float trans[16];
  orc_translate(0, 0, 0, trans);
  int32_t it = 0, niter = 0;
  int rc = ORC_OK;
  for (;;) {
    double sums[30];
    int64_t np;
    orc_plane_sums(t, normals, tt, nt, max_dist, sums);
    niter++;
    rc = orc_plane_finish(sums, min_pairs, out_value, out_grad6, out_hess36, &np);
    if (rc) break;
    int c = orc_gauss_newton_update(thresh6, damping, max_iter, &it, out_grad6, out_hess36, trans);
    if (c < 0) { rc = ORC_E_SINGULAR; break; }
    if (c > 0) break;
    for (int64_t i = 0; i < nt; i++) orc_mat4_transform(trans, target + 3 * i, tt + 3 * i);
  }
  memcpy(trans16, trans, sizeof trans);
  *num_iteration = niter;
  free(tt);
  return rc;
}
