/*
 * segment_oracle.c -- CPU ORACLE (test infrastructure, NOT product code): plain-C restatement
 * of the reference's bucket voxel grid, voxel flood-fill segmentation and region growing
 * (SURVEY.md 8(f) N2 / N3).  Linked into liboracle.so with pcgol_oracle.c; same rules as there:
 * only tests/, smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Pinned by the reference's own known-answer tests, transcribed as data in
 * tests/golden/ref_segment.json and checked by tests/test_oracle_golden.py:
 *   pc/storage/voxelgrid/voxelgrid_test.go:10-87
 *   pc/segmentation/voxelgrid/voxelgrid_test.go:11-40
 *   pc/segmentation/regiongrowing/regiongrowing_test.go:15-175 (scene + expected sets; the
 *     reference's +-0.01 noise is unseeded -- the fixture uses a seeded draw of the same kind)
 *
 * float32, left to right, no FMA (gcc -ffp-contract=off), Go truncation int(x).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct okdtree okdtree;
int64_t orc_kdtree_range(okdtree *t, const float *p, float max_range, int64_t *ids, float *dsq, int64_t cap);
const float *orc_kdtree_point(const okdtree *t, int64_t id);
int64_t orc_kdtree_len(const okdtree *t);

/* ---- pc/storage/voxelgrid/voxelgrid.go:7-13 VoxelGrid ([][]int buckets) */
typedef struct {
  int64_t *head, *tail, *count; /* per voxel: first / last entry, length */
  int64_t *next, *index;        /* entries in insertion order */
  int64_t n_entries, cap;
  int64_t size[3];
  float origin[3];
  float resolution, resolution_inv;
} ogrid;

/* voxelgrid.go:15-23 New */
ogrid *orc_grid_new(float resolution, const int64_t *size, const float *origin) {
  ogrid *g = calloc(1, sizeof *g);
  if (!g) return NULL;
  const int64_t len = size[0] * size[1] * size[2];
  g->head = malloc((size_t)(len ? len : 1) * sizeof(int64_t));
  g->tail = malloc((size_t)(len ? len : 1) * sizeof(int64_t));
  g->count = calloc((size_t)(len ? len : 1), sizeof(int64_t));
  for (int64_t i = 0; i < len; i++) g->head[i] = g->tail[i] = -1;
  memcpy(g->size, size, sizeof g->size);
  memcpy(g->origin, origin, sizeof g->origin);
  g->resolution = resolution;
  g->resolution_inv = 1 / resolution;
  return g;
}
void orc_grid_free(ogrid *g) {
  if (!g) return;
  free(g->head); free(g->tail); free(g->count); free(g->next); free(g->index);
  free(g);
}
int64_t orc_grid_len(const ogrid *g) { return g->size[0] * g->size[1] * g->size[2]; } /* :110-112 */

/* voxelgrid.go:94-108 PosInt: int(pos*resolutionInv + 0.5), truncation toward zero */
int orc_grid_pos_int(const ogrid *g, const float *p, int64_t *xyz) {
  for (int k = 0; k < 3; k++) {
    float pos = p[k] - g->origin[k];
    int64_t v = (int64_t)(pos * g->resolution_inv + 0.5f);
    if (v < 0 || v >= g->size[k]) return 0;
    xyz[k] = v;
  }
  return 1;
}
/* voxelgrid.go:64-79 Addr */
int orc_grid_addr(const ogrid *g, const float *p, int64_t *addr) {
  int64_t v[3];
  if (!orc_grid_pos_int(g, p, v)) return 0;
  *addr = v[0] + (v[1] + v[2] * g->size[1]) * g->size[0];
  return 1;
}
/* voxelgrid.go:81-92 AddrByPosInt */
static int addr_by_pos_int(const ogrid *g, const int64_t *v, int64_t *addr) {
  if (v[0] < 0 || v[1] < 0 || v[2] < 0 || v[0] >= g->size[0] || v[1] >= g->size[1] || v[2] >= g->size[2]) return 0;
  *addr = v[0] + (v[1] + v[2] * g->size[1]) * g->size[0];
  return 1;
}
/* voxelgrid.go:47-50 AddByAddr */
void orc_grid_add_by_addr(ogrid *g, int64_t a, int64_t index) {
  if (g->n_entries == g->cap) {
    g->cap = g->cap ? 2 * g->cap : 1024;
    g->next = realloc(g->next, (size_t)g->cap * sizeof(int64_t));
    g->index = realloc(g->index, (size_t)g->cap * sizeof(int64_t));
  }
  const int64_t e = g->n_entries++;
  g->index[e] = index;
  g->next[e] = -1;
  if (g->tail[a] >= 0) g->next[g->tail[a]] = e; else g->head[a] = e;
  g->tail[a] = e;
  g->count[a]++;
}
/* voxelgrid.go:37-45 Add */
int orc_grid_add(ogrid *g, const float *p, int64_t index) {
  int64_t a;
  if (!orc_grid_addr(g, p, &a)) return 0;
  orc_grid_add_by_addr(g, a, index);
  return 1;
}
/* voxelgrid.go:60-62 GetByAddr: copies the bucket, returns its length */
int64_t orc_grid_get_by_addr(const ogrid *g, int64_t a, int64_t *out, int64_t cap) {
  int64_t k = 0;
  for (int64_t e = g->head[a]; e >= 0; e = g->next[e]) {
    if (k < cap) out[k] = g->index[e];
    k++;
  }
  return k;
}
/* voxelgrid.go:52-58 Get: -1 = nil (outside the grid) */
int64_t orc_grid_get(const ogrid *g, const float *p, int64_t *out, int64_t cap) {
  int64_t a;
  if (!orc_grid_addr(g, p, &a)) return -1;
  return orc_grid_get_by_addr(g, a, out, cap);
}
/* voxelgrid.go:114-120 Indice */
int64_t orc_grid_indice(const ogrid *g, int64_t *out) {
  int64_t k = 0;
  const int64_t len = orc_grid_len(g);
  for (int64_t a = 0; a < len; a++)
    for (int64_t e = g->head[a]; e >= 0; e = g->next[e]) out[k++] = g->index[e];
  return k;
}

/* ---- pc/segmentation/voxelgrid/voxelgrid.go:39-73 Segment: 26-neighbour flood fill, FIFO,
 * cursor order x, y, z in {-1, 0, 1} (:13-25).  out must hold n_entries values. */
int64_t orc_grid_segment(const ogrid *g, const float *p, int64_t *out) {
  int64_t pos[3];
  if (!orc_grid_pos_int(g, p, pos)) return 0;
  const int64_t len = orc_grid_len(g);
  uint8_t *searched = calloc((size_t)(len ? len : 1), 1);
  int64_t qcap = 1024, qh = 0, qt = 0;
  int64_t(*queue)[3] = malloc((size_t)qcap * sizeof *queue);
  memcpy(queue[qt++], pos, sizeof pos);
  int64_t k = 0;
  while (qh < qt) {
    int64_t cur[3];
    memcpy(cur, queue[qh++], sizeof cur);
    int64_t addr;
    if (!addr_by_pos_int(g, cur, &addr) || searched[addr]) continue;
    searched[addr] = 1;
    if (g->count[addr] == 0) continue;
    for (int64_t e = g->head[addr]; e >= 0; e = g->next[e]) out[k++] = g->index[e];
    for (int dx = -1; dx <= 1; dx++)
      for (int dy = -1; dy <= 1; dy++)
        for (int dz = -1; dz <= 1; dz++) {
          if (dx == 0 && dy == 0 && dz == 0) continue;
          int64_t n[3] = {cur[0] + dx, cur[1] + dy, cur[2] + dz}, a2;
          if (!addr_by_pos_int(g, n, &a2) || searched[a2]) continue;
          if (qt == qcap) {
            qcap *= 2;
            queue = realloc(queue, (size_t)qcap * sizeof *queue);
          }
          memcpy(queue[qt++], n, sizeof n);
        }
  }
  free(queue);
  free(searched);
  return k;
}

/* ---- pc/segmentation/regiongrowing/regiongrowing.go:23-56 Segment.  labels: the property
 * accessor (Uint32At).  out must hold Len() values; returns the number written (BFS order). */
int64_t orc_region_growing_segment(okdtree *t, const uint32_t *labels, const float *p, float max_range,
                                   int64_t *out) {
  const int64_t n = orc_kdtree_len(t);
  int64_t *nb = malloc((size_t)n * sizeof(int64_t));
  float *nd = malloc((size_t)n * sizeof(float));
  int64_t *next = malloc((size_t)n * sizeof(int64_t));
  uint8_t *to_visit = calloc((size_t)n, 1);
  int64_t k = 0, qh = 0, qt = 0;
  int64_t m = orc_kdtree_range(t, p, max_range, nb, nd, n);
  if (m > 0) {
    const uint32_t target = labels[nb[0]]; /* :31 */
    for (int64_t i = 0; i < m; i++) { next[qt++] = nb[i]; to_visit[nb[i]] = 1; }
    while (qh < qt) {
      const int64_t id = next[qh++];
      if (labels[id] != target) continue;
      out[k++] = id;
      m = orc_kdtree_range(t, orc_kdtree_point(t, id), max_range, nb, nd, n);
      for (int64_t i = 0; i < m; i++)
        if (!to_visit[nb[i]]) { next[qt++] = nb[i]; to_visit[nb[i]] = 1; }
    }
  }
  free(nb); free(nd); free(next); free(to_visit);
  return k;
}
