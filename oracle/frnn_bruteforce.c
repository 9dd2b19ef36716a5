/* CPU oracle (TEST INFRASTRUCTURE ONLY): exhaustive fixed-radius kNN and exact kNN.
 *
 * Restates the published semantics of the third-party FRNN package that the reference calls at
 * src/models/utils/point_ops.py:459 (frnn.frnn_grid_points; github.com/lxxue/FRNN, un-vendored
 * submodule, pin unrecoverable -> "parity unpinned", see oracle/torch_ref.py) and of
 * pytorch3d.ops.knn_points (point_ops.py:91).  The grid in FRNN only prunes candidates, so an
 * exhaustive search defines the same result set.
 *
 * Arithmetic (fixed here AND in the HIP kernels):
 *     d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx)),   d = p2 - p1 component-wise,
 *     neighbour iff d2 < r * r (float),  order ascending by (d2, index).
 * Build with -ffp-contract=off so that nothing else is fused.
 */
#include <math.h>
#include <stdint.h>

static void search(const float *p1, const float *p2, int64_t n1, int64_t n2, int64_t K, int use_r, float r2,
                   int64_t *idx, float *dist) {
#pragma omp parallel for schedule(static, 64)
  for (int64_t i = 0; i < n1; ++i) {
    int64_t *oi = idx + i * K;
    float *od = dist + i * K;
    int64_t have = 0;
    const float qx = p1[3 * i], qy = p1[3 * i + 1], qz = p1[3 * i + 2];
    for (int64_t j = 0; j < n2; ++j) {
      const float dx = p2[3 * j] - qx, dy = p2[3 * j + 1] - qy, dz = p2[3 * j + 2] - qz;
      const float d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
      if (use_r && !(d2 < r2)) continue;
      if (have == K && !(d2 < od[K - 1])) continue; /* equal distance: the smaller index stays */
      int64_t s = have < K ? have : K - 1;
      while (s > 0 && od[s - 1] > d2) {
        od[s] = od[s - 1];
        oi[s] = oi[s - 1];
        --s;
      }
      od[s] = d2;
      oi[s] = j;
      if (have < K) ++have;
    }
  }
}

void ccn_oracle_frnn(const float *points1, const float *points2, const int64_t *lengths1, const int64_t *lengths2,
                     int64_t B, int64_t P1, int64_t P2, int64_t K, const float *r, int64_t *idx, float *dist) {
  for (int64_t b = 0; b < B; ++b)
    search(points1 + b * P1 * 3, points2 + b * P2 * 3, lengths1[b], lengths2[b], K, 1, r[b] * r[b],
           idx + b * P1 * K, dist + b * P1 * K);
}

void ccn_oracle_knn(const float *points1, const float *points2, const int64_t *lengths1, const int64_t *lengths2,
                    int64_t B, int64_t P1, int64_t P2, int64_t K, int64_t *idx, float *dist) {
  for (int64_t b = 0; b < B; ++b)
    search(points1 + b * P1 * 3, points2 + b * P2 * 3, lengths1[b], lengths2[b], K, 0, 0.0f,
           idx + b * P1 * K, dist + b * P1 * K);
}

/* pytorch3d.ops.ball_query semantics (point_ops.py:81): the first K points2 (in index order) with d2 < r*r; -1 padded. */
void ccn_oracle_ball_query(const float *points1, const float *points2, const int64_t *lengths1,
                           const int64_t *lengths2, int64_t B, int64_t P1, int64_t P2, int64_t K, float r,
                           int64_t *idx) {
  const float r2 = r * r;
  for (int64_t b = 0; b < B; ++b) {
    const float *p1 = points1 + b * P1 * 3, *p2 = points2 + b * P2 * 3;
#pragma omp parallel for schedule(static, 64)
    for (int64_t i = 0; i < lengths1[b]; ++i) {
      int64_t have = 0;
      for (int64_t j = 0; j < lengths2[b] && have < K; ++j) {
        const float dx = p2[3 * j] - p1[3 * i], dy = p2[3 * j + 1] - p1[3 * i + 1], dz = p2[3 * j + 2] - p1[3 * i + 2];
        if (fmaf(dz, dz, fmaf(dy, dy, dx * dx)) < r2) idx[(b * P1 + i) * K + have++] = j;
      }
    }
  }
}
