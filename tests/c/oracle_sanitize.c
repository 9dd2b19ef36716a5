/* Sanitizer driver of the CPU oracle's C part (VERDICT r5 Next 9): built by tests/test_sanitizers.py with
 *   gcc -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -ffp-contract=off
 * together with oracle/frnn_bruteforce.c, run under `pytest -m "not gpu"`.  Every output buffer is allocated at EXACTLY the size the
 * callers in oracle/torch_ref.py allocate (AddressSanitizer red zones sit right behind them), inputs cover the shapes the tests use:
 * ragged clouds, an empty cloud, K larger than the cloud, K = 1, duplicate points (ties), P1 != P2, per-cloud radii.  Besides
 * memory / undefined-behaviour errors it checks the invariants every consumer relies on: -1 padding, ascending (d2, index) order,
 * strict d2 < r^2, ball query = first K in index order.  Exit code 0 = clean. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

void ccn_oracle_frnn(const float *, const float *, const int64_t *, const int64_t *, int64_t, int64_t, int64_t, int64_t,
                     const float *, int64_t *, float *);
void ccn_oracle_knn(const float *, const float *, const int64_t *, const int64_t *, int64_t, int64_t, int64_t, int64_t,
                    int64_t *, float *);
void ccn_oracle_ball_query(const float *, const float *, const int64_t *, const int64_t *, int64_t, int64_t, int64_t, int64_t,
                           float, int64_t *);

static uint32_t state = 12345u;
static float rnd(void) {
  state = state * 1664525u + 1013904223u;
  return (float)(state >> 8) / 16777216.0f;
}

static int fail(const char *what, int64_t b, int64_t i) {
  fprintf(stderr, "oracle_sanitize: %s (cloud %lld, query %lld)\n", what, (long long)b, (long long)i);
  return 1;
}

static int run_case(int64_t B, int64_t P1, int64_t P2, int64_t K, const int64_t *l1, const int64_t *l2, int dup) {
  float *p1 = malloc(sizeof(float) * (size_t)(B * P1 * 3 ? B * P1 * 3 : 1));
  float *p2 = malloc(sizeof(float) * (size_t)(B * P2 * 3 ? B * P2 * 3 : 1));
  float *r = malloc(sizeof(float) * (size_t)B);
  int64_t *idx = malloc(sizeof(int64_t) * (size_t)(B * P1 * K ? B * P1 * K : 1));
  float *d2 = malloc(sizeof(float) * (size_t)(B * P1 * K ? B * P1 * K : 1));
  for (int64_t i = 0; i < B * P1 * 3; ++i) p1[i] = rnd();
  for (int64_t i = 0; i < B * P2 * 3; ++i) p2[i] = dup ? floorf(rnd() * 4.0f) * 0.25f : rnd();   /* lattice: equal distances */
  for (int64_t b = 0; b < B; ++b) r[b] = 0.2f + 0.1f * (float)b;
  int bad = 0;
  /* --- fixed radius */
  for (int64_t i = 0; i < B * P1 * K; ++i) { idx[i] = -1; d2[i] = -1.0f; }     /* as oracle.torch_ref.frnn_bruteforce pre-fills */
  ccn_oracle_frnn(p1, p2, l1, l2, B, P1, P2, K, r, idx, d2);
  for (int64_t b = 0; b < B && !bad; ++b)
    for (int64_t i = 0; i < P1 && !bad; ++i) {
      const int64_t *oi = idx + (b * P1 + i) * K;
      const float *od = d2 + (b * P1 + i) * K;
      if (i >= l1[b]) {
        for (int64_t k = 0; k < K; ++k) if (oi[k] != -1) bad = fail("frnn: a row beyond lengths1 was written", b, i);
        continue;
      }
      int64_t want = 0;
      for (int64_t j = 0; j < l2[b]; ++j) {
        const float dx = p2[(b * P2 + j) * 3] - p1[(b * P1 + i) * 3], dy = p2[(b * P2 + j) * 3 + 1] - p1[(b * P1 + i) * 3 + 1],
                    dz = p2[(b * P2 + j) * 3 + 2] - p1[(b * P1 + i) * 3 + 2];
        if (fmaf(dz, dz, fmaf(dy, dy, dx * dx)) < r[b] * r[b]) ++want;
      }
      if (want > K) want = K;
      for (int64_t k = 0; k < K; ++k) {
        if (k < want) {
          if (oi[k] < 0 || oi[k] >= l2[b]) bad = fail("frnn: index out of the cloud", b, i);
          else if (!(od[k] < r[b] * r[b])) bad = fail("frnn: neighbour not strictly inside the radius", b, i);
          else if (k && (od[k - 1] > od[k] || (od[k - 1] == od[k] && oi[k - 1] >= oi[k])))
            bad = fail("frnn: not ascending by (d2, index)", b, i);
        } else if (oi[k] != -1) {
          bad = fail("frnn: missing -1 padding", b, i);
        }
      }
    }
  /* --- exact kNN */
  for (int64_t i = 0; i < B * P1 * K; ++i) { idx[i] = -1; d2[i] = -1.0f; }
  ccn_oracle_knn(p1, p2, l1, l2, B, P1, P2, K, idx, d2);
  for (int64_t b = 0; b < B && !bad; ++b)
    for (int64_t i = 0; i < l1[b] && !bad; ++i) {
      const int64_t *oi = idx + (b * P1 + i) * K;
      const float *od = d2 + (b * P1 + i) * K;
      const int64_t want = l2[b] < K ? l2[b] : K;
      for (int64_t k = 0; k < K; ++k) {
        if (k < want) {
          if (oi[k] < 0 || oi[k] >= l2[b]) bad = fail("knn: index out of the cloud", b, i);
          else if (k && (od[k - 1] > od[k] || (od[k - 1] == od[k] && oi[k - 1] >= oi[k]))) bad = fail("knn: order", b, i);
        } else if (oi[k] != -1) {
          bad = fail("knn: missing -1 padding", b, i);
        }
      }
    }
  /* --- ball query (one radius for the batch) */
  for (int64_t i = 0; i < B * P1 * K; ++i) idx[i] = -1;
  ccn_oracle_ball_query(p1, p2, l1, l2, B, P1, P2, K, 0.3f, idx);
  for (int64_t b = 0; b < B && !bad; ++b)
    for (int64_t i = 0; i < l1[b] && !bad; ++i) {
      const int64_t *oi = idx + (b * P1 + i) * K;
      for (int64_t k = 0; k < K; ++k) {
        if (oi[k] >= l2[b] || oi[k] < -1) bad = fail("ball query: index out of the cloud", b, i);
        if (k && oi[k] != -1 && oi[k - 1] >= oi[k]) bad = fail("ball query: not in index order", b, i);
        if (k && oi[k] != -1 && oi[k - 1] == -1) bad = fail("ball query: a hole in the list", b, i);
      }
    }
  free(p1); free(p2); free(r); free(idx); free(d2);
  return bad;
}

int main(void) {
  int bad = 0;
  { const int64_t l1[] = {37, 0, 64}, l2[] = {50, 12, 0};   bad |= run_case(3, 64, 50, 8, l1, l2, 0); }   /* ragged, empty clouds */
  { const int64_t l1[] = {5}, l2[] = {3};                   bad |= run_case(1, 5, 3, 32, l1, l2, 0); }    /* K > cloud */
  { const int64_t l1[] = {200, 131}, l2[] = {200, 177};     bad |= run_case(2, 200, 200, 1, l1, l2, 0); } /* K = 1 */
  { const int64_t l1[] = {90, 90}, l2[] = {300, 299};       bad |= run_case(2, 90, 300, 20, l1, l2, 1); } /* lattice ties */
  { const int64_t l1[] = {0}, l2[] = {0};                   bad |= run_case(1, 1, 1, 4, l1, l2, 0); }     /* nothing at all */
  if (!bad) printf("oracle_sanitize: clean\n");
  return bad;
}
