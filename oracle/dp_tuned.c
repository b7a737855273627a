/*
 * oracle/dp_tuned.c -- a TUNED CPU figure for the profile x profile DP.  TEST / BENCH INFRASTRUCTURE, NOT PRODUCT CODE.
 * "Parity unpinned": same specification as dp_oracle.h (this repo's own; the reference has no DP); this file
 * only exists so that bench.py can quote an honest host-core rate beside the scalar port (cpu_baseline.tuned) -- the port
 * walks a full matrix with a 25-multiply column score per cell, which no one would run on a CPU on purpose.
 *
 * What is tuned (scores only, results identical to dp_oracle_score, checked by tests/test_dp_oracle.py):
 *   - B's columns folded with the matrix once per pair (5 weights per column), so a cell's score is a 5-term dot;
 *   - two rows of H/F instead of the matrix;
 *   - skewed coordinates V~ = V + (i+j) * gap_extend, as the GPU kernel carries them: extending a gap costs nothing,
 *     E = max(E, H_left - gop), F = max(F, H_up - gop), H = max(diag + s, E, F);
 *   - INTER-PAIR SIMD: DP_LANES consecutive pairs of identical shape run in lock step, one pair per vector lane (the
 *     batch layout bench.py's configurations have); every loop over `l` below is a plain int32 loop the compiler turns
 *     into AVX2 / AVX-512 code under -O3 -march=native.  Pairs that do not form such a group take the scalar loop.
 * Built by oracle/pyoracle.py on the machine it runs on (gcc -O3 -march=native), never shipped.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define DP_NEG_INF (-(1 << 29))
#ifndef DP_LANES
#define DP_LANES 16
#endif

typedef struct dp_params {
  int32_t sub[25];
  int32_t gap_open;
  int32_t gap_extend;
} dp_params_t;

static inline int32_t max32(int32_t a, int32_t b) { return a > b ? a : b; }

static int32_t score_one(const uint8_t *cols_a, int32_t la, const uint8_t *cols_b, int32_t lb, const dp_params_t *p, int32_t *h, int32_t *f,
                         int32_t *wb) {
  const int32_t go = p->gap_open, ge = p->gap_extend, gop = go - ge;
  if(la == 0 || lb == 0) {
    int32_t n = la + lb;
    return n == 0 ? 0 : -(go + (n - 1) * ge);
  }
  for(int32_t j = 0; j < lb; ++j) {
    for(int a = 0; a < 5; ++a) {
      int32_t s = 0;
      for(int b = 0; b < 5; ++b) {
        s += (int32_t)cols_b[(size_t)j * 8 + b] * p->sub[a * 5 + b];
      }
      wb[(size_t)j * 5 + a] = s;
    }
  }
  /* skewed: H~[0][j] = -(go + (j-1) ge) + j ge = -gop for j >= 1 */
  h[0] = 0;
  f[0] = DP_NEG_INF;
  for(int32_t j = 1; j <= lb; ++j) {
    h[j] = -gop;
    f[j] = DP_NEG_INF;
  }
  for(int32_t i = 1; i <= la; ++i) {
    const uint8_t *ca = cols_a + (size_t)(i - 1) * 8;
    const int32_t a0 = ca[0], a1 = ca[1], a2 = ca[2], a3 = ca[3], a4 = ca[4];
    int32_t diag = h[0];
    h[0] = -gop; /* H~[i][0] */
    int32_t e = DP_NEG_INF, left = h[0];
    for(int32_t j = 1; j <= lb; ++j) {
      const int32_t *wj = wb + (size_t)(j - 1) * 5;
      e = max32(e, left - gop);
      const int32_t fv = max32(f[j], h[j] - gop);
      const int32_t d = diag + 2 * ge + a0 * wj[0] + a1 * wj[1] + a2 * wj[2] + a3 * wj[3] + a4 * wj[4]; /* 2 ge: the diagonal step's skew */
      const int32_t best = max32(d, max32(e, fv));
      diag = h[j];
      h[j] = best;
      f[j] = fv;
      left = best;
    }
  }
  return h[lb] - (la + lb) * ge;
}

/* DP_LANES pairs of one shape in lock step; arrays are [column][lane] */
static void score_group(const uint8_t *cols_a, const int64_t *off_a, const uint8_t *cols_b, const int64_t *off_b, int32_t la, int32_t lb,
                        const dp_params_t *p, int32_t *scores, int32_t *h, int32_t *f, int32_t *wb, int32_t *av) {
  enum { V = DP_LANES };
  const int32_t go = p->gap_open, ge = p->gap_extend, gop = go - ge;
  for(int l = 0; l < V; ++l) {
    const uint8_t *cb = cols_b + off_b[l] * 8;
    for(int32_t j = 0; j < lb; ++j) {
      for(int a = 0; a < 5; ++a) {
        int32_t s = 0;
        for(int b = 0; b < 5; ++b) {
          s += (int32_t)cb[(size_t)j * 8 + b] * p->sub[a * 5 + b];
        }
        wb[((size_t)j * 5 + a) * V + l] = s;
      }
    }
    const uint8_t *ca = cols_a + off_a[l] * 8;
    for(int32_t i = 0; i < la; ++i) {
      for(int a = 0; a < 5; ++a) {
        av[((size_t)i * 5 + a) * V + l] = ca[(size_t)i * 8 + a];
      }
    }
  }
  for(int l = 0; l < V; ++l) {
    h[l] = 0;
    f[l] = DP_NEG_INF;
  }
  for(int32_t j = 1; j <= lb; ++j) {
    for(int l = 0; l < V; ++l) {
      h[(size_t)j * V + l] = -gop;
      f[(size_t)j * V + l] = DP_NEG_INF;
    }
  }
  const int32_t skew2 = 2 * ge;
  for(int32_t i = 1; i <= la; ++i) {
    const int32_t *ai = av + (size_t)(i - 1) * 5 * V;
    int32_t diag[V], e[V], left[V];
    for(int l = 0; l < V; ++l) {
      diag[l] = h[l];
      h[l] = -gop;
      e[l] = DP_NEG_INF;
      left[l] = -gop;
    }
    for(int32_t j = 1; j <= lb; ++j) {
      const int32_t *wj = wb + (size_t)(j - 1) * 5 * V;
      int32_t *hj = h + (size_t)j * V, *fj = f + (size_t)j * V;
      for(int l = 0; l < V; ++l) {
        const int32_t ev = max32(e[l], left[l] - gop);
        const int32_t fv = max32(fj[l], hj[l] - gop);
        const int32_t d = diag[l] + skew2 + ai[l] * wj[l] + ai[V + l] * wj[V + l] + ai[2 * V + l] * wj[2 * V + l] +
                          ai[3 * V + l] * wj[3 * V + l] + ai[4 * V + l] * wj[4 * V + l];
        const int32_t best = max32(d, max32(ev, fv));
        diag[l] = hj[l];
        hj[l] = best;
        fj[l] = fv;
        e[l] = ev;
        left[l] = best;
      }
    }
  }
  for(int l = 0; l < V; ++l) {
    scores[l] = h[(size_t)lb * V + l] - (la + lb) * ge;
  }
}

/* Scores of a batch in the product's layout.  Returns 0, or -1 when out of memory. */
int dp_tuned_score_batch(const uint8_t *cols_a, const int64_t *off_a, const uint8_t *cols_b, const int64_t *off_b, int64_t n_pairs,
                         const dp_params_t *p, int32_t *scores) {
  enum { V = DP_LANES };
  int64_t max_la = 0, max_lb = 0;
  for(int64_t k = 0; k < n_pairs; ++k) {
    if(off_a[k + 1] - off_a[k] > max_la) {
      max_la = off_a[k + 1] - off_a[k];
    }
    if(off_b[k + 1] - off_b[k] > max_lb) {
      max_lb = off_b[k + 1] - off_b[k];
    }
  }
  int32_t *h = (int32_t *)aligned_alloc(64, ((size_t)(max_lb + 1) * V * 4 + 63) / 64 * 64);
  int32_t *f = (int32_t *)aligned_alloc(64, ((size_t)(max_lb + 1) * V * 4 + 63) / 64 * 64);
  int32_t *wb = (int32_t *)aligned_alloc(64, ((size_t)(max_lb + 1) * 5 * V * 4 + 63) / 64 * 64);
  int32_t *av = (int32_t *)aligned_alloc(64, ((size_t)(max_la + 1) * 5 * V * 4 + 63) / 64 * 64);
  if(!h || !f || !wb || !av) {
    free(h);
    free(f);
    free(wb);
    free(av);
    return -1;
  }
  int64_t k = 0;
  while(k < n_pairs) {
    const int32_t la = (int32_t)(off_a[k + 1] - off_a[k]), lb = (int32_t)(off_b[k + 1] - off_b[k]);
    int same = k + V <= n_pairs && la > 0 && lb > 0;
    for(int l = 1; same && l < V; ++l) {
      same = off_a[k + l + 1] - off_a[k + l] == la && off_b[k + l + 1] - off_b[k + l] == lb;
    }
    if(same) {
      score_group(cols_a, off_a + k, cols_b, off_b + k, la, lb, p, scores + k, h, f, wb, av);
      k += V;
    }
    else {
      scores[k] = score_one(cols_a + off_a[k] * 8, la, cols_b + off_b[k] * 8, lb, p, h, f, wb);
      ++k;
    }
  }
  free(h);
  free(f);
  free(wb);
  free(av);
  return 0;
}
