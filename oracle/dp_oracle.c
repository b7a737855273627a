/*
 * oracle/dp_oracle.c -- CPU ORACLE of the profile x profile DP.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * "Parity unpinned": there is no reference implementation of this computation (see dp_oracle.h).
 * Plain scalar C, one cell at a time, exactly as the specification in dp_oracle.h reads.
 */
#include "dp_oracle.h"

#include <stdlib.h>
#include <string.h>

static inline int32_t max32(int32_t a, int32_t b) { return a > b ? a : b; }

static int32_t column_score(const uint8_t *ca, const uint8_t *cb, const dp_params_t *p) {
  int32_t s = 0;
  for(int a = 0; a < 5; ++a) {
    for(int b = 0; b < 5; ++b) {
      s += (int32_t)ca[a] * (int32_t)cb[b] * p->sub[a * 5 + b];
    }
  }
  return s;
}

/* trace byte per cell: bits 0-1 = source of H (0 diag, 1 E, 2 F), bit 2 = E extended, bit 3 = F extended */
int dp_oracle_align(const uint8_t *cols_a, int32_t la, const uint8_t *cols_b, int32_t lb, const dp_params_t *p,
                    int32_t *score, uint8_t *ops, int32_t *n_ops) {
  size_t w = (size_t)lb + 1;
  uint8_t *tb = (uint8_t *)malloc(((size_t)la + 1) * w);
  int32_t *h = (int32_t *)malloc(w * sizeof(int32_t));
  int32_t *f = (int32_t *)malloc(w * sizeof(int32_t));
  if(!tb || !h || !f) {
    free(tb);
    free(h);
    free(f);
    return -1;
  }
  const int32_t go = p->gap_open, ge = p->gap_extend;
  h[0] = 0;
  f[0] = DP_NEG_INF;
  for(int32_t j = 1; j <= lb; ++j) {
    h[j] = -(go + (j - 1) * ge);
    f[j] = DP_NEG_INF;
  }
  for(int32_t i = 1; i <= la; ++i) {
    int32_t diag = h[0];
    h[0] = -(go + (i - 1) * ge);
    int32_t e = DP_NEG_INF;
    int32_t left = h[0];
    uint8_t *row = tb + (size_t)i * w;
    for(int32_t j = 1; j <= lb; ++j) {
      uint8_t t = 0;
      int32_t e_ext = e - ge, e_open = left - go;
      if(e_open >= e_ext) {
        e = e_open;
      }
      else {
        e = e_ext;
        t |= 4;
      }
      int32_t f_ext = f[j] - ge, f_open = h[j] - go;
      int32_t fv;
      if(f_open >= f_ext) {
        fv = f_open;
      }
      else {
        fv = f_ext;
        t |= 8;
      }
      int32_t d = diag + column_score(cols_a + (size_t)(i - 1) * 8, cols_b + (size_t)(j - 1) * 8, p);
      int32_t best;
      if(d >= e && d >= fv) {
        best = d;
      }
      else if(e >= fv) {
        best = e;
        t |= 1;
      }
      else {
        best = fv;
        t |= 2;
      }
      diag = h[j];
      h[j] = best;
      f[j] = fv;
      left = best;
      row[j] = t;
    }
  }
  *score = h[lb];
  /* walk back */
  int32_t i = la, j = lb, n = 0;
  int state = 0; /* 0 = H, 1 = E, 2 = F */
  while(i > 0 || j > 0) {
    if(i == 0) {
      ops[n++] = DP_OP_I;
      --j;
      continue;
    }
    if(j == 0) {
      ops[n++] = DP_OP_D;
      --i;
      continue;
    }
    uint8_t t = tb[(size_t)i * w + j];
    if(state == 0) {
      int src = t & 3;
      if(src == 0) {
        ops[n++] = DP_OP_M;
        --i;
        --j;
      }
      else {
        state = src;
      }
    }
    else if(state == 1) {
      ops[n++] = DP_OP_I;
      if(!(t & 4)) {
        state = 0;
      }
      --j;
    }
    else {
      ops[n++] = DP_OP_D;
      if(!(t & 8)) {
        state = 0;
      }
      --i;
    }
  }
  for(int32_t a = 0, b = n - 1; a < b; ++a, --b) {
    uint8_t tmp = ops[a];
    ops[a] = ops[b];
    ops[b] = tmp;
  }
  *n_ops = n;
  free(tb);
  free(h);
  free(f);
  return 0;
}

int32_t dp_oracle_score(const uint8_t *cols_a, int32_t la, const uint8_t *cols_b, int32_t lb, const dp_params_t *p) {
  size_t w = (size_t)lb + 1;
  int32_t *h = (int32_t *)malloc(w * sizeof(int32_t));
  int32_t *f = (int32_t *)malloc(w * sizeof(int32_t));
  const int32_t go = p->gap_open, ge = p->gap_extend;
  /* B's columns folded with the matrix once: wb[j][a] = sum_b colB[j][b] * sub[a][b] */
  int32_t *wb = (int32_t *)malloc((size_t)lb * 5 * sizeof(int32_t));
  for(int32_t j = 0; j < lb; ++j) {
    for(int a = 0; a < 5; ++a) {
      int32_t s = 0;
      for(int b = 0; b < 5; ++b) {
        s += (int32_t)cols_b[(size_t)j * 8 + b] * p->sub[a * 5 + b];
      }
      wb[(size_t)j * 5 + a] = s;
    }
  }
  h[0] = 0;
  f[0] = DP_NEG_INF;
  for(int32_t j = 1; j <= lb; ++j) {
    h[j] = -(go + (j - 1) * ge);
    f[j] = DP_NEG_INF;
  }
  for(int32_t i = 1; i <= la; ++i) {
    const uint8_t *ca = cols_a + (size_t)(i - 1) * 8;
    int32_t diag = h[0];
    h[0] = -(go + (i - 1) * ge);
    int32_t e = DP_NEG_INF, left = h[0];
    for(int32_t j = 1; j <= lb; ++j) {
      const int32_t *wj = wb + (size_t)(j - 1) * 5;
      e = max32(e - ge, left - go);
      int32_t fv = max32(f[j] - ge, h[j] - go);
      int32_t d = diag + ca[0] * wj[0] + ca[1] * wj[1] + ca[2] * wj[2] + ca[3] * wj[3] + ca[4] * wj[4];
      int32_t best = max32(d, max32(e, fv));
      diag = h[j];
      h[j] = best;
      f[j] = fv;
      left = best;
    }
  }
  int32_t s = h[lb];
  free(h);
  free(f);
  free(wb);
  return s;
}

int dp_oracle_score_of_path(const uint8_t *cols_a, int32_t la, const uint8_t *cols_b, int32_t lb, const dp_params_t *p,
                            const uint8_t *ops, int32_t n_ops, int32_t *score) {
  int32_t i = 0, j = 0, s = 0;
  int prev = DP_OP_M;
  for(int32_t k = 0; k < n_ops; ++k) {
    int op = ops[k];
    if(op == DP_OP_M) {
      if(i >= la || j >= lb) {
        return -1;
      }
      s += column_score(cols_a + (size_t)i * 8, cols_b + (size_t)j * 8, p);
      ++i;
      ++j;
    }
    else if(op == DP_OP_I) {
      if(j >= lb) {
        return -1;
      }
      s -= (prev == DP_OP_I) ? p->gap_extend : p->gap_open;
      ++j;
    }
    else if(op == DP_OP_D) {
      if(i >= la) {
        return -1;
      }
      s -= (prev == DP_OP_D) ? p->gap_extend : p->gap_open;
      ++i;
    }
    else {
      return -1;
    }
    prev = op;
  }
  if(i != la || j != lb) {
    return -1;
  }
  *score = s;
  return 0;
}

/* Every path of a batch in pm_dp_batch_fetch's layout (pair k's path: the LAST n_ops[k] bytes of its slot at off_a[k] + off_b[k])
 * re-scored under the specification: rc[k] != 0 when the path does not span its pair, else scores[k] = what it scores.  The
 * full-size GPU tests run this over slices of a 100 000-pair batch on every host core. */
void dp_oracle_score_of_paths(const uint8_t *cols_a, const int64_t *off_a, const uint8_t *cols_b, const int64_t *off_b, int64_t first, int64_t end,
                              const dp_params_t *p, const uint8_t *ops, const int32_t *n_ops, int32_t *rc, int32_t *scores) {
  for(int64_t k = first; k < end; ++k) {
    const int32_t la = (int32_t)(off_a[k + 1] - off_a[k]), lb = (int32_t)(off_b[k + 1] - off_b[k]);
    const int64_t slot_end = off_a[k + 1] + off_b[k + 1];
    scores[k] = 0;
    rc[k] = n_ops[k] < 0 || n_ops[k] > la + lb
                ? -1
                : dp_oracle_score_of_path(cols_a + off_a[k] * 8, la, cols_b + off_b[k] * 8, lb, p, ops + slot_end - n_ops[k], n_ops[k], &scores[k]);
  }
}

int dp_oracle_align_batch(const uint8_t *cols_a, const int64_t *off_a, const uint8_t *cols_b, const int64_t *off_b, int64_t n_pairs,
                          const dp_params_t *p, int32_t *scores, uint8_t *ops, int32_t *n_ops) {
  for(int64_t k = 0; k < n_pairs; ++k) {
    int32_t la = (int32_t)(off_a[k + 1] - off_a[k]), lb = (int32_t)(off_b[k + 1] - off_b[k]);
    int rc = dp_oracle_align(cols_a + off_a[k] * 8, la, cols_b + off_b[k] * 8, lb, p, &scores[k], ops + off_a[k] + off_b[k], &n_ops[k]);
    if(rc) {
      return rc;
    }
  }
  return 0;
}

void dp_oracle_score_batch(const uint8_t *cols_a, const int64_t *off_a, const uint8_t *cols_b, const int64_t *off_b, int64_t n_pairs,
                           const dp_params_t *p, int32_t *scores) {
  for(int64_t k = 0; k < n_pairs; ++k) {
    scores[k] = dp_oracle_score(cols_a + off_a[k] * 8, (int32_t)(off_a[k + 1] - off_a[k]), cols_b + off_b[k] * 8,
                                (int32_t)(off_b[k + 1] - off_b[k]), p);
  }
}
