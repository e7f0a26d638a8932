/* TEST INFRASTRUCTURE ONLY (imported by tests/ through oracle/c_ref.py, never by ranking_amd/).
 *
 * Plain-C fp64 restatements of two listwise losses of the reference, forward and analytic backward, written as the
 * O(L^2) double loops the definitions are -- independent of the sort + scan formulation of the HIP kernels
 * (ranking_amd/csrc/listwise.hip), so that they can arbitrate between those and the torch restatement.
 *
 *   ListMLELoss        tensorflow_ranking/python/losses_impl.py:1541-1576 (ListMLELambdaWeight :457-480)
 *   UniqueSoftmaxLoss  tensorflow_ranking/python/losses_impl.py:1250-1281
 *
 * Tie rule between equal labels (the reference shuffles them at random, utils.py:100-112: unpinned): lower index first,
 * the same rule as oracle/tfr_ref.py and the kernels.  mask == NULL means label >= 0.
 */
#include <math.h>
#include <stdlib.h>

#define LOG_EPS -23.025850929940457 /* log(1e-10): the reference's masked logit (losses_impl.py:1552) */

typedef struct { double label; int valid; int index; } item_t;

static int by_valid_label_index(const void* a, const void* b) {
  const item_t* x = (const item_t*)a;
  const item_t* y = (const item_t*)b;
  if (x->valid != y->valid) return y->valid - x->valid;            /* valid first */
  if (x->label != y->label) return (x->label < y->label) ? 1 : -1; /* label descending */
  return x->index - y->index;
}

/* loss[b] = sum_p w_p (log sum_{q >= p} exp(x_q) - x_p) over ALL L positions of the label-sorted list, x = logit / T for
 * valid items and log(1e-10) for the others (they sort last); w_p = pos_weight[p] (rank discount of position p + 1) or
 * 1.  dlogits[b, k] = d loss[b] / d logits[b, k] (0 for invalid items). */
int tfr_c_list_mle_f64(const float* logits, const float* labels, const unsigned char* mask, const float* pos_weight,
                       int B, int L, float temperature, float* loss_out, float* dlogits_out) {
  if (!logits || !labels || !loss_out || B < 0 || L <= 0 || !(temperature > 0.0f)) return -1;
#pragma omp parallel for schedule(dynamic)
  for (int b = 0; b < B; ++b) {
    item_t* it = (item_t*)malloc(sizeof(item_t) * (size_t)L);
    double* x = (double*)malloc(sizeof(double) * (size_t)L);
    double* S = (double*)malloc(sizeof(double) * (size_t)L);
    const float* lg = logits + (size_t)b * L;
    const float* lb = labels + (size_t)b * L;
    for (int i = 0; i < L; ++i) {
      const int v = mask ? (mask[(size_t)b * L + i] != 0) : (lb[i] >= 0.0f);
      it[i].valid = v; it[i].label = v ? (double)lb[i] : 0.0; it[i].index = i;
    }
    qsort(it, (size_t)L, sizeof(item_t), by_valid_label_index);
    double mx = -INFINITY;
    for (int p = 0; p < L; ++p) {
      x[p] = it[p].valid ? (double)lg[it[p].index] / (double)temperature : LOG_EPS;
      if (x[p] > mx) mx = x[p];
    }
    double loss = 0.0;
    for (int p = 0; p < L; ++p) {
      double s = 0.0;
      for (int q = p; q < L; ++q) s += exp(x[q] - mx);
      S[p] = s;
      const double w = pos_weight ? (double)pos_weight[p] : 1.0;
      loss += w * (log(s) + mx - x[p]);
    }
    loss_out[b] = (float)loss;
    if (dlogits_out) {
      for (int k = 0; k < L; ++k) {                 /* position k */
        double c = 0.0;
        for (int p = 0; p <= k; ++p) c += (pos_weight ? (double)pos_weight[p] : 1.0) / S[p];
        const double w = pos_weight ? (double)pos_weight[k] : 1.0;
        const double g = exp(x[k] - mx) * c - w;
        dlogits_out[(size_t)b * L + it[k].index] = it[k].valid ? (float)(g / (double)temperature) : 0.0f;
      }
    }
    free(it); free(x); free(S);
  }
  return 0;
}

/* loss[b] = sum_i (2^{l_i} - 1) (log(e^{s_i} + sum_{j: l_j < l_i} e^{s_j}) - s_i) over valid i, j; s = logit / T. */
int tfr_c_unique_softmax_f64(const float* logits, const float* labels, const unsigned char* mask, int B, int L,
                             float temperature, float* loss_out, float* dlogits_out) {
  if (!logits || !labels || !loss_out || B < 0 || L <= 0 || !(temperature > 0.0f)) return -1;
#pragma omp parallel for schedule(dynamic)
  for (int b = 0; b < B; ++b) {
    const float* lg = logits + (size_t)b * L;
    const float* lb = labels + (size_t)b * L;
    double* s = (double*)malloc(sizeof(double) * (size_t)L);
    double* D = (double*)malloc(sizeof(double) * (size_t)L);
    int* v = (int*)malloc(sizeof(int) * (size_t)L);
    double mx = -INFINITY;
    for (int i = 0; i < L; ++i) {
      v[i] = mask ? (mask[(size_t)b * L + i] != 0) : (lb[i] >= 0.0f);
      s[i] = (double)lg[i] / (double)temperature;
      if (v[i] && s[i] > mx) mx = s[i];
    }
    double loss = 0.0;
    for (int i = 0; i < L; ++i) {
      D[i] = 1.0;
      if (!v[i]) continue;
      double d = exp(s[i] - mx);
      for (int j = 0; j < L; ++j)
        if (v[j] && lb[j] < lb[i]) d += exp(s[j] - mx);
      D[i] = d;
      loss += (exp2((double)lb[i]) - 1.0) * (log(d) + mx - s[i]);
    }
    loss_out[b] = (float)loss;
    if (dlogits_out) {
      for (int k = 0; k < L; ++k) {
        double g = 0.0;
        if (v[k]) {
          const double gk = exp2((double)lb[k]) - 1.0;
          double acc = gk / D[k];
          for (int i = 0; i < L; ++i)
            if (v[i] && lb[i] > lb[k]) acc += (exp2((double)lb[i]) - 1.0) / D[i];
          g = -gk + exp(s[k] - mx) * acc;
        }
        dlogits_out[(size_t)b * L + k] = (float)(g / (double)temperature);
      }
    }
    free(s); free(D); free(v);
  }
  return 0;
}

/* CircleLoss (tensorflow_ranking/python/losses_impl.py:1036-1116): scores s (already clipped to [0, 1] by get_logits when
 * `clip` != 0, :1082-1085), alpha_i = relu(1 + margin - s_i), alpha'_j = relu(s_j + margin) held constant,
 *   loss[b] = log1p( sum_{valid i, j: y_i > y_j} exp(gamma (alpha_i (1 - margin - s_i) + alpha'_j (s_j - margin))) ).
 * dlogits = d loss / d raw logits (the clip passes the gradient on [0, 1]).  The double sum runs in fp64 with the exponent's
 * maximum pulled out, so gamma = 64 does not overflow (the fp32 reference overflows to inf near s = 1; the kernel and
 * this arbiter do not).  has_pair[b] = 1 when the list has a preference pair (the reference's weight is 0 / 0 = NaN
 * without one, :1109-1111). */
int tfr_c_circle_f64(const float* logits, const float* labels, const unsigned char* mask, int B, int L, float gamma,
                     float margin, int clip, float* loss_out, unsigned char* has_pair_out, float* dlogits_out) {
  if (!logits || !labels || !loss_out || B < 0 || L <= 0) return -1;
#pragma omp parallel for schedule(dynamic)
  for (int b = 0; b < B; ++b) {
    const float* lg = logits + (size_t)b * L;
    const float* lb = labels + (size_t)b * L;
    double* a = (double*)malloc(sizeof(double) * (size_t)L);   /* gamma * alpha_i (1 - margin - s_i) */
    double* c = (double*)malloc(sizeof(double) * (size_t)L);   /* gamma * alpha'_j (s_j - margin)   */
    double* s = (double*)malloc(sizeof(double) * (size_t)L);
    int* v = (int*)malloc(sizeof(int) * (size_t)L);
    for (int i = 0; i < L; ++i) {
      v[i] = mask ? (mask[(size_t)b * L + i] != 0) : (lb[i] >= 0.0f);
      double x = (double)lg[i];
      if (clip) x = x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x);
      s[i] = x;
      const double ai = fmax(1.0 - x + (double)margin, 0.0), aj = fmax(x + (double)margin, 0.0);
      a[i] = (double)gamma * ai * (1.0 - x - (double)margin);
      c[i] = (double)gamma * aj * (x - (double)margin);
    }
    double mx = -INFINITY;
    for (int i = 0; i < L; ++i)
      for (int j = 0; j < L; ++j)
        if (v[i] && v[j] && lb[i] > lb[j] && a[i] + c[j] > mx) mx = a[i] + c[j];
    const int any = mx > -INFINITY;
    double W = 0.0;                                             /* sum of exp(. - mx) */
    if (any)
      for (int i = 0; i < L; ++i)
        for (int j = 0; j < L; ++j)
          if (v[i] && v[j] && lb[i] > lb[j]) W += exp(a[i] + c[j] - mx);
    /* log1p(W e^mx) = mx + log(W) + log1p(e^{-(mx + log W)}) for large arguments, direct otherwise */
    const double lw = any ? mx + log(W) : -INFINITY;
    loss_out[b] = (float)(any ? (fmax(lw, 0.0) + log1p(exp(-fabs(lw)))) : 0.0);
    if (has_pair_out) has_pair_out[b] = (unsigned char)any;
    if (dlogits_out) {
      const double sig = any ? 1.0 / (1.0 + exp(-lw)) : 0.0;     /* W' / (1 + W'), W' = the un-shifted sum */
      for (int k = 0; k < L; ++k) {
        double g = 0.0;
        const int inside = !clip || ((double)lg[k] >= 0.0 && (double)lg[k] <= 1.0);
        if (any && v[k] && inside) {
          double hi = 0.0, lo = 0.0;                           /* k as the preferred item / as the other one */
          for (int j = 0; j < L; ++j) if (v[j] && lb[k] > lb[j]) hi += exp(a[k] + c[j] - lw);
          for (int i = 0; i < L; ++i) if (v[i] && lb[i] > lb[k]) lo += exp(a[i] + c[k] - lw);
          const double ai = fmax(1.0 - s[k] + (double)margin, 0.0), aj = fmax(s[k] + (double)margin, 0.0);
          g = (double)gamma * sig * (-ai * hi + aj * lo);
        }
        dlogits_out[(size_t)b * L + k] = (float)g;
      }
    }
    free(a); free(c); free(s); free(v);
  }
  return 0;
}
