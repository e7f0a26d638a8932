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
