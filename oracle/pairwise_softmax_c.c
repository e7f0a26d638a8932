/* TEST INFRASTRUCTURE ONLY -- plain-C (fp64 inside) restatements of three more rows of the hot path, independent of
 * the torch restatement in tfr_ref.py; loaded by tests/ only (oracle/c_ref.py), never by the product.
 *
 * (1) PairwiseLogisticLoss with the Keras NDCGLambdaWeight defaults (config 3).  tensorflow_ranking/python:
 *     losses_impl.py:483-500  _compute_ranks: invalid scores := min(s) - 1e-6 (fp32 arithmetic, as the reference
 *                             computes it), 1-based ranks by descending score, ties -> lower index first
 *     losses_impl.py:503-537  _pairwise_comparison: pair (i, j) counts when label_i > label_j and both are valid
 *     losses_impl.py:255-279  pair_weights: |gain_i - gain_j| * inverse_max_dcg * pair_rank_discount * list_size,
 *                             gain = 2^l - 1, rank discount D(r) = ln 2 / log1p(r) (keras/losses.py:197-212)
 *     losses_impl.py:334-369  _pair_rank_discount with smooth_fraction = 0, topn = list_size:
 *                             |D(|r_i - r_j|) - D(|r_i - r_j| + 1)|
 *     losses_impl.py:109-134  inverse_max_dcg over the cleaned labels (invalid -> 0)
 *     losses_impl.py:933-940  loss_ij = relu(-d) + log1p(exp(-|d|)), d = s_i - s_j; the weights carry no gradient
 *     out[b] = sum_ij w_ij * loss_ij (the [B, L, L] product summed per list); dlogits = d(sum_b out[b]) / d logits.
 *
 * (2) SoftmaxLoss without lambda weight (config 2).  losses_impl.py:1119-1197:
 *     labels: invalid -> 0; logits: invalid -> log(1e-10); lists whose labels sum to <= 0 get labels 1e-10 on
 *     the valid items; p = labels / sum; loss_b = -sum_i p_i * log_softmax(s)_i; weight_b = sum of the labels.
 *     dlogits = d(sum_b loss_b) / d logits = (sum p) * softmax(s) - p  on every position (invalid ones included:
 *     the reference's `where` stops their gradient, so they are reported as 0).
 */
#include <math.h>
#include <stdlib.h>

typedef struct { float score; int index; } Item;

static int by_score_desc(const void* a, const void* b) {
  const Item* x = (const Item*)a; const Item* y = (const Item*)b;
  if (x->score > y->score) return -1;
  if (x->score < y->score) return 1;
  return (x->index > y->index) - (x->index < y->index);      /* ties: lower index first */
}

static int dbl_desc(const void* a, const void* b) {
  const double x = *(const double*)a, y = *(const double*)b;
  return (x < y) - (x > y);
}

static double disc(double r) { return log(2.0) / log1p(r); }

int tfr_c_pairwise_logistic_ndcg_f64(const float* logits, const float* labels, const unsigned char* mask, int B, int L,
                                     float temperature, float* out, float* dlogits_out) {
  if (!logits || !labels || !out || B < 0 || L <= 0 || !(temperature > 0)) return -1;
#pragma omp parallel
  {
    Item* items = (Item*)malloc((size_t)L * sizeof(Item));
    int* rank = (int*)malloc((size_t)L * sizeof(int));
    double* gain = (double*)malloc((size_t)L * sizeof(double));
    double* sorted = (double*)malloc((size_t)L * sizeof(double));
    double* grad = (double*)malloc((size_t)L * sizeof(double));
    unsigned char* valid = (unsigned char*)malloc((size_t)L);
    float* s = (float*)malloc((size_t)L * sizeof(float));
#pragma omp for schedule(dynamic, 4)
    for (int b = 0; b < B; ++b) {
      const float* lg = logits + (size_t)b * L;
      const float* lb = labels + (size_t)b * L;
      float smin = INFINITY;
      for (int i = 0; i < L; ++i) {
        valid[i] = mask ? mask[(size_t)b * L + i] != 0 : lb[i] >= 0.0f;
        s[i] = lg[i] / temperature;                            /* get_logits, fp32 like the reference */
        if (s[i] < smin) smin = s[i];
      }
      const float fill = -1e-6f + smin;                        /* fp32: may round back onto smin */
      for (int i = 0; i < L; ++i) {
        items[i].score = valid[i] ? s[i] : fill;
        items[i].index = i;
        gain[i] = valid[i] ? pow(2.0, (double)lb[i]) - 1.0 : 0.0;   /* cleaned label 0 -> gain 0 */
        sorted[i] = gain[i];
        grad[i] = 0.0;
      }
      qsort(items, (size_t)L, sizeof(Item), by_score_desc);
      for (int p = 0; p < L; ++p) rank[items[p].index] = p + 1;
      qsort(sorted, (size_t)L, sizeof(double), dbl_desc);
      double ideal = 0.0;
      for (int p = 0; p < L; ++p) ideal += sorted[p] * disc((double)(p + 1));
      const double inv = ideal > 0.0 ? 1.0 / ideal : 0.0;
      double total = 0.0;
      for (int i = 0; i < L; ++i) {
        if (!valid[i]) continue;
        for (int j = 0; j < L; ++j) {
          if (!valid[j] || !(lb[i] > lb[j])) continue;
          const double rd = fabs((double)(rank[i] - rank[j]));
          const double pd = rd > 0.0 ? fabs(disc(rd) - disc(rd + 1.0)) : 0.0;
          const double w = fabs(gain[i] - gain[j]) * inv * pd * (double)L;
          const double d = (double)s[i] - (double)s[j];
          total += w * ((d < 0.0 ? -d : 0.0) + log1p(exp(-fabs(d))));
          const double dd = -w / (1.0 + exp(d));               /* d softplus(-d) / dd */
          grad[i] += dd; grad[j] -= dd;
        }
      }
      out[b] = (float)total;
      if (dlogits_out)
        for (int i = 0; i < L; ++i) dlogits_out[(size_t)b * L + i] = (float)(grad[i] / (double)temperature);
    }
    free(items); free(rank); free(gain); free(sorted); free(grad); free(valid); free(s);
  }
  return 0;
}

int tfr_c_softmax_f64(const float* logits, const float* labels, const unsigned char* mask, int B, int L,
                      float temperature, float* loss_out, float* weight_out, float* dlogits_out) {
  if (!logits || !labels || !loss_out || !weight_out || B < 0 || L <= 0 || !(temperature > 0)) return -1;
  const double log_eps = log(1e-10);
#pragma omp parallel for schedule(dynamic, 16)
  for (int b = 0; b < B; ++b) {
    const float* lg = logits + (size_t)b * L;
    const float* lb = labels + (size_t)b * L;
    double label_sum = 0.0, smax = -INFINITY;
    for (int i = 0; i < L; ++i) {
      const int v = mask ? mask[(size_t)b * L + i] != 0 : lb[i] >= 0.0f;
      const double s = v ? (double)lg[i] / (double)temperature : log_eps;
      if (v) label_sum += (double)lb[i];
      if (s > smax) smax = s;
    }
    double z = 0.0, psum = 0.0;
    for (int i = 0; i < L; ++i) {
      const int v = mask ? mask[(size_t)b * L + i] != 0 : lb[i] >= 0.0f;
      const double s = v ? (double)lg[i] / (double)temperature : log_eps;
      z += exp(s - smax);
      if (v) psum += label_sum > 0.0 ? (double)lb[i] : 1e-10;
    }
    const double lse = smax + log(z);
    double loss = 0.0;
    for (int i = 0; i < L; ++i) {
      const int v = mask ? mask[(size_t)b * L + i] != 0 : lb[i] >= 0.0f;
      const double s = v ? (double)lg[i] / (double)temperature : log_eps;
      const double p = v && psum > 0.0 ? (label_sum > 0.0 ? (double)lb[i] : 1e-10) / psum : 0.0;
      loss -= p * (s - lse);
      if (dlogits_out) {
        const double total_p = psum > 0.0 ? 1.0 : 0.0;         /* sum of the normalised labels */
        dlogits_out[(size_t)b * L + i] = v ? (float)((total_p * exp(s - lse) - p) / (double)temperature) : 0.0f;
      }
    }
    loss_out[b] = (float)loss;
    weight_out[b] = (float)label_sum;
  }
  return 0;
}

/* (3) NDCG@topn and MRR@topn per list, unweighted (metrics_impl.py:210-291 preparation, :429-459 MRR, :631-670 NDCG,
 *     :589-628 _discounted_cumulative_gain; utils.py:115-164 sort_by_scores with the deterministic tie rule: descending
 *     score, ties -> lower index first, invalid items after every valid one).  gain = 2^l - 1, discount = 1 / log2(1 + rank);
 *     NDCG = DCG / ideal DCG (0 when the ideal is 0); MRR = 1 / rank of the first item with label >= 1 (0 if none).
 *     topn <= 0 means the whole list.  fp64 inside: the torch restatement and the HIP kernel agree with each other bit
 *     for bit through a shared fp32 summation order; this one pins the value itself. */
int tfr_c_ndcg_mrr_f64(const float* predictions, const float* labels, const unsigned char* mask, int B, int L, int topn,
                       float* ndcg_out, float* mrr_out) {
  if (!predictions || !labels || !ndcg_out || !mrr_out || B < 0 || L <= 0) return -1;
  const int K = (topn <= 0 || topn > L) ? L : topn;
#pragma omp parallel
  {
    Item* items = (Item*)malloc((size_t)L * sizeof(Item));
    double* ideal = (double*)malloc((size_t)L * sizeof(double));
#pragma omp for schedule(dynamic, 16)
    for (int b = 0; b < B; ++b) {
      const float* pr = predictions + (size_t)b * L;
      const float* lb = labels + (size_t)b * L;
      int n = 0;
      for (int i = 0; i < L; ++i) {
        const int v = mask ? mask[(size_t)b * L + i] != 0 : lb[i] >= 0.0f;
        if (!v) continue;
        items[n].score = pr[i]; items[n].index = i;
        ideal[n] = pow(2.0, (double)lb[i]) - 1.0;
        ++n;
      }
      qsort(items, (size_t)n, sizeof(Item), by_score_desc);
      qsort(ideal, (size_t)n, sizeof(double), dbl_desc);
      double dcg = 0.0, idcg = 0.0, mrr = 0.0;
      for (int p = 0; p < n && p < K; ++p) {
        const double d = log(2.0) / log1p((double)(p + 1));
        const double l = (double)lb[items[p].index];
        dcg += (pow(2.0, l) - 1.0) * d;
        idcg += ideal[p] * d;
        if (mrr == 0.0 && l >= 1.0) mrr = 1.0 / (double)(p + 1);
      }
      ndcg_out[b] = idcg > 0.0 ? (float)(dcg / idcg) : 0.0f;
      mrr_out[b] = (float)mrr;
    }
    free(items); free(ideal);
  }
  return 0;
}
