/* TEST INFRASTRUCTURE ONLY -- a second, independent CPU restatement of the headline path (ApproxNDCG forward and
 * backward) in plain C, next to the torch restatement in tfr_ref.py.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it; the product path never does.
 *
 * What it restates (tensorflow_ranking/python/losses_impl.py):
 *   :77-106   approx_ranks          r_i = 0.5 + sum_j sigmoid(s_j - s_i)            (j = i contributes 0.5)
 *   :33-49    _safe_default_gain_fn g_i = 2^(l_i - max l) - 2^(-max l)
 *   :109-134  inverse_max_dcg       1 / sum_p g_(p) / log1p(p),  labels sorted descending, 0 when the sum is 0
 *   :137-167  ndcg                  sum_i g_i / log1p(r_i) * inverse_max_dcg
 *   :1579-1603 ApproxNDCGLoss._compute_unreduced_loss_impl: invalid items (label < 0 or mask off) get label 0 and the
 *             score min(s) - 1e3, whose sigmoid against any valid score is exactly 0 in fp32; lists whose labels sum
 *             to <= 0 get labels 1e-10 (all gains round to 0 in fp32 => loss 0) and weight 0; loss = -ndcg.
 *   :502-503  get_logits            s = logits / temperature
 * The backward is the analytic derivative of the same expression (the reference uses TF autodiff):
 *   d loss / d s_k = sum_{i != k} (c_i - c_k) * sigma'(s_k - s_i),  c_i = inv * g_i / (log1p(r_i)^2 * (1 + r_i)).
 *
 * Two builds of this one file: the default = the arbiter tfr_c_approx_ndcg_f64 (strict IEEE, fp64 inside);
 * -DTFR_C_FLOAT -Ofast -march=native = tfr_c_approx_ndcg_f32_fast, "what a fused, vectorised CPU loop can do" -- the
 * stricter CPU baseline of bench.py (checked against the arbiter in tests).
 */
#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

#ifdef TFR_C_FLOAT
typedef float real;
#define TFR_C_NAME tfr_c_approx_ndcg_f32_fast
#define R_EXP expf
#define R_LOG1P log1pf
#else
typedef double real;
#define TFR_C_NAME tfr_c_approx_ndcg_f64
#define R_EXP exp
#define R_LOG1P log1p
#endif

static int cmp_desc(const void* a, const void* b) {
  const real x = *(const real*)a, y = *(const real*)b;
  return (x < y) - (x > y);
}

/* One list.  scratch: 5 * L reals. */
static void one_list(const float* logits, const float* labels, const unsigned char* mask, int L, real inv_t,
                     float* loss_out, float* weight_out, float* dlogits, real* scratch) {
  real* s = scratch;            /* scaled scores of the valid items */
  real* g = scratch + L;        /* gains */
  real* r = scratch + 2 * L;    /* approximate ranks */
  real* c = scratch + 3 * L;    /* backward coefficients */
  real* sorted = scratch + 4 * L;
  int idx_n = 0;
  real label_sum = 0, max_label = 0;
  int* idx = (int*)malloc((size_t)L * sizeof(int));
  for (int i = 0; i < L; ++i) {
    const int valid = mask ? mask[i] != 0 : labels[i] >= 0.0f;
    if (dlogits) dlogits[i] = 0.0f;
    if (!valid) continue;
    idx[idx_n] = i;
    s[idx_n] = (real)logits[i] * inv_t;
    g[idx_n] = (real)labels[i];
    label_sum += (real)labels[i];
    ++idx_n;
  }
  /* the max over the row includes the zeros that stand in for invalid items (:1588) */
  max_label = idx_n < L ? 0 : g[0];
  for (int i = 0; i < idx_n; ++i) if (g[i] > max_label) max_label = g[i];
  if (!(label_sum > 0)) {       /* no relevant item: weight 0, gains vanish in fp32 */
    *loss_out = 0.0f; *weight_out = 0.0f;
    free(idx);
    return;
  }
  const real base = (real)pow(2.0, -(double)max_label);
  for (int i = 0; i < idx_n; ++i) {
    g[i] = (real)pow(2.0, (double)(g[i] - max_label)) - base;
    sorted[i] = g[i];
  }
  qsort(sorted, (size_t)idx_n, sizeof(real), cmp_desc);       /* gains are monotone in the labels */
  real ideal = 0;
  for (int p = 0; p < idx_n; ++p) ideal += sorted[p] / R_LOG1P((real)(p + 1));
  const real inv = ideal > 0 ? 1 / ideal : 0;
  real dcg = 0;
  for (int i = 0; i < idx_n; ++i) {
    real acc = 0.5f;
    const real si = s[i];
    for (int j = 0; j < idx_n; ++j) acc += 1 / (1 + R_EXP(si - s[j]));     /* sigmoid(s_j - s_i) */
    r[i] = acc;
    const real lg = R_LOG1P(acc);
    dcg += g[i] / lg;
    c[i] = inv * g[i] / (lg * lg * (1 + acc));
  }
  *loss_out = (float)(-dcg * inv);
  *weight_out = 1.0f;
  if (dlogits) {
    for (int k = 0; k < idx_n; ++k) {
      real acc = 0;
      const real sk = s[k], ck = c[k];
      for (int i = 0; i < idx_n; ++i) {
        const real sg = 1 / (1 + R_EXP(s[i] - sk));                         /* sigmoid(s_k - s_i) */
        acc += (c[i] - ck) * sg * (1 - sg);
      }
      dlogits[idx[k]] = (float)(acc * inv_t);
    }
  }
  free(idx);
}

/* logits, labels [B, L] fp32; mask nullable [B, L] (NULL: label >= 0); loss_out, weight_out [B];
 * dlogits_out nullable [B, L] = d(sum_b loss_b) / d logits.  Returns 0, or -1 on a bad argument. */
int TFR_C_NAME(const float* logits, const float* labels, const unsigned char* mask, int B, int L, float temperature,
               float* loss_out, float* weight_out, float* dlogits_out) {
  if (!logits || !labels || !loss_out || !weight_out || B < 0 || L <= 0 || !(temperature > 0)) return -1;
  const real inv_t = 1 / (real)temperature;
#pragma omp parallel
  {
    real* scratch = (real*)malloc((size_t)5 * L * sizeof(real));
#pragma omp for schedule(dynamic, 8)
    for (int b = 0; b < B; ++b)
      one_list(logits + (size_t)b * L, labels + (size_t)b * L, mask ? mask + (size_t)b * L : NULL, L, inv_t,
               loss_out + b, weight_out + b, dlogits_out ? dlogits_out + (size_t)b * L : NULL, scratch);
    free(scratch);
  }
  return 0;
}

#ifndef TFR_C_FLOAT
int tfr_c_threads(void) { return omp_get_max_threads(); }     /* what "cores" means for a timing of these loops */
#endif
