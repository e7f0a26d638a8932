/* tfr_hip.h -- C ABI of libtfr_hip.so: the MI355X (gfx950) ranking-loss hot path.
 *
 * The reference (tensorflow/ranking) has NO native / FFI boundary: its hot path
 * is a chain of TensorFlow ops behind duck-typed Python classes.  This header
 * is therefore the boundary we DEFINE; each entry point cites the reference
 * Python interface it replaces (paths relative to
 * /root/reference/tensorflow_ranking/python/).  The Python mirror of those
 * classes lives in ranking_amd/ and binds these symbols with ctypes
 * (see INTEGRATION.md for the stub a reference maintainer would add).
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer owned by the caller unless the name
 *     ends in _host; kernels allocate nothing and keep no state;
 *   - tensors are dense row-major fp32 [B, L] unless stated; `mask` is uint8
 *     (0/1) and nullable: when NULL an item is valid iff label >= 0
 *     (utils.py:78-81);
 *   - work is enqueued on `stream` (a hipStream_t passed as void*); the call
 *     never synchronises and is re-entrant;
 *   - return 0 = ok, <0 = invalid argument (TFR_EINVAL -1, TFR_ETOOLARGE -2: list_size beyond the entry point's
 *     TFR_MAX_LIST_SIZE* limit below), >0 = hipError_t from the launch.
 *   - tie rule (the reference shuffles ties at random, utils.py:100-112):
 *     descending score, equal scores by `tiebreak` then by index, invalid last.
 */
#ifndef TFR_HIP_H_
#define TFR_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TFR_MAX_TOPN 8

/* List-size limit (TFR_ETOOLARGE beyond) of every entry point that takes a list_size.  These macros are what the
 * kernels' launchers test; every "list_size <= N" statement in this header is checked against them by
 * tests/test_host_logic.py. */
#define TFR_MAX_LIST_SIZE 8192
/* Up to these list sizes the workgroup kernels keep a list's working arrays in LDS (160 KB per workgroup); beyond, up to
 * TFR_MAX_LIST_SIZE, the arrays live in a WORKSPACE the caller passes (`workspace`, `workspace_bytes`: device memory,
 * contents irrelevant before and after the call, one per stream in flight).  tfr_list_workspace_bytes(op, L) is what ONE
 * list in flight needs (0: none at this list size); a launch runs min(B, workspace_bytes / that, 1024) lists
 * concurrently.  Without a workspace of at least one slot such a list size is TFR_ETOOLARGE. */
#define TFR_LDS_LIST_SIZE_METRIC 4096       /* rank metrics other than NDCG / MRR, diversity metrics: 28-32 B of LDS per item */
#define TFR_LDS_LIST_SIZE_LISTWISE 4096     /* ListMLE, UniqueSoftmax, Circle: LDS block sort + scans, 24-36 B per item */
#define TFR_LDS_LIST_SIZE_NEURAL_SORT 2048  /* NeuralSort losses: one wavefront per list, 40 / 60 B of LDS per item */
#define TFR_WS_LIST_MLE 0
#define TFR_WS_UNIQUE_SOFTMAX 1
#define TFR_WS_CIRCLE 2
#define TFR_WS_RANK_METRIC 3
#define TFR_WS_DIV_METRIC 4
#define TFR_WS_NEURAL_SORT_NDCG 5
#define TFR_WS_NEURAL_SORT_CE 6
long tfr_list_workspace_bytes(int op, int L);

/* gain_kind */
#define TFR_GAIN_IDENTITY 0   /* keras/utils.py:51  identity                  */
#define TFR_GAIN_POW2M1 1     /* keras/utils.py:79  2^l - 1                   */
#define TFR_GAIN_CUSTOM 2     /* caller passes gains[B,L] = gain_fn(clean l)  */

/* loss_kind of tfr_pairwise_loss_f32 */
#define TFR_PAIR_LOGISTIC 0
#define TFR_PAIR_HINGE 1
#define TFR_PAIR_SOFT_ZERO_ONE 2
#define TFR_PAIR_MSE 3           /* losses_impl.py:961-998: ((s_i - s_j) - (y_i - y_j))^2 over all i != j */
/* OR-ed into loss_kind TFR_PAIR_LOGISTIC: the reference's gradient at EXACTLY tied scores.  losses_impl.py:936-940 writes the
 * pair loss as relu(-t) + log1p(exp(-|t|)); under TF autodiff that differentiates to 0 at t = 0 (relu'(0) = 0, sign(0) = 0),
 * not to the analytic -sigma(0) = -1/2 the kernels return by default.  With the flag a pair with s_i == s_j contributes its
 * loss (log 2) and no gradient -- what a reference run does on an all-equal initial logit vector.  The flag declines the
 * LambdaRank fast paths (their factorised exponentials cannot see an exact tie): general wave / workgroup kernels. */
#define TFR_PAIR_TIED_ZERO 0x100

/* lambda_kind */
#define TFR_LAMBDA_NONE 0
#define TFR_LAMBDA_DCG 2      /* losses_impl.py:299-369 DCGLambdaWeight       */
#define TFR_LAMBDA_LABELDIFF 1 /* losses_impl.py:210-217                      */
#define TFR_LAMBDA_DCG_V2 3    /* losses_impl.py:372-394 DCGLambdaWeightV2     */
#define TFR_LAMBDA_YETI_DCG 4  /* losses_impl.py:397-407 YetiDCGLambdaWeight   */
#define TFR_LAMBDA_PRECISION 5 /* losses_impl.py:410-454 PrecisionLambdaWeight: `gains` = positive_fn(labels)
                                  as 0/1 floats (TFR_GAIN_CUSTOM), topn mandatory */

/* Version of this header's ABI.  2 (round 5): the `*_sum_f32` entry points below; `tfr_tower_dropout` carries the device
 * pointer `step` since round 4 (24 bytes, was 12 -- a caller built against version 1 passes a short struct).  A binding
 * must check the value at load time (ranking_amd/_lib.py does). */
#define TFR_HIP_ABI_VERSION 2
int tfr_hip_abi_version(void);

/* utils.sort_by_scores / utils.sorted_ranks / losses_impl._compute_ranks
 * (utils.py:115-195, losses_impl.py:483-500).
 *   valid_i = mask ? mask_i : (labels ? labels_i >= 0 : 1)
 *   order_out[b, p]  = index of the item at sorted position p (nullable)
 *   ranks_out[b, i]  = 1-based rank of item i (nullable)
 *   tiebreak[b, i]   = optional int32 secondary key in [0, 32768) (nullable). */
int tfr_sort_ranks_f32(const float* scores, const float* labels, const uint8_t* mask,
                       const int32_t* tiebreak, int B, int L,
                       int32_t* ranks_out, int32_t* order_out, void* stream);

/* metrics_impl.NDCGMetric.compute (metrics_impl.py:228-291, 631-670) for up
 * to TFR_MAX_TOPN cutoffs at once.
 *   weights      nullable; [B, L] per item, or [B] per list when weights_per_list
 *   gains        nullable [B, L] = gain_fn(label or 0 if masked) for a custom
 *                gain_fn; NULL -> 2^l - 1 evaluated in-kernel
 *   discount     [L] fp32: rank_discount_fn(r), r = 1..L (host computed table)
 *   topn_host    HOST array of K cutoffs (<= 0 means L)
 *   ndcg_out     [K, B]
 *   stats_out    [B, 3] = (sum w, sum gain, sum w*gain) in tree_sum order; the
 *                cross-list part of _per_example_weights_to_per_list_weights
 *                (metrics_impl.py:63-119) is done by the caller on [B] vectors.
 *   tie_seed     (every metric entry point) the reference sorts the predictions with shuffle_ties=True
 *                (utils.py:84-164): equal predictions in a random order.  tie_seed != 0 orders them by a counter-based
 *                15-bit hash of (tie_seed, list, item), then by index (NDCG then runs on its sort kernel instead of the
 *                counting / bucket forms); tie_seed == 0 keeps index order.
 * NDCG and MRR (below): list_size <= 8192 (TFR_MAX_LIST_SIZE) -- one wavefront per list up to 512 / 256 items, one
 * workgroup with 16 B of LDS per item beyond. */
int tfr_ndcg_metric_f32(const float* labels, const float* predictions, const float* weights,
                        int weights_per_list, const uint8_t* mask, const float* gains,
                        const float* discount, const int32_t* topn_host, int K, int B, int L,
                        float* ndcg_out, float* stats_out, uint32_t tie_seed, void* stream);

/* The sort-based metrics of metrics_impl.py behind one entry point; `kind`:
 *   TFR_METRIC_NDCG / TFR_METRIC_MRR  = the two entry points above;
 *   TFR_METRIC_DCG (:673-705)        metric_out = sum_{p<k} w gain discount (the caller divides by the list weight)
 *   TFR_METRIC_HITS (:462-506)       TFR_METRIC_RECALL (:154-177, 539-561)   TFR_METRIC_PRECISION (:180-207, 564-586)
 *   TFR_METRIC_MAP (:589-628)        TFR_METRIC_ARP (:509-536; stats_out[:, 2] = its per-list weight)
 * stats_out [B, 3] = (sum w, sum rel, sum w*rel) with rel = gain (DCG), label (ARP) or 1{label >= 1}.
 * NDCG / MRR as above (no workspace at any size); the other kinds: one wavefront per list up to 512 items, one workgroup
 * (32 B of LDS per item) beyond; workspace (TFR_WS_RANK_METRIC) above TFR_LDS_LIST_SIZE_METRIC. */
#define TFR_METRIC_NDCG 0
#define TFR_METRIC_MRR 1
#define TFR_METRIC_DCG 2
#define TFR_METRIC_HITS 3
#define TFR_METRIC_RECALL 4
#define TFR_METRIC_PRECISION 5
#define TFR_METRIC_MAP 6
#define TFR_METRIC_ARP 7
#define TFR_METRIC_BPREF 8          /* :825-898 BPrefMetric, TREC version (denominator min(R, N)) */
#define TFR_METRIC_BPREF_NONTREC 9  /*          use_trec_version=False (denominator R)            */
#define TFR_METRIC_PWA 10           /* :901-965 PWAMetric (the caller supplies mean(weights) as the list weight) */
#define TFR_METRIC_OPA 11           /* :708-743 OPAMetric: K = 1, stats_out[:, 2] = sum of pair weights     */
int tfr_rank_metric_f32(int kind, const float* labels, const float* predictions, const float* weights,
                        int weights_per_list, const uint8_t* mask, const float* gains,
                        const float* discount, const int32_t* topn_host, int K, int B, int L,
                        float* metric_out, float* stats_out, uint32_t tie_seed, void* workspace, long workspace_bytes,
                        void* stream);

/* Diversity metrics on subtopic labels [B, L, S] (metrics_impl.py:313-426, :746-822).
 *   TFR_DIV_ALPHA_DCG    AlphaDCGMetric: metric_out[q*B+b] = sum_{p<k} w gain discount, gain = sum_s y_ps (1-alpha)^{cum_s};
 *                        discount[p] = rank_discount_fn(p + 1); the caller divides by the per-list weight
 *   TFR_DIV_PRECISION_IA PrecisionIAMetric: the metric itself
 * mask [B, L] or NULL (then an item is valid when any of its subtopic labels is >= 0); stats_out as above
 * with relevance = any_s [y >= 1].  Workspace (TFR_WS_DIV_METRIC) above TFR_LDS_LIST_SIZE_METRIC. */
#define TFR_DIV_ALPHA_DCG 0
#define TFR_DIV_PRECISION_IA 1
int tfr_div_metric_f32(int kind, const float* labels, const float* predictions, const float* weights,
                       int weights_per_list, const uint8_t* mask, const float* discount, float alpha,
                       const int32_t* topn_host, int K, int B, int L, int S, float* metric_out,
                       float* stats_out, uint32_t tie_seed, void* workspace, long workspace_bytes, void* stream);

/* Per-list metric weights [B] from the stats_out [B, 3] of the metric entry points above
 * (metrics_impl.py:63-119 _per_example_weights_to_per_list_weights): sum(w rel)/sum(rel); lists without
 * relevance get the batch mean over the lists that have it (1 if none has); lists with all-zero weights get 0. */
int tfr_metric_list_weights_f32(const float* stats, int B, float* weights_out, void* stream);

/* metrics_impl.MRRMetric.compute (metrics_impl.py:429-459).
 *   mrr_out [K, B]; stats_out [B, 3] = (sum w, sum rel, sum w*rel), rel = 1{l>=1}. */
int tfr_mrr_metric_f32(const float* labels, const float* predictions, const float* weights,
                       int weights_per_list, const uint8_t* mask, const int32_t* topn_host,
                       int K, int B, int L, float* mrr_out, float* stats_out, uint32_t tie_seed, void* stream);

/* losses_impl.ApproxNDCGLoss._compute_unreduced_loss_impl fused with its
 * backward (losses_impl.py:77-167, 1579-1603).
 *   x = logits / temperature is applied inside (losses_impl.py:773-785)
 *   inv_log1p    [L] fp32: 1/log1p(r), r = 1..L (host computed table)
 *   list_scale   nullable [B]: dlogits_out rows are multiplied by it
 *   loss_out     [B]  -ApproxNDCG per list
 *   weight_out   [B]  1{sum label > 0}
 *   dlogits_out  nullable [B, L] = list_scale_b * d loss_b / d logits[b, :]
 *   lanes_per_row  tuning knob (1,2,4,..,64; 0 = default). */
int tfr_approx_ndcg_f32(const float* logits, const float* labels, const uint8_t* mask,
                        const float* inv_log1p, const float* list_scale, int B, int L,
                        float temperature, int lanes_per_row, float* loss_out, float* weight_out,
                        float* dlogits_out, const int32_t* list_order, void* stream);

/* tfr_approx_ndcg_f32 that also returns the reduced scalar without a launch of its own:
 *   loss_sum_out [1]  sum_b loss_out[b] * list_scale[b] (list_scale NULL: sum_b loss_out[b]), added in a fixed order
 *                     by the last workgroup to finish its forward pass (compute_weighted_loss / the Keras reduction,
 *                     losses_impl.py:787-814, keras/losses.py:264-280)
 *   ticket       [tfr_grid_sum_state_ints()] uint32 in device memory (group tickets + partial sums), zero before the
 *                     first launch, left zero; one per stream in flight. */
int tfr_grid_sum_state_ints(void);
int tfr_approx_ndcg_sum_f32(const float* logits, const float* labels, const uint8_t* mask,
                            const float* inv_log1p, const float* list_scale, int B, int L,
                            float temperature, int lanes_per_row, float* loss_out, float* weight_out,
                            float* dlogits_out, const int32_t* list_order, float* loss_sum_out,
                            uint32_t* ticket, void* stream);

/* The same reduced scalar from the other loss launches (round 5; `ticket` as above, zero before the first launch, left zero,
 * one per stream in flight; every output of the plain entry point is still written, bit for bit the same):
 *   tfr_softmax_loss_sum_f32        sum_b loss_out[b] * weight_out[b]   (keras/losses.py:824-832); `sum_scratch` = B floats
 *                                   of device scratch (the per-contributor products; the streaming form adds up one
 *                                   partial per wavefront, in the order the wavefront walks its lists).
 *                                   PARTIALS MODE: loss_sum_out == NULL and ticket == NULL -- the launch only leaves the
 *                                   n = tfr_softmax_sum_contributors(...) per-contributor values in sum_scratch[0 .. n) and
 *                                   the caller adds them with tfr_list_dot_f32(sum_scratch, NULL, n): no device-memory
 *                                   tickets at the end of a 5-25 us kernel (measured: the ticket chain is five dependent
 *                                   memory round trips, +16 us behind a softmax launch -- DESIGN 4), and a reduction over
 *                                   8 192 values instead of 2 x 65 536.
 *   tfr_pairwise_loss_sum_f32       sum_b list_loss_out[b] (list_loss_out must be given: its entries are what is added up)
 *   tfr_list_mle_sum_f32 / tfr_unique_softmax_sum_f32    sum_b loss_out[b] * list_scale[b] (list_scale NULL: plain sum)
 *   tfr_pointwise_loss_sum_f32      sum_b list_loss_out[b]
 * Summation order: fixed by (B, launch geometry), independent of which workgroup finishes last: the same bits on every
 * run.  B == 0: loss_sum_out[0] = 0. */
int tfr_softmax_loss_sum_f32(const float* logits, const float* labels, const uint8_t* mask,
                             const float* item_weights, int weights_per_list, int lambda_kind,
                             int topn, int normalized, int gain_kind, const float* gains,
                             const float* discount, int B, int L, float temperature, float poly_epsilon,
                             float* loss_out, float* weight_out, float* dlogits_out, float* loss_sum_out,
                             float* sum_scratch, uint32_t* ticket, void* stream);
/* has_weights: 0 = none; 1 = one per list; 2 = one per item */
int tfr_softmax_sum_contributors(int B, int L, int has_mask, int has_weights, int lambda_kind, int want_grad);
int tfr_pairwise_loss_sum_f32(int loss_kind, const float* logits, const float* labels, const uint8_t* mask,
                              const float* item_weights, const float* list_weights,
                              int lambda_kind, int topn, float smooth_fraction,
                              int normalized, int gain_kind, const float* gains,
                              const float* discount, int B, int L, float temperature,
                              float* row_loss_out, float* row_weight_out, float* nnz_out,
                              float* dlogits_out, const int32_t* list_order, float* list_loss_out,
                              float* loss_sum_out, uint32_t* ticket, uint32_t tie_seed, void* stream);
int tfr_list_mle_sum_f32(const float* logits, const float* labels, const uint8_t* mask,
                         const float* pos_weight, const float* list_scale, int B, int L,
                         float temperature, float* loss_out, float* dlogits_out, float* loss_sum_out,
                         uint32_t* ticket, uint32_t tie_seed, void* workspace, long workspace_bytes, void* stream);
int tfr_unique_softmax_sum_f32(const float* logits, const float* labels, const uint8_t* mask,
                               const float* list_scale, int B, int L, float temperature,
                               float* loss_out, float* dlogits_out, float* loss_sum_out, uint32_t* ticket,
                               void* workspace, long workspace_bytes, void* stream);
int tfr_pointwise_loss_sum_f32(int kind, const float* logits, const float* labels, const uint8_t* mask,
                               const float* item_weights, const float* list_weights, int B, int L,
                               float temperature, float* list_loss_out, float* list_weight_out,
                               float* list_nnz_out, float* dlogits_out, float* loss_sum_out, uint32_t* ticket,
                               void* stream);

/* Longest-first launch order for the O(n^2) loss kernels (their `list_order` argument, nullable):
 * order_out[B] = list indices by decreasing number of valid items (64 length classes; arbitrary
 * order inside a class).  Results of the loss kernels
 * do not depend on it (each list is written to its own rows); it shortens the end-of-kernel tail.
 *   workspace  int32[B] scratch owned by the caller. */
int tfr_list_order_i32(const float* labels, const uint8_t* mask, int B, int L, int32_t* order_out,
                       int32_t* workspace, void* stream);
/* An APPROXIMATELY longest-first order from one launch (round 6): every workgroup sorts its 128 / 256 lists by length class
 * and writes them interleaved with the other workgroups' (position = rank inside the segment * segments + segment): a
 * permutation of [0, B) whose first `segments` entries are the longest list of every segment, and so on.  No workspace. */
int tfr_list_order_interleaved_i32(const float* labels, const uint8_t* mask, int B, int L, int32_t* order_out,
                                   void* stream);

/* losses_impl.ApproxMRRLoss._compute_unreduced_loss_impl fused with its backward
 * (losses_impl.py:77-106, 1606-1632): loss_b = -sum_i (l_i / sum l) / approx_rank_i; same
 * conventions as tfr_approx_ndcg_f32 (temperature applied inside, weight_out = 1{sum label > 0}). */
int tfr_approx_mrr_f32(const float* logits, const float* labels, const uint8_t* mask,
                       const float* list_scale, int B, int L, float temperature, float* loss_out,
                       float* weight_out, float* dlogits_out, const int32_t* list_order, void* stream);

/* losses_impl.ListMLELoss._compute_unreduced_loss_impl fused with its backward
 * (losses_impl.py:1541-1576; ListMLELambdaWeight :457-480).
 *   pos_weight   nullable [L]: rank_discount_fn(p + 1) of a ListMLELambdaWeight (host table)
 *   loss_out     [B] negative log likelihood per list (the list weight is 1)
 *   dlogits_out  nullable [B, L] = list_scale_b * d loss_b / d logits[b, :]
 * One wavefront per list up to 1024 items, one workgroup beyond; workspace (TFR_WS_LIST_MLE) above
 * TFR_LDS_LIST_SIZE_LISTWISE.  tie_seed: the reference sorts with shuffle_ties=True (:1558-1561, utils.py:84-112) -- equal
 * labels in a random order, new in every step; tie_seed != 0 orders them by a counter-based 15-bit hash of (tie_seed,
 * list, item) (then by index), tie_seed == 0 keeps index order. */
int tfr_list_mle_f32(const float* logits, const float* labels, const uint8_t* mask,
                     const float* pos_weight, const float* list_scale, int B, int L,
                     float temperature, float* loss_out, float* dlogits_out, uint32_t tie_seed, void* workspace,
                     long workspace_bytes, void* stream);

/* losses_impl.UniqueSoftmaxLoss._compute_unreduced_loss_impl fused with its backward
 * (losses_impl.py:1250-1281): loss_b = sum_i (2^l_i - 1) (log(e^s_i + sum_{j: l_j < l_i} e^s_j) - s_i).
 * Same conventions as tfr_list_mle_f32 (list weight 1; workspace TFR_WS_UNIQUE_SOFTMAX). */
int tfr_unique_softmax_f32(const float* logits, const float* labels, const uint8_t* mask,
                           const float* list_scale, int B, int L, float temperature,
                           float* loss_out, float* dlogits_out, void* workspace, long workspace_bytes, void* stream);

/* NeuralSort losses (losses_impl.py:1635-1673 NeuralSortCrossEntropyLoss, :1676-1713 NeuralSortNDCGLoss,
 * :1716-1801 neural_sort): per-list loss [B] and d loss / d logits [B, L] (x list_scale[b] when given),
 * no [L, L] tensor materialised.  inv_log1p[r] = 1 / log1p(r + 1), r < L (NDCG kind only).  One wavefront per list
 * (40 / 60 B of LDS per item) up to 1024 items, one workgroup per list beyond: everything in LDS up to
 * TFR_LDS_LIST_SIZE_NEURAL_SORT items, the row statistics in the workspace (TFR_WS_NEURAL_SORT_NDCG / _CE) above.  The
 * Gumbel variants are this kernel on the sampler's expanded batch. */
#define TFR_NEURAL_SORT_NDCG 0
#define TFR_NEURAL_SORT_CE 1
int tfr_neural_sort_loss_f32(int kind, const float* logits, const float* labels, const uint8_t* mask,
                             const float* inv_log1p, const float* list_scale, int B, int L,
                             float temperature, float* loss_out, float* dlogits_out, void* workspace,
                             long workspace_bytes, void* stream);

/* CircleLoss (losses_impl.py:1036-1116): loss[b] = log1p(sum_{y_i > y_j} exp(gamma (a_i + b_j))) on scores
 * clipped to [0, 1]; weight[b] = 1, or NaN for a list without any preference pair (the reference's 0 / 0);
 * dlogits = d loss / d logits (x list_scale[b]).  clip != 0 applies get_logits' clip_by_value(0, 1) in
 * the kernel (compute()); compute_per_list / compute_unreduced_loss hand the scores over as they are.
 * One wavefront per list up to 1024 items, one workgroup beyond; workspace (TFR_WS_CIRCLE) above
 * TFR_LDS_LIST_SIZE_LISTWISE. */
int tfr_circle_loss_f32(const float* logits, const float* labels, const uint8_t* mask,
                        const float* list_scale, int B, int L, float gamma, float margin, int clip,
                        float* loss_out, float* weight_out, float* dlogits_out, void* workspace, long workspace_bytes,
                        void* stream);

/* Pointwise losses (losses_impl.py:1284-1321 _PointwiseLoss, :1425-1446 SigmoidCrossEntropyLoss, :1449-1469
 * MeanSquaredLoss), forward + backward in one pass: per list sum(w l), sum(w), #(w != 0) and
 * dlogits = d sum(w l) / d logits, with w = (label >= 0 ? item_weight * list_weight : 0) * [mask].
 * item_weights [B, L] / list_weights [B] / mask / the last three outputs are nullable.  Any list size. */
#define TFR_POINT_SIGMOID_CE 0
#define TFR_POINT_MSE 1
int tfr_pointwise_loss_f32(int kind, const float* logits, const float* labels, const uint8_t* mask,
                           const float* item_weights, const float* list_weights, int B, int L,
                           float temperature, float* list_loss_out, float* list_weight_out,
                           float* list_nnz_out, float* dlogits_out, void* stream);

/* losses_impl.PairwiseLogisticLoss (+ optional DCGLambdaWeight pair weights)
 * fused with its backward (losses_impl.py:255-369, 483-537, 863-940).
 *   item_weights nullable [B, L] (w_i multiplies row i, losses_impl.py:917-930)
 *   list_weights nullable [B]   (per-list weight, multiplies every row)
 *   gains        nullable [B, L] when gain_kind == TFR_GAIN_CUSTOM
 *   discount     [L + 1] fp32: rank_discount_fn(r), r = 1..L+1 (DCG lambda only)
 *   topn <= 0 means L
 *   row_loss_out   [B, L] sum_j w_i W_ij loss_ij    (nullable)
 *   row_weight_out [B, L] sum_j w_i W_ij            (nullable)
 *   nnz_out        [B]    #{(i,j): w_i W_ij != 0}    (nullable)
 *   dlogits_out    [B, L] d(sum_ij w_i W_ij loss_ij)/d logits (nullable). */
int tfr_pairwise_logistic_f32(const float* logits, const float* labels, const uint8_t* mask,
                              const float* item_weights, const float* list_weights,
                              int lambda_kind, int topn, float smooth_fraction, int normalized,
                              int gain_kind, const float* gains, const float* discount,
                              int B, int L, float temperature,
                              float* row_loss_out, float* row_weight_out, float* nnz_out,
                              float* dlogits_out, void* stream);

/* The same machinery for the other pairwise losses of losses_impl.py:936-958.
 *   loss_kind  TFR_PAIR_LOGISTIC (PairwiseLogisticLoss), TFR_PAIR_HINGE (PairwiseHingeLoss :943-948,
 *              relu(1 - d)), TFR_PAIR_SOFT_ZERO_ONE (PairwiseSoftZeroOneLoss :951-958, sigma(-d)),
 *              TFR_PAIR_MSE (PairwiseMSELoss :961-998: all ordered pairs of distinct valid items).
 *   lambda_kind additionally accepts TFR_LAMBDA_DCG_V2 / TFR_LAMBDA_YETI_DCG / TFR_LAMBDA_PRECISION.
 *   list_order     nullable [B] launch order (tfr_list_order_i32)
 *   list_loss_out  nullable [B]: sum over the rows of a list of row_loss (what every scalar reduction consumes;
 *                  with row_loss_out = NULL nothing [B, L]-sized is written for the loss).
 * PairwiseLogisticLoss with a DCGLambdaWeight (smooth_fraction 0, no topn, identity / 2^l - 1 gain, no mask) and
 * list_size <= 256 runs the LambdaRank fast path: items re-homed by grade, only the pairs with l_i > l_j visited (from 512
 * lists: the group kernel of csrc/lambdarank_group.h -- several lists per workgroup, shared pair sweeps); any
 * list_size <= 8192 (TFR_MAX_LIST_SIZE) through the general wave / workgroup kernels.
 * tie_seed: the ranks behind the lambda weights are `_compute_ranks(logits, shuffle_ties=True)` in the reference (:483-500):
 * equal scores in a random order.  tie_seed != 0 ranks them by the 15-bit hash of (tie_seed, list, item) (then by index) --
 * on the workgroup kernel, whatever the list size; tie_seed == 0 keeps index order and the fast paths. */
int tfr_pairwise_loss_f32(int loss_kind, const float* logits, const float* labels, const uint8_t* mask,
                          const float* item_weights, const float* list_weights,
                          int lambda_kind, int topn, float smooth_fraction, int normalized,
                          int gain_kind, const float* gains, const float* discount,
                          int B, int L, float temperature,
                          float* row_loss_out, float* row_weight_out, float* nnz_out,
                          float* dlogits_out, const int32_t* list_order, float* list_loss_out, uint32_t tie_seed,
                          void* stream);

/* losses_impl.SoftmaxLoss.precompute + _compute_unreduced_loss_impl fused with
 * the backward (losses_impl.py:1119-1197, 281-296).
 *   item_weights nullable, [B, L] or [B] when weights_per_list
 *   lambda_kind  TFR_LAMBDA_NONE or TFR_LAMBDA_DCG (individual_weights)
 *   loss_out     [B] per-list cross entropy
 *   weight_out   [B] sum of (weighted) labels
 *   dlogits_out  nullable [B, L] = weight_b * d loss_b / d logits. */
int tfr_softmax_loss_f32(const float* logits, const float* labels, const uint8_t* mask,
                         const float* item_weights, int weights_per_list,
                         int lambda_kind, int topn, int normalized, int gain_kind,
                         const float* gains, const float* discount, int B, int L,
                         float temperature, float* loss_out, float* weight_out,
                         float* dlogits_out, void* stream);
/* PolyOneSoftmaxLoss (losses_impl.py:1200-1247): the same with loss += epsilon * (1 - sum_i p_i softmax_i). */
int tfr_poly1_softmax_loss_f32(const float* logits, const float* labels, const uint8_t* mask,
                               const float* item_weights, int weights_per_list, int lambda_kind,
                               int topn, int normalized, int gain_kind, const float* gains,
                               const float* discount, int B, int L, float temperature, float epsilon,
                               float* loss_out, float* weight_out, float* dlogits_out, void* stream);

/* losses_impl.GumbelSampler.sample (losses_impl.py:556-649), dense path.
 *   uniform      nullable [B, S, L] injected U(0,1) noise; NULL -> in-kernel
 *                Philox4x32-10 keyed by (seed, offset)
 *   sampled_out  [B*S, L] = log(softmax((logits + G)/gumbel_temperature) + 1e-20)
 * Backward: dlogits_out[B, L] = sum_s J^T upstream[b*S+s, :]. */
int tfr_gumbel_sample_f32(const float* logits, const float* labels, const uint8_t* mask,
                          const float* uniform, uint64_t seed, uint64_t offset, int B, int S,
                          int L, float gumbel_temperature, float* sampled_out, void* stream);
int tfr_gumbel_sample_bwd_f32(const float* sampled, const float* labels, const uint8_t* mask,
                              const float* upstream, int B, int S, int L,
                              float gumbel_temperature, float* dlogits_out, void* stream);
/* The same two for a training step that is REPLAYED from a hipGraph (lists the one-wavefront-per-list kernels take, up to a thousand items; TFR_EINVAL beyond): the Philox offset of the draw is
 * `offset + *step` with `step` a uint64 in device memory, and the backward advances it (*step_inc += 1) -- every replay
 * draws new noise (a host-side offset is frozen into the graph).  `labels_out` (nullable, [B * S, L]): the labels of the
 * S copies of every list, written by the same launch (what the loss of the sampled lists consumes). */
int tfr_gumbel_sample_step_f32(const float* logits, const float* labels, const uint8_t* mask,
                               const float* uniform, uint64_t seed, uint64_t offset, const uint64_t* step,
                               int B, int S, int L, float gumbel_temperature, float* sampled_out,
                               float* labels_out, void* stream);
int tfr_gumbel_sample_bwd_step_f32(const float* sampled, const float* labels, const uint8_t* mask,
                                   const float* upstream, int B, int S, int L, float gumbel_temperature,
                                   float* dlogits_out, uint64_t* step_inc, void* stream);

/* ---------------------------------------------------------------------------------------
 * Scorer tower: tfr.keras.layers.create_tower (keras/layers.py:26-77) on the flattened
 * [M = B*L, F] matrix of DNNScorer._score_flattened (keras/model.py:800-817); the groupwise
 * scorer's group_score_fn (model.py:276-306) is the same tower on [B*G, group_size*F].
 * bf16 operands / fp32 accumulation on the MFMA units; "bf16" pointers are uint16 bit
 * patterns, row-major with the stated pitch (elements).  prologue: 0 = A as is,
 * 1 = A*scale[k] + shift[k], 2 = relu(A*scale[k] + shift[k]) -- i.e. the BatchNormalization +
 * Activation of the layer below applied while its pre-activation is loaded.
 * epilogue: 0 = plain, 1 = + per-column partial sums (sum z, sum z^2) per 64-row half tile into
 * stats[tfr_tower_gemm_stats_rows(M)][2][N] (the next BatchNormalization's batch statistics),
 * 2 = backward of relu/BN-input: C = acc * 1[Zp*e_scale + e_shift > 0], stats = partial
 * (sum dy, sum dy * zhat) with zhat = (Zp - e_mean) * e_rstd.
 * Activations other than ReLU (keras/layers.py:66-70 takes any Keras activation): prologue 3 | act << 8 =
 * act(A*scale[k] + shift[k]), epilogue 3 | act << 8 = C = acc * act'(Zp*e_scale + e_shift) with the same stats;
 * act: 1 tanh, 2 sigmoid, 3 elu (alpha = 1), 4 softplus, 5 swish.  (Prologue 3 with epilogues 2 / 3, and
 * epilogue 3 with a prologue, are TFR_EINVAL: no tower runs them.)                              */

/* Dropout after a hidden activation (keras/layers.py:72-73).  HOST struct, nullable everywhere
 * (NULL or threshold16 == 0: no dropout).  Element (row m, column k) of the layer is kept iff its
 * field of hash(seed', m, k / columns-per-word) >= the rate in field units (the narrowest field --
 * 1, 2, 4, 8 or 16 bits -- that holds threshold16 = rate * 65536 exactly; with 16-bit fields the
 * scale is 65536 / (65536 - threshold16) whatever `scale` says), and then scaled by `scale` =
 * 1 / (1 - rate); forward prologues and backward
 * kernels evaluate the same hash.  seed' = seed + *step * 0x9E3779B9 when `step` (DEVICE pointer to
 * one uint32, nullable) is given: the training-step counter lives in device memory so that a
 * replayed hipGraph of the step draws a fresh mask (a by-value seed is baked into the capture). */
typedef struct tfr_tower_dropout {
  uint32_t seed;
  uint32_t threshold16;
  float scale;
  const uint32_t* step;
} tfr_tower_dropout;

/* Dense input cast: fp32 x[M, F] (pitch ldx) -> bf16 out[M, Kp], Kp = F rounded up to 8, zero
 * padded; optional per-column affine (an input BatchNormalization folded in). */
int tfr_tower_cast_f32_bf16(const float* x, long ldx, int M, int F, int Kp, const float* scale,
                            const float* shift, void* out_bf16, void* stream);
/* dst[j][0..n[j]) += src[j][0..n[j]) for j < count (host arrays of device pointers): gradient accumulation of the
 * tower's small vectors (biases, gamma, beta, output weights) in one launch per 16 vectors. */
int tfr_tower_multi_add(float* const* dst, const float* const* src, const int* n, int count, void* stream);

/* FlattenList's gather index (keras/layers.py:122-183; utils.py:203-230, :308-356 with shuffle=False) in one
 * launch: rows[b * L + p] = b * L + v_b[p mod max(n_b, 1)], v_b = valid positions of list b in index order
 * (0 when the list has none).  mask uint8 [B, L]; rows int32 [B * L]; list_size <= 8192 (TFR_MAX_LIST_SIZE). */
int tfr_flatten_row_index(const unsigned char* mask, int B, int L, int* rows, void* stream);

/* The same with a row gather: out[m] = cast(x[row_index[m]]) (row_index NULL = identity).  Fuses FlattenList's
 * circular padding (keras/layers.py:126-182: padded slots re-use the list's valid items) into the cast. */
int tfr_tower_cast_gather_f32_bf16(const float* x, long ldx, int M, int F, int Kp, const float* scale,
                                   const float* shift, const int* row_index, void* out_bf16, void* stream);
/* create_tower(input_batch_norm=True) (keras/layers.py:57-60): per-column partial sums of the raw fp32 features,
 * partial[n_blocks][2][F] = (sum x, sum x^2) over the block's rows (rows gathered through row_index like the cast);
 * tfr_tower_bn_finalize turns them into the scale / shift the cast applies.  `pivot` [F] (nullable): the sums are taken
 * of x - pivot (a sample of each column, e.g. the first row): mean = pivot + finalize's mean, the variance is unchanged
 * and keeps its digits when |mean| >> std. */
int tfr_tower_input_stats_f32(const float* x, long ldx, int M, int F, const int* row_index, float* partial,
                              int n_blocks, const float* pivot, void* stream);
/* bf16 feature ingest (the host parser's tfr_io_parse_elwc_batch_bf16; no reference counterpart -- the reference feeds
 * fp32 tensors to a Keras model whose Dense layers run at whatever precision the policy says, keras/layers.py:26-77):
 * the two entry points above for features that arrive as bfloat16 x_bf16[R, F] (pitch ldx in ELEMENTS).  Same row
 * gather, zero padding to Kp, affine and statistics, every element widened exactly; without an affine the values
 * pass through bit for bit, i.e. parse-to-bf16 + this gather == parse-to-fp32 + tfr_tower_cast_gather_f32_bf16. */
int tfr_tower_cast_gather_bf16_bf16(const void* x_bf16, long ldx, int M, int F, int Kp, const float* scale,
                                    const float* shift, const int* row_index, void* out_bf16, void* stream);
int tfr_tower_input_stats_bf16(const void* x_bf16, long ldx, int M, int F, const int* row_index, float* partial,
                               int n_blocks, const float* pivot, void* stream);
/* fp32 w[R, C] -> bf16 [R, pitch] or (transpose) bf16 [C, pitch]: the per-step operand copy of a
 * Dense kernel (fp32 master weights stay with the optimizer). */
int tfr_tower_weight_cast(const float* w, int R, int C, int transpose, int pitch, void* out_bf16,
                          void* stream);
/* `count` such casts in one launch per 8 matrices (host arrays of device pointers / shapes): all the weight casts
 * of a training step -- forward operands and transposed dgrad operands -- together. */
int tfr_tower_weight_cast_batch(const float* const* w, const int* R, const int* C, const int* transpose,
                                const int* pitch, void* const* out_bf16, int count, void* stream);
/* The same launch also advances the Dropout step counter (tfr_tower_dropout.step): *step += 1, *step_copy = the new
 * value -- one-element int32 device tensors (both or neither); a captured step draws new masks at every replay. */
int tfr_tower_weight_cast_batch_step(const float* const* w, const int* R, const int* C, const int* transpose,
                                     const int* pitch, void* const* out_bf16, int count, int32_t* step,
                                     int32_t* step_copy, void* stream);
/* Dense (+ fused neighbours): C[M, N] = prologue(A)[M, K] . B[N, K]^T + bias, bf16 out. */
int tfr_tower_gemm_bf16(const void* A, long lda, const void* B, long ldb, void* C, long ldc,
                        int M, int N, int K, int prologue, const float* a_scale, const float* a_shift,
                        const float* bias, int epilogue, float* stats, const void* Zp, long ldz,
                        const float* e_scale, const float* e_shift, const float* e_mean,
                        const float* e_rstd, const tfr_tower_dropout* pro_dropout,
                        const tfr_tower_dropout* epi_dropout, void* stream);
/* The same, and the kernel also writes the operand it forms in registers: a_out[M, K] (pitch ldao, bf16) =
 * prologue(A) -- activation(BatchNorm(z)) times the Dropout keep mask of the layer below -- for the weight gradient of
 * THIS layer (dW = dz^T . prologue(A)) to read back without a prologue of its own.  Only where the persistent
 * 256 x 256 kernel runs over full tiles: tfr_tower_gemm_writes_operand(M, N, K) != 0, a prologue, epilogue 0 / 1;
 * TFR_EINVAL otherwise.  a_out NULL = tfr_tower_gemm_bf16. */
int tfr_tower_gemm_writes_operand(int M, int N, int K);
int tfr_tower_gemm_bf16_aout(const void* A, long lda, const void* B, long ldb, void* C, long ldc,
                             int M, int N, int K, int prologue, const float* a_scale, const float* a_shift,
                             const float* bias, int epilogue, float* stats, const void* Zp, long ldz,
                             const float* e_scale, const float* e_shift, const float* e_mean,
                             const float* e_rstd, const tfr_tower_dropout* pro_dropout,
                             const tfr_tower_dropout* epi_dropout, void* a_out, long ldao, void* stream);
/* Dense in the REFERENCE's precision (keras/layers.py:26-77 builds fp32 Dense layers; model.py:755-817 trains through
 * them): C[M, N] = op(A)[M, K] . op(B)[K, N] (+ bias[N]) with fp32 operands, fp32 accumulation on the matrix cores
 * (v_mfma_f32_32x32x2_f32: a k-ordered fp32 fma chain).  Any M, N, K >= 0 and any pitches.
 *   a_k_contiguous != 0: A(m, k) = A[m * lda + k], else A(m, k) = A[k * lda + m]
 *   b_k_contiguous != 0: B(k, n) = B[n * ldb + k], else B(k, n) = B[k * ldb + n]
 * so that   y  = x . W^T + b : (x, 1, W, 1)      dx = dy . W : (dy, 1, W, 0)      dW = dy^T . x : (dy, 0, x, 0).
 *   splits > 1 cuts the contraction into that many slabs (the weight gradient contracts over batch_size * list_size
 *   rows): `workspace` = splits * M * N floats, summed in ascending order by a second launch (same bits every run);
 *   tfr_tower_gemm_f32_splits(M, N, K) proposes a count. */
int tfr_tower_gemm_f32(const float* A, long lda, int a_k_contiguous, const float* B, long ldb, int b_k_contiguous,
                       float* C, long ldc, int M, int N, int K, const float* bias, int splits, float* workspace,
                       void* stream);
int tfr_tower_gemm_f32_splits(int M, int N, int K);
/* Bias gradient of that Dense layer: out[n] = sum_m X[m * ldx + n], two deterministic stages;
 * `partial` = tfr_tower_colsum_rows(M) * N floats of scratch. */
int tfr_tower_colsum_f32(const float* X, long ldx, int M, int N, float* partial, float* out, void* stream);
int tfr_tower_colsum_rows(int M);
int tfr_tower_gemm_stats_rows(int M);            /* rows of `stats` for a given M            */
int tfr_tower_reduce_scratch_rows(int T);     /* rows of the `scratch` buffers below      */
/* BatchNormalization (training): partial[T][2][N] -> mean / biased variance -> scale = gamma *
 * rsqrt(var + eps), shift = beta - mean * scale; moving averages updated in place
 * (moving * momentum + batch * (1 - momentum)).  scratch: [scratch_rows(T)][2N] or NULL. */
int tfr_tower_bn_finalize(const float* partial, int T, int N, long M, const float* gamma,
                          const float* beta, float eps, float momentum, float* moving_mean,
                          float* moving_var, float* scale, float* shift, float* mean_out,
                          float* rstd_out, float* scratch, void* stream);
/* BatchNormalization backward coefficients: pqr[3][N] with dz = p * dy + q * z + r, from c[2][N] = (sum dy,
 * sum dy * zhat) = (d beta, d gamma), the forward's mean / rstd and gamma; M = rows of the batch. */
int tfr_tower_bn_bwd_coeffs(const float* gamma, const float* rstd, const float* mean, const float* c,
                            int N, long M, float* pqr, void* stream);
/* out[i] = sum_t partial[t][i], i < W.  scratch: [scratch_rows(T)][W] or NULL. */
int tfr_tower_reduce_partials(const float* partial, int T, int W, float* out, float* scratch,
                              void* stream);
/* The same for partial[T][J][N] (J stacked rows) and, when gamma is given, the coefficients of tfr_tower_bn_bwd_coeffs
 * from rows 0 / 1 of the result in the same launch (one launch for T <= 1024, J <= 6; else the two entry points in turn). */
int tfr_tower_reduce_partials_coeffs(const float* partial, int T, int J, int N, float* out, float* scratch,
                                     const float* gamma, const float* rstd, const float* mean, long M,
                                     float* pqr, void* stream);
/* The same launch also adds up the columns of dl[Mr][O] (O <= 4: the output layer's bias gradient) into db[O] -- only in
 * the one-launch form (tfr_tower_reduce_partials_serves_db(T, J) == 1); dl = NULL: exactly the call above. */
int tfr_tower_reduce_partials_serves_db(int T, int J);
int tfr_tower_reduce_partials_coeffs_db(const float* partial, int T, int J, int N, float* out, float* scratch,
                                        const float* gamma, const float* rstd, const float* mean, long M,
                                        float* pqr, const float* dl, long Mr, int O, float* db, void* stream);
/* Output Dense(output_units <= 4): out[M, O] = prologue(z)[M, K] . w[O, K]^T + b (fp32). */
int tfr_tower_out_f32(const void* z, long ldz, int M, int K, int prologue, const float* scale,
                      const float* shift, const float* w, const float* b, int O, float* out,
                      const tfr_tower_dropout* dropout, void* stream);
/* Its backward: dy[M, K] (bf16) = (dlogits . w) * relu mask; partial[n_blocks][2 + O][K] =
 * per-block (sum dy, sum dy * zhat, d w[o, :]). */
int tfr_tower_out_bwd(const void* z, long ldz, int M, int K, int prologue, const float* scale,
                      const float* shift, const float* mean, const float* rstd, const float* w,
                      const float* dlogits, int O, void* dy_bf16, long lddy, float* partial,
                      int n_blocks, const tfr_tower_dropout* dropout, void* stream);
/* The same in two passes around the BatchNorm-backward coefficients, so that the [M, K] gradient is written once:
 *   pass 1: dy_bf16 = NULL           -> only `partial` (sum dy, sum dy * zhat, dW_out) ;
 *   pass 2: partial = NULL, pqr[3][K] -> dy_bf16 receives dz = p * bf16(dy) + q * z + r (tfr_tower_bn_bwd_apply fused in).
 * pqr = NULL and both outputs given = tfr_tower_out_bwd. */
int tfr_tower_out_bwd2(const void* z, long ldz, int M, int K, int prologue, const float* scale,
                       const float* shift, const float* mean, const float* rstd, const float* w,
                       const float* dlogits, int O, void* dy_bf16, long lddy, float* partial,
                       int n_blocks, const tfr_tower_dropout* dropout, const float* pqr, void* stream);
/* BatchNormalization backward, in place: dy <- p[k]*dy + q[k]*z + r[k]; pqr is fp32 [3][K]. */
int tfr_tower_bn_bwd_apply(void* dy_bf16, long lddy, const void* z, long ldz, int M, int K,
                           const float* pqr, void* stream);
/* Dense kernel gradient: slab[s][N][ldw] = partial dz[M, N]^T . prologue(A)[M, K] over the
 * s-th slice of M (fp32); tfr_tower_slab_reduce sums the slices. */
int tfr_tower_wgrad_bf16(const void* DZ, long lddz, const void* A, long lda, int M, int N, int K,
                         int prologue, const float* a_scale, const float* a_shift, float* slab,
                         long ldw, int splits, const tfr_tower_dropout* dropout, void* stream);
int tfr_tower_slab_reduce(const float* slab, int S, long n, float* out, int accumulate, void* stream);
/* The same for slabs [S][R][Cs] whose rows are wider than the result's: out[R][Cout] (+)= sum_s slab[s][:, :Cout]. */
int tfr_tower_slab_reduce_cols(const float* slab, int S, int R, int Cs, int Cout, float* out, int accumulate,
                               void* stream);

/* out[0] = sum_i x[i] * w[i] (w NULL: sum_i x[i]), n <= 65536, 16-byte aligned inputs: the scalar reduction of a per-list
 * loss vector (compute_weighted_loss / the Keras reduction, losses_impl.py:787-814) in one launch with a fixed
 * summation order. */
int tfr_list_dot_f32(const float* x, const float* w, int n, float* out, void* stream);

/* ---- groupwise multi-item scoring (model.py:164-244, 313-421; csrc/groupwise.hip) -------------------------------
 * Group indices of _form_group_indices_nd (model.py:205-244) for one shuffle: the valid items of every list in
 * organize_valid_indices order (utils.py:203-230: index order when `keys` is NULL, else by DESCENDING key, ties by
 * index -- the caller draws keys ~ U[0,1)), then the rolling windows of _rolling_window_indices (model.py:164-202).
 *   is_valid   [B, L] uint8;  keys nullable [B, L] fp32
 *   idx_out    [B, L, group_size] int32: item index (inside its list) of member k of group g
 *   gmask_out  [B, L] uint8: group g is valid (its first member is one of the n valid items) */
int tfr_group_indices_i32(const uint8_t* is_valid, const float* keys, int B, int L, int group_size,
                          int32_t* idx_out, uint8_t* gmask_out, void* stream);
/* tf.gather_nd(example_features, feature_gather_indices) + reshape + the tower's bf16 input cast (model.py:374-381):
 *   x   fp32 [B, L, F] with row pitch ldx (elements);  idx [B, G, group_size] int32
 *   out bf16 [B * G, Kp], Kp >= group_size * F a multiple of 8; out[(b, g), k * F + f] = x[b, idx[b, g, k], f],
 *   padding columns are zero. */
int tfr_group_gather_cast_f32_bf16(const float* x, long ldx, const int32_t* idx, int B, int L, int G,
                                   int group_size, int F, int Kp, void* out_bf16, void* stream);
/* The two tf.scatter_nd + div_no_nan of model.py:389-409: logits[b, i] = (sum of scores[b, g, k] over valid groups
 * with idx[b, g, k] == i) / (their number), 0 where none lands.  Contributions are added in (g, k) order.
 *   scores [B * G, group_size] fp32;  gmask [B, G] uint8;  logits_out [B, L];  counts_out nullable [B, L]. */
int tfr_group_scatter_avg_f32(const float* scores, const int32_t* idx, const uint8_t* gmask, int B, int L,
                              int G, int group_size, float* logits_out, float* counts_out, void* stream);
/* Its backward: dscores[b, g, k] = gmask[b, g] ? dlogits[b, idx] / counts[b, idx] : 0  (0 where count is 0). */
int tfr_group_scatter_avg_bwd_f32(const float* dlogits, const float* counts, const int32_t* idx,
                                  const uint8_t* gmask, int B, int L, int G, int group_size,
                                  float* dscores_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif  /* TFR_HIP_H_ */
