"""Pins the CPU oracle at the KERAS level against the reference's own known-answer tests
(python/keras/losses_test.py, python/keras/metrics_test.py): the expectations are the closed
forms those tests state (sums over preference pairs, softmax picks, DCG ratios), re-derived here in plain
Python -- TF is not installable, so these are the executable link between `oracle/tfr_ref.py` and the
Keras `__call__` / `update_state` + `result` semantics (weight broadcasting, AUTO / SUM reductions,
per-list metric weights).
"""
import math

import pytest
import torch

from oracle import tfr_ref as R

ln = math.log
RED = R.Reduction
T = lambda x: torch.tensor(x, dtype=torch.float32)


def near(a, b, tol=1e-5):
    a = float(torch.as_tensor(a).reshape(()))
    assert abs(a - b) <= tol * max(1.0, abs(b)), (a, b)


# ---------------------------------------------------------------------------- pairwise (keras/losses_test.py:146-243)
_FORMS = {
    'hinge': (R.PairwiseHingeLoss, lambda d: max(0.0, 1.0 - d)),
    'logistic': (R.PairwiseLogisticLoss, lambda d: ln(1.0 + math.exp(-d))),
    'soft_zero_one': (R.PairwiseSoftZeroOneLoss, lambda d: 1.0 / (1.0 + math.exp(d))),
}


def _pair_sum(labels, scores, weights, phi, log_discount=False):
    """(sum over pairs label_i > label_j of w_i * phi(s_i - s_j) * delta_ij, number of items): delta = 1, or
    |l_i - l_j| * |1/ln(1+r_i) - 1/ln(1+r_j)| with r = 1-based rank by descending score (smooth_fraction = 1)."""
    order = sorted(range(len(scores)), key=lambda i: -scores[i])
    rank = {i: r + 1 for r, i in enumerate(order)}
    total = 0.0
    for i in range(len(labels)):
        for j in range(len(labels)):
            if labels[i] > labels[j]:
                delta = 1.0
                if log_discount:
                    delta = abs(labels[i] - labels[j]) * abs(1.0 / ln(1.0 + rank[i]) - 1.0 / ln(1.0 + rank[j]))
                total += weights[i] * phi(scores[i] - scores[j]) * delta
    return total, float(len(labels))


@pytest.mark.parametrize('form', sorted(_FORMS))
def test_keras_pairwise_losses(form):
    ctor, phi = _FORMS[form]
    scores = [[1., 3., 2.], [1., 2., 3.]]
    labels = [[0., 0., 1.], [0., 0., 2.]]
    list_w, item_w = [[2.], [1.]], [[2., 3., 4.], [1., 1., 1.]]
    ones = [1.] * 3
    agg = lambda parts: sum(p[0] for p in parts) / sum(p[1] for p in parts)
    for b in (0, 1):                                                   # individual lists, then item weights
        near(R.keras_loss_call(ctor(), T([labels[b]]), T([scores[b]])), agg([_pair_sum(labels[b], scores[b], ones, phi)]))
        near(R.keras_loss_call(ctor(), T([labels[b]]), T([scores[b]]), T([item_w[b]])),
             agg([_pair_sum(labels[b], scores[b], item_w[b], phi)]))
    near(R.keras_loss_call(ctor(), T(labels), T(scores), T(list_w)),   # per-list weights
         agg([_pair_sum(labels[b], scores[b], [list_w[b][0]] * 3, phi) for b in (0, 1)]))
    lam = R.DCGLambdaWeight(rank_discount_fn=lambda r: 1. / torch.log1p(r), smooth_fraction=1.)
    near(R.keras_loss_call(ctor(lambda_weight=lam), T(labels), T(scores), T(list_w)),
         agg([_pair_sum(labels[b], scores[b], [list_w[b][0]] * 3, phi, log_discount=True) for b in (0, 1)]) * 3.)


def test_keras_pairwise_logistic_invalid_labels_sum_and_temperature():  # keras/losses_test.py:710-733
    yt, yp = T([[0., -1., 1.]]), T([[1., 3., 2.]])
    near(R.keras_loss_call(R.PairwiseLogisticLoss(), yt, yp), ln(1 + math.exp(-1.)) / 3.)
    near(R.keras_loss_call(R.PairwiseLogisticLoss(), yt, yp, reduction=RED.KERAS_SUM), ln(1 + math.exp(-1.)))
    near(R.keras_loss_call(R.PairwiseLogisticLoss(temperature=0.1), yt, yp, reduction=RED.KERAS_SUM),
         ln(1 + math.exp(-10.)))


# ---------------------------------------------------------------------------- softmax (keras/losses_test.py:284-308, 735-741)
def _pick(scores, k):
    z = sum(math.exp(s) for s in scores)
    return math.exp(scores[k]) / z


def test_keras_softmax_loss():
    scores = [[1., 3., 2.], [1., 2., 3.], [1., 2., 3.]]
    labels = [[0., 0., 1.], [0., 0., 2.], [0., 0., 0.]]
    near(R.keras_loss_call(R.SoftmaxLoss(), T(labels), T(scores)),
         -(ln(_pick(scores[0], 2)) + ln(_pick(scores[1], 2)) * 2.) / 3.)
    near(R.keras_loss_call(R.SoftmaxLoss(), T(labels), T(scores), T([[2.], [1.], [1.]])),
         -(ln(_pick(scores[0], 2)) * 2. + ln(_pick(scores[1], 2)) * 2. * 1.) / 3.)
    lam = R.DCGLambdaWeight(rank_discount_fn=lambda r: 1. / torch.log1p(r))
    near(R.keras_loss_call(R.SoftmaxLoss(lambda_weight=lam), T(labels), T(scores)),
         -(ln(_pick(scores[0], 2)) / ln(1. + 2.) + ln(_pick(scores[1], 2)) * 2. / ln(1. + 1.)) / 3.)
    # an invalid (-1) label drops the item from the softmax
    near(R.keras_loss_call(R.SoftmaxLoss(), T([[0., -1., 1.]]), T([[1., 3., 2.]])), -ln(_pick([1., 2.], 1)))


def _calibrated_expected(scores, labels, list_w, virtual_label):
    """keras/losses_test.py:936-970 closed form: softmax with the denominator shifted by e^0 = 1 plus the virtual term."""
    den = lambda v: sum(math.exp(x) for x in v)
    tot = 0.0
    for sc, lb, w in zip(scores, labels, list_w):
        for k, y in enumerate(lb):
            if y:
                tot += w * y * ln(math.exp(sc[k]) / (den(sc) + 1.0))
        tot -= w * virtual_label * ln(1.0 + den(sc))
    return -tot / len(scores)


def test_keras_calibrated_softmax_loss():                                # keras/losses_test.py:936-970
    scores = [[1.0, 3.0, 2.0], [1.0, 2.0, 3.0], [1.0, 2.0, 3.0]]
    labels = [[0.0, 0.0, 1.0], [0.0, 0.0, 2.0], [0.0, 0.0, 0.0]]
    near(R.keras_calibrated_softmax_call(T(labels), T(scores), virtual_label=0.5),
         _calibrated_expected(scores, labels, [1., 1., 1.], 0.5))
    near(R.keras_calibrated_softmax_call(T(labels), T(scores), T([[2.0], [1.0], [1.0]]), virtual_label=0.5),
         _calibrated_expected(scores, labels, [2., 1., 1.], 0.5))
    # docstring value, keras/losses.py:850-854
    near(R.keras_calibrated_softmax_call(T([[1., 0.]]), T([[0.6, 0.8]]), virtual_label=0.1), 1.1808171)


# ---------------------------------------------------------------------------- ApproxNDCG (keras/losses_test.py:576-602, 650-693)
def _norm_weight(w, l):
    return sum(wi * li for wi, li in zip(w, l)) / sum(l) if sum(l) > 0 else 0.0


@pytest.mark.parametrize('reduction,div', [(RED.AUTO, 3.), (RED.KERAS_SUM, 1.), (RED.SUM_OVER_BATCH_SIZE, 3.)])
def test_keras_approx_ndcg_loss(reduction, div):
    scores = [[1.4, -2.8, -0.4], [0., 1.8, 10.2], [1., 1.2, -3.2]]       # ranks [[1,3,2],[3,2,1],[2,1,3]]
    labels = [[0., 2., 1.], [1., 0., 3.], [0., 0., 0.]]
    item_w = [[1., 2., 3.], [4., 5., 6.], [7., 8., 9.]]
    n0 = (1 / (3 / ln(2) + 1 / ln(3))) * (3 / ln(4) + 1 / ln(3))
    n1 = (1 / (7 / ln(2) + 1 / ln(3))) * (7 / ln(2) + 1 / ln(4))
    call = lambda w=None: R.keras_loss_call(R.ApproxNDCGLoss(temperature=0.1), T(labels), T(scores),
                                            None if w is None else T(w), reduction=reduction)
    near(call(), -(n0 + n1) / div)
    near(call([[2.], [1.], [1.]]), -(2 * n0 + n1) / div)
    nw = [_norm_weight(w, l) for w, l in zip(item_w, labels)]
    near(call(item_w), -(nw[0] * n0 + nw[1] * n1) / div)


# ---------------------------------------------------------------------------- metrics (keras/metrics_test.py:296-396, 856-992)
def _dcg(label, rank, weight=1.0):
    return weight * (2.0 ** label - 1.0) / math.log2(1.0 + rank)


def _ndcg_list_weights(weights, labels):
    """metrics_impl.py:63-119 for NDCG: per list sum(w * gain-free relevance) / sum(relevance); lists with no
    relevant item take the mean weight of the others."""
    out = []
    for w, l in zip(weights, labels):
        rel = [2.0 ** x - 1.0 for x in l]
        out.append(sum(wi * ri for wi, ri in zip(w, rel)) / sum(rel) if sum(rel) > 0 else None)
    have = [x for x in out if x is not None]
    fill = sum(have) / len(have) if have else 0.0
    return [fill if x is None else x for x in out]


def test_keras_ndcg_metric():
    scores = [[1., 3., 2.], [1., 2., 3.]]                               # ranks [[3,1,2],[3,2,1]]
    labels = [[0., 0., 1.], [0., 1., 2.]]
    weights = [[1., 2., 3.], [4., 5., 6.]]
    m = lambda metric, yt, yp, w=None: R.keras_metric_mean(metric, [(yt, yp, w)])
    n1 = (_dcg(0., 1) + _dcg(1., 2) + _dcg(0., 3)) / (_dcg(1., 1) + _dcg(0., 2) + _dcg(0., 3))
    near(m(R.NDCGMetric(), [labels[0]], [scores[0]]), n1)
    near(m(R.NDCGMetric(), labels, scores), (n1 + 1.0) / 2.0)
    near(m(R.NDCGMetric(), [[0., 0., 0.], [0., 1., 2.]], scores), 0.5)            # zero relevance list counts as 0
    # per-item weights
    near(m(R.NDCGMetric(topn=1), [labels[0]], [scores[0]], [weights[0]]), _dcg(0., 1, 2.) / _dcg(1., 1, 3.))
    w1 = (_dcg(0., 1, 2.) + _dcg(1., 2, 3.) + _dcg(0., 3, 1.)) / (_dcg(1., 1, 3.) + _dcg(0., 2, 1.) + _dcg(0., 3, 2.))
    near(m(R.NDCGMetric(), [labels[0]], [scores[0]], [weights[0]]), w1)
    lw = _ndcg_list_weights(weights, labels)
    near(m(R.NDCGMetric(), labels, scores, weights), (w1 * lw[0] + 1.0 * lw[1]) / sum(lw))
    t1 = _dcg(0., 1, 2.) / _dcg(1., 1, 3.)
    near(m(R.NDCGMetric(topn=1), labels, scores, weights), (t1 * lw[0] + 1.0 * lw[1]) / sum(lw))
    near(m(R.NDCGMetric(), labels, scores, [[1.], [2.]]), (n1 + 2.0) / 3.0)       # per-list weights
    near(m(R.NDCGMetric(), labels, scores, [[0.], [0.]]), 0.0)
    near(m(R.NDCGMetric(topn=1), [[0., 0., 0.]], [scores[0]], [weights[0]]), 0.0)
    # zero-relevance list with item weights: it takes the mean of the other lists' weights (5.75 each)
    z = [[0., 0., 0.], [0., 1., 2.]]
    lwz = _ndcg_list_weights(weights, z)
    assert abs(lwz[0] - 5.75) < 1e-6 and abs(lwz[1] - 5.75) < 1e-6
    near(m(R.NDCGMetric(), z, scores, weights), (0.0 * lwz[0] + 1.0 * lwz[1]) / sum(lwz))


def test_keras_mrr_metric():
    scores = [[1., 3., 2.], [1., 2., 3.], [3., 1., 2.]]                 # ranks [[3,1,2],[3,2,1],[1,3,2]]
    labels = [[0., 0., 1.], [0., 1., 2.], [0., 1., 0.]]
    weights = [[1., 2., 3.], [4., 5., 6.], [7., 8., 9.]]
    rel_rank = [2, 1, 3]
    mean_rel_w = [weights[0][2], sum(weights[1][1:]) / 2, weights[2][1]]
    m = lambda metric, yt, yp, w=None: R.keras_metric_mean(metric, [(yt, yp, w)])
    near(m(R.MRRMetric(), [labels[0]], [scores[0]]), 1. / rel_rank[0])
    near(m(R.MRRMetric(topn=1), [labels[0]], [scores[0]]), 0.)
    near(m(R.MRRMetric(topn=2), [labels[0]], [scores[0]]), 1. / rel_rank[0])
    near(m(R.MRRMetric(), [labels[1]], [scores[1]]), 1. / rel_rank[1])
    near(m(R.MRRMetric(topn=1), [labels[1]], [scores[1]]), 1. / rel_rank[1])
    near(m(R.MRRMetric(topn=6), [labels[1]], [scores[1]]), 1. / rel_rank[1])
    near(m(R.MRRMetric(), [labels[2]], [scores[2]]), 1. / rel_rank[2])
    near(m(R.MRRMetric(topn=2), [labels[2]], [scores[2]]), 0.)
    near(m(R.MRRMetric(topn=3), [labels[2]], [scores[2]]), 1. / rel_rank[2])
    near(m(R.MRRMetric(), labels[:2], scores[:2]), (0.5 + 1.0) / 2)
    near(m(R.MRRMetric(), labels[:2], scores[:2], weights[:2]), (3. * 0.5 + (6. + 5.) / 2. * 1.) / (3. + (6. + 5) / 2.))
    near(m(R.MRRMetric(), labels, scores), sum(1. / r for r in rel_rank) / 3.)
    near(m(R.MRRMetric(topn=2), labels, scores), (1. / rel_rank[0] + 1. / rel_rank[1]) / 3.)
    near(m(R.MRRMetric(topn=1), labels, scores, weights), (mean_rel_w[1] / rel_rank[1]) / sum(mean_rel_w))
    near(m(R.MRRMetric(topn=1), labels, scores), (0. + 1. + 0.) / 3.)
    near(m(R.MRRMetric(), labels, scores, weights),
         sum(w / r for w, r in zip(mean_rel_w, rel_rank)) / sum(mean_rel_w))


# ---------------------------------------------------------------------------- the other Keras metrics
# (keras/metrics_test.py:413-1354: Hits, ARP, Precision, MAP, DCG, OPA, Recall -- dense and ragged, item / list / zero
# weights).  The Keras metric is the mean of the per-list metric under the per-list weights (keras/metrics.py:69-201);
# expectations are the reference tests' closed forms, evaluated by the helpers below.
def _mean(metric, yt, yp, w=None):
    return R.keras_metric_mean(metric, [(yt, yp, w)])


def _binary_list_weights(weights, labels):
    """Per-list weights of the binary-relevance metrics: sum(w * [label >= 1]) / #[label >= 1]; lists without a
    relevant item take the mean of the others (keras/metrics_test.py:121-160 `_example_weights_to_list_weights`)."""
    out = []
    for w, l in zip(weights, labels):
        rel = [1.0 if x >= 1.0 else 0.0 for x in l]
        out.append(sum(a * b for a, b in zip(w, rel)) / sum(rel) if sum(rel) > 0 else None)
    have = [x for x in out if x is not None]
    fill = sum(have) / len(have) if have else 0.0
    return [fill if x is None else x for x in out]


def _average_precision(rels, scores, topn=None):
    order = sorted(range(len(scores)), key=lambda i: -scores[i])[:topn or len(scores)]
    hits, total = 0.0, 0.0
    for k, i in enumerate(order, start=1):
        if rels[i]:
            hits += 1.0
            total += hits / k
    return total / sum(rels) if sum(rels) else 0.0


RAGGED_SCORES = [[1., 2., 0.], [1., 2.], [3., 4., 2., 1.]]
RAGGED_LABELS = [[1., 0., 0.], [0., 2.], [2., 0., 1., 0.]]
RAGGED_WEIGHTS = [[2., 1., 4.], [2., 1.], [2., 1., 4., 8.]]


def test_keras_hits_metric():  # keras/metrics_test.py:413-542
    scores = [[1., 3., 2.], [1., 2., 3.], [3., 1., 2.]]                 # ranks [[3,1,2],[3,2,1],[1,3,2]]
    labels = [[0., 0., 1.], [0., 1., 2.], [0., 1., 0.]]
    weights = [[1., 2., 3.], [4., 5., 6.], [7., 8., 9.]]
    mean_rel_w = [weights[0][2], sum(weights[1][1:]) / 2, weights[2][1]]
    for b, expect in ((0, {None: 1., 1: 0., 2: 1.}), (1, {None: 1., 1: 1., 2: 1.}), (2, {None: 1., 1: 0., 2: 0., 3: 1.})):
        for topn, e in expect.items():
            near(_mean(R.HitsMetric(topn=topn), [labels[b]], [scores[b]]), e)
    near(_mean(R.HitsMetric(topn=1), labels[:2], scores[:2]), 0.5)
    near(_mean(R.HitsMetric(topn=1), labels[:2], scores[:2], weights[:2]), ((6. + 5.) / 2.) / (3. + (6. + 5.) / 2.))
    for topn, e in ((1, 1. / 3.), (2, 2. / 3.), (3, 1.)):
        near(_mean(R.HitsMetric(topn=topn), labels, scores), e)
    near(_mean(R.HitsMetric(topn=1), labels, scores, weights), mean_rel_w[1] / sum(mean_rel_w))
    sc, lb, wt = [[1., 2., 0.], [1., 2.], [1., 4., 2., 3.]], RAGGED_LABELS, RAGGED_WEIGHTS    # ragged (:509-542)
    for topn, hits in ((1, [0., 1., 0.]), (2, [1., 1., 0.]), (3, [1., 1., 1.])):
        near(_mean(R.HitsMetric(topn=topn, ragged=True), lb, sc), sum(hits) / 3.)
        near(_mean(R.HitsMetric(topn=topn, ragged=True), lb, sc, wt), (hits[0] * 2. + hits[1] * 1. + hits[2] * 3.) / 6.)


def test_keras_arp_precision_recall_metrics():  # keras/metrics_test.py:544-660, 1294-1354
    scores = [[1., 3., 2.], [1., 2., 3.]]
    labels = [[0., 0., 1.], [0., 1., 2.]]
    weights = [[1., 2., 3.], [4., 5., 6.]]
    near(_mean(R.ARPMetric(), [labels[0]], [scores[0]]), 2.)
    near(_mean(R.ARPMetric(), labels, scores), (1. * 2. + 2. * 1. + 1. * 2.) / 4.)
    near(_mean(R.ARPMetric(), labels, scores, weights), (3. * 1. * 2. + 6. * 2. * 1. + 5 * 1. * 2.) / (3. + 12. + 5.))
    near(_mean(R.PrecisionMetric(), [labels[0]], [scores[0]]), 1. / 3.)
    near(_mean(R.PrecisionMetric(topn=1), [labels[0]], [scores[0]]), 0.)
    near(_mean(R.PrecisionMetric(), labels, scores), (1. / 3. + 2. / 3.) / 2.)
    zero = [[0., 0., 0.], [0., 1., 2.]]                                                       # :583-595
    near(_mean(R.PrecisionMetric(), [zero[0]], [scores[0]]), 0.)
    near(_mean(R.PrecisionMetric(), zero, scores), (0. + 2. / 3.) / 2.)
    lw = _binary_list_weights(weights, labels)                                                # :615-645
    near(_mean(R.PrecisionMetric(), labels, scores, weights), ((1. / 3.) * lw[0] + (2. / 3.) * lw[1]) / sum(lw))
    near(_mean(R.PrecisionMetric(topn=2), labels, scores, weights), ((1. / 2.) * lw[0] + 1. * lw[1]) / sum(lw))
    near(_mean(R.PrecisionMetric(), labels, scores, [[1.], [2.]]), ((1. / 3.) * 1. + (2. / 3.) * 2.) / 3.)
    near(_mean(R.PrecisionMetric(topn=2), labels, scores, [[0., 0., 0.], [0., 0., 0.]]), 0.)
    s3 = [[1., 3., 2.], [1., 3., 2.], [1., 2., 3.]]                                           # :597-613, 1327-1339
    l3 = [[0., 0., 0.], [0., 0., 1.], [0., 1., 2.]]
    w3 = [[0., 0., 1.], [1., 2., 3.], [4., 5., 6.]]
    lw3 = _binary_list_weights(w3, l3)
    assert lw3 == pytest.approx([(3 + 5.5) / 2., 3, 5.5])
    near(_mean(R.PrecisionMetric(topn=2), l3, s3, w3), (0. * lw3[0] + .5 * lw3[1] + 1. * lw3[2]) / sum(lw3))
    near(_mean(R.PrecisionMetric(), l3[:2], s3[:2], [[0., 0., 0.], [0., 0., 0.]]), 0.)
    near(_mean(R.RecallMetric(topn=1), l3, s3, w3), (0. * lw3[0] + 0. * lw3[1] + .5 * lw3[2]) / sum(lw3))
    near(_mean(R.RecallMetric(), [labels[0]], [scores[0]]), 1.)
    near(_mean(R.RecallMetric(topn=1), [labels[0]], [scores[0]]), 0.)
    near(_mean(R.RecallMetric(topn=2), labels, scores), 1.)
    near(_mean(R.RecallMetric(), [zero[0]], [scores[0]]), 0.)
    near(_mean(R.RecallMetric(), zero, scores), 0.5)
    near(_mean(R.PrecisionMetric(ragged=True), RAGGED_LABELS, RAGGED_SCORES), (1. / 3. + 1. / 2. + 2. / 4.) / 3.)        # :647-660
    near(_mean(R.PrecisionMetric(ragged=True), RAGGED_LABELS, RAGGED_SCORES, RAGGED_WEIGHTS), (2. / 3. + 1. / 2. + 3. * 2. / 4.) / 6.)
    near(_mean(R.RecallMetric(topn=2, ragged=True), RAGGED_LABELS, RAGGED_SCORES), (1. + 1. + 1. / 2.) / 3.)            # :1341-1354
    near(_mean(R.RecallMetric(topn=2, ragged=True), RAGGED_LABELS, RAGGED_SCORES, RAGGED_WEIGHTS), (2. + 1. + 3. / 2.) / 6.)


def test_keras_map_metric():  # keras/metrics_test.py:746-854
    scores = [[1., 3., 2.], [1., 2., 3.]]
    labels = [[0., 0., 1.], [0., 1., 2.]]
    rels = [[0, 0, 1], [0, 1, 1]]
    weights = [[1., 2., 3.], [4., 5., 6.]]
    M = R.MeanAveragePrecisionMetric
    for topn in (None, 1, 2):
        near(_mean(M(topn=topn), [labels[0]], [scores[0]]), _average_precision(rels[0], scores[0], topn))
    near(_mean(M(), labels, scores), sum(_average_precision(rels[i], scores[i]) for i in range(2)) / 2.)
    near(_mean(M(topn=1), labels, scores), sum(_average_precision(rels[i], scores[i], 1) for i in range(2)) / 2.)
    lw = _binary_list_weights(weights, labels)
    ap0 = ((1. / 2.) * 3.) / 3.
    ap1 = ((1. / 1.) * 6. + (2. / 2.) * 5.) / (5. + 6.)
    near(_mean(M(), [labels[0]], [scores[0]], [weights[0]]), ap0)
    near(_mean(M(), [labels[1]], [scores[1]], [weights[1]]), ap1)
    near(_mean(M(), labels, scores, weights), (ap0 * lw[0] + ap1 * lw[1]) / sum(lw))
    near(_mean(M(topn=1), labels, scores, weights), (0. * lw[0] + (6. / 11.) * lw[1]) / sum(lw))
    near(_mean(M(topn=2), labels, scores, weights), (ap0 * lw[0] + ap1 * lw[1]) / sum(lw))
    near(_mean(M(), labels, scores, [[1.], [2.]]), sum(_average_precision(rels[i], scores[i]) * (i + 1.) for i in range(2)) / 3.)
    near(_mean(M(ragged=True), RAGGED_LABELS, RAGGED_SCORES), (1. / 2. + 1. + (1. / 2. + 2. / 3.) / 2.) / 3.)
    near(_mean(M(ragged=True), RAGGED_LABELS, RAGGED_SCORES, RAGGED_WEIGHTS), (2. * 1. / 2. + 1. + 3. * (2. * 1. / 2. + 2. * 4. / 3.) / 6.) / 6.)


def test_keras_dcg_and_opa_metrics():  # keras/metrics_test.py:1015-1057, 1250-1292
    scores = [[1., 3., 2.], [1., 2., 3.]]
    labels = [[0., 0., 1.], [0., 1., 2.]]
    weights = [[1., 1., 1.], [2., 2., 1.]]
    d1 = _dcg(0., 1) + _dcg(1., 2) + _dcg(0., 3)
    d2 = _dcg(2., 1) + _dcg(1., 2)
    d2w = _dcg(2., 1) + _dcg(1., 2) * 2.
    w2 = ((4 - 1) * 1. + (2 - 1) * 2.) / (4 - 1 + 2 - 1)
    near(_mean(R.DCGMetric(), [labels[0]], [scores[0]]), d1)
    near(_mean(R.DCGMetric(), labels, scores), (d1 + d2) / 2.)
    near(_mean(R.DCGMetric(), labels, scores, weights), (d1 + d2w) / (1. + w2))
    near(_mean(R.DCGMetric(ragged=True), RAGGED_LABELS, RAGGED_SCORES), (_dcg(1., 2) + _dcg(2., 1) + _dcg(2., 2) + _dcg(1., 3)) / 3.)
    near(_mean(R.DCGMetric(ragged=True), RAGGED_LABELS, RAGGED_SCORES, RAGGED_WEIGHTS),
         (2. * _dcg(1., 2) + 1. * _dcg(2., 1) + 2. * _dcg(2., 2) + 4. * _dcg(1., 3)) / 5.5)
    labels = [[-1., 0., 1.], [0., 1., 2.]]
    near(_mean(R.OPAMetric(), [labels[0]], [scores[0]]), 0.)
    near(_mean(R.OPAMetric(), [labels[1]], [scores[1]]), 1.)
    near(_mean(R.OPAMetric(), labels, scores), 3. / 4.)
    near(_mean(R.OPAMetric(), labels, scores, [[1.], [2.]]), 6. / 7.)
    near(_mean(R.OPAMetric(), labels, scores, [[1., 1., 1.], [2., 2., 3.]]), (2. + 3. + 3.) / (1. + 2. + 3. + 3.))
    near(_mean(R.OPAMetric(ragged=True), RAGGED_LABELS, RAGGED_SCORES), (1. + 1. + 3.) / (2. + 1. + 5.))
    near(_mean(R.OPAMetric(ragged=True), RAGGED_LABELS, RAGGED_SCORES, RAGGED_WEIGHTS), (2. + 1. + 8.) / (4. + 1. + 14.))
