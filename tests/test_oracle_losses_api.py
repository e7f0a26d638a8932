"""The oracle's `make_loss_fn` (oracle/tfr_ref.py, restating python/losses.py:57-311) against the expectations of the
reference's python/losses_test.py -- the estimator-era factory the Keras wrappers sit beside: loss keys (single, list,
'key:weight' strings), `weights_feature_name` with 1-D / 2-D / 3-D weights, the caller's lambda weight, per-key
defaults (ApproxNDCG / ApproxMRR temperature 0.1), reductions, error messages, and the lambda-weight factories.

The expected values are the reference tests' closed forms, evaluated here by small pure-Python helpers written for this
file (pairs enumerated in score order, plain `math`).  Not transcribable: `test_make_yeti_logistic_loss` (:450-495) pins
two numbers on TF's random stream; `test_make_gumbel_approx_ndcg_fn` (:796-854) is reproduced through the sampled ranks
the reference test documents in its comments (noise injected so that those ranks come out).
"""
import math

import pytest
import torch

from oracle import tfr_ref as R

ln = math.log
SUM, MEAN = R.Reduction.SUM, R.Reduction.MEAN


def val(x):
    return float(torch.as_tensor(x).item())


def softmax(v):
    z = sum(math.exp(x) for x in v)
    return [math.exp(x) / z for x in v]


def pair_sums(labels, scores, weights, kind, log_discount=False):
    """(sum of w_i * delta_ij * loss_ij, number of pairs with a non-zero weight) over pairs label_i > label_j, items
    taken in descending score order; delta = 1, or |dl| * |1/ln(1 + r_i) - 1/ln(1 + r_j)| with the LOG rank discount."""
    order = sorted(range(len(scores)), key=lambda i: -scores[i])
    total, count = 0.0, 0
    for ri, i in enumerate(order):
        for rj, j in enumerate(order):
            if labels[i] <= labels[j]:
                continue
            delta = 1.0
            if log_discount:
                delta = abs(labels[i] - labels[j]) * abs(1. / ln(2. + ri) - 1. / ln(2. + rj))
            d = scores[i] - scores[j]
            term = {'pairwise_hinge_loss': max(0., 1. - d), 'pairwise_logistic_loss': ln(1. + math.exp(-d)),
                    'pairwise_soft_zero_one_loss': 1. / (1. + math.exp(d))}[kind]
            total += weights[i] * delta * term
            count += 1 if weights[i] * delta > 0 else 0
    return total, count


def aggregate(parts):
    return sum(p[0] for p in parts) / sum(p[1] for p in parts)


SCORES = [[1., 3., 2.], [1., 2., 3.]]
LABELS = [[0., 0., 1.], [0., 0., 2.]]


@pytest.mark.parametrize('key', ['pairwise_hinge_loss', 'pairwise_logistic_loss', 'pairwise_soft_zero_one_loss'])
def test_make_pairwise_losses(key):  # losses_test.py:302-408
    itemwise = [[2., 3., 4.], [1., 1., 1.]]
    listwise = [[2.], [1.]]
    ones = [1.] * 3
    fn = R.make_loss_fn(key)
    for b in range(2):
        assert val(fn([LABELS[b]], [SCORES[b]], {})) == pytest.approx(aggregate([pair_sums(LABELS[b], SCORES[b], ones, key)]), abs=1e-5)
    fn = R.make_loss_fn(key, weights_feature_name='weights')
    for b in range(2):
        got = fn([LABELS[b]], [SCORES[b]], {'weights': [itemwise[b]]})
        assert val(got) == pytest.approx(aggregate([pair_sums(LABELS[b], SCORES[b], itemwise[b], key)]), abs=1e-5)
    got = fn(LABELS, SCORES, {'weights': listwise})
    want = aggregate([pair_sums(LABELS[0], SCORES[0], [2.] * 3, key), pair_sums(LABELS[1], SCORES[1], ones, key)])
    assert val(got) == pytest.approx(want, abs=1e-5)
    # the caller's lambda weight: DCG with 1 / log1p(rank), fully smoothed -> |dl| * |dD|, times list_size (:366-387)
    lw = R.DCGLambdaWeight(rank_discount_fn=lambda r: 1. / torch.log1p(r), smooth_fraction=1.)
    got = R.make_loss_fn(key, weights_feature_name='weights', lambda_weight=lw)(LABELS, SCORES, {'weights': listwise})
    want = aggregate([pair_sums(LABELS[0], SCORES[0], [2.] * 3, key, True), pair_sums(LABELS[1], SCORES[1], ones, key, True)]) * 3.
    assert val(got) == pytest.approx(want, abs=1e-5)
    assert abs(val(R.make_loss_fn(key, reduction=SUM)(LABELS, SCORES, {})) - val(R.make_loss_fn(key, reduction=MEAN)(LABELS, SCORES, {}))) > 1e-3


def test_make_pairwise_mse_loss():  # losses_test.py:410-448
    feats = {'weights': [[1.], [2.]]}
    sq = lambda ds, dl: (ds - dl) ** 2
    first = sq(2. - 3., 1. - 0.) + sq(2. - 1., 1. - 0.) + sq(3. - 1., 0. - 0.)
    second = sq(3. - 2., 2. - 0.) + sq(3. - 1., 2. - 0.) + sq(2. - 1., 0. - 0.)
    assert val(R.make_loss_fn('pairwise_mse_loss', reduction=MEAN)(LABELS, SCORES, feats)) == pytest.approx((first + second) / 6., abs=1e-5)
    got = R.make_loss_fn('pairwise_mse_loss', reduction=MEAN, weights_feature_name='weights')(LABELS, SCORES, feats)
    assert val(got) == pytest.approx((first + 2. * second) / 9., abs=1e-5)


def test_make_circle_loss():  # losses_test.py:497-530 (gamma 64, margin 0.25)
    scores = [[0.1, 0.3, 0.2], [0.1, 0.2, 0.3]]
    labels = [[0., 0., 1.], [0., 1., 2.]]

    def circle(l, s, gamma=64., margin=0.25):
        tot = 0.
        for i in range(3):
            for j in range(3):
                if l[i] > l[j]:
                    tot += math.exp(gamma * max(0., (1. + margin) - s[i]) * ((1. - margin) - s[i])
                                    + gamma * max(0., s[j] + margin) * (s[j] - margin))
        return math.log1p(tot)
    c0, c1 = circle(labels[0], scores[0]), circle(labels[1], scores[1])
    feats = {'weights': [[2.], [1.]]}
    assert val(R.make_loss_fn('circle_loss')(labels, scores, feats)) == pytest.approx((c0 + c1) / 2., rel=1e-5)
    got = R.make_loss_fn('circle_loss', weights_feature_name='weights')(labels, scores, feats)
    assert val(got) == pytest.approx((2. * c0 + c1) / 2., rel=1e-5)


def test_make_softmax_family():  # losses_test.py:532-642
    feats = {'weights': [[2.], [1.]]}
    p0, p1 = softmax(SCORES[0])[2], softmax(SCORES[1])[2]
    assert val(R.make_loss_fn('softmax_loss')(LABELS, SCORES, feats)) == pytest.approx(-(ln(p0) + 2. * ln(p1)) / 2., abs=1e-5)
    got = R.make_loss_fn('softmax_loss', weights_feature_name='weights')(LABELS, SCORES, feats)
    assert val(got) == pytest.approx(-(2. * ln(p0) + 2. * ln(p1)) / 2., abs=1e-5)
    poly = lambda p: ln(p) - 1. + p                                                           # epsilon = 1 (:567-605)
    assert val(R.make_loss_fn('poly_one_softmax_loss')(LABELS, SCORES, feats)) == pytest.approx(-(poly(p0) + 2. * poly(p1)) / 2., abs=1e-5)
    got = R.make_loss_fn('poly_one_softmax_loss', weights_feature_name='weights')(LABELS, SCORES, feats)
    assert val(got) == pytest.approx(-(2. * poly(p0) + 2. * poly(p1)) / 2., abs=1e-5)
    labels = [[0., 0., 1.], [0., 1., 2.]]                                                     # unique softmax (:607-642)
    q = softmax(SCORES[1][:2])[1]
    assert val(R.make_loss_fn('unique_softmax_loss')(labels, SCORES, feats)) == pytest.approx(-(ln(p0) + ln(q) + 3. * ln(p1)) / 2., abs=1e-5)
    got = R.make_loss_fn('unique_softmax_loss', weights_feature_name='weights')(labels, SCORES, feats)
    assert val(got) == pytest.approx(-(2. * ln(p0) + ln(q) + 3. * ln(p1)) / 2., abs=1e-5)
    for key in ('softmax_loss', 'poly_one_softmax_loss', 'unique_softmax_loss'):
        assert abs(val(R.make_loss_fn(key, reduction=SUM)(labels, SCORES, feats)) - val(R.make_loss_fn(key, reduction=MEAN)(labels, SCORES, feats))) > 1e-3


def test_make_pointwise_losses():  # losses_test.py:644-712
    scores = [[0.2, 0.5, 0.3], [0.2, 0.3, 0.5]]
    labels = [[0., 0., 1.], [0., 0., 1.]]
    feats = {'weights': [[2.], [1.]]}
    ce = lambda l, s: sum(max(x, 0.) - x * y + ln(1. + math.exp(-abs(x))) for y, x in zip(l, s))
    se = lambda l, s: sum((x - y) ** 2 for y, x in zip(l, s))
    for key, f in (('sigmoid_cross_entropy_loss', ce), ('mean_squared_loss', se)):
        a, b = f(labels[0], scores[0]), f(labels[1], scores[1])
        assert val(R.make_loss_fn(key)(labels, scores, feats)) == pytest.approx((a + b) / 6., abs=1e-5)
        assert val(R.make_loss_fn(key, weights_feature_name='weights')(labels, scores, feats)) == pytest.approx((2. * a + b) / 6., abs=1e-5)


def test_make_list_mle_loss():  # losses_test.py:714-746
    scores = [[0., ln(3), ln(2)], [0., ln(2), ln(3)]]
    labels = [[0., 2., 1.], [1., 0., 2.]]
    feats = {'weights': [[2.], [1.]]}
    a = ln(3. / 6.) + ln(2. / 3.) + ln(1. / 1.)
    b = ln(3. / 6.) + ln(1. / 3.) + ln(2. / 2.)
    assert val(R.make_loss_fn('list_mle_loss')(labels, scores, feats)) == pytest.approx(-(a + b) / 2., abs=1e-5)
    assert val(R.make_loss_fn('list_mle_loss', weights_feature_name='weights')(labels, scores, feats)) == pytest.approx(-(2. * a + b) / 2., abs=1e-5)


def test_make_approx_losses_use_temperature_0_1():  # losses_test.py:748-794, 856-900
    scores = [[1.4, -2.8, -0.4], [0., 1.8, 10.2], [1., 1.2, -3.2]]
    labels = [[0., 2., 1.], [1., 0., 3.], [0., 0., 0.]]
    feats = {'weights': [[2.], [1.], [1.]]}
    a = (1. / (3. / ln(2) + 1. / ln(3))) * (3. / ln(4) + 1. / ln(3))
    b = (1. / (7. / ln(2) + 1. / ln(3))) * (7. / ln(2) + 1. / ln(4))
    assert val(R.make_loss_fn('approx_ndcg_loss', reduction=SUM)(labels, scores, feats)) == pytest.approx(-(a + b), abs=1e-5)
    got = R.make_loss_fn('approx_ndcg_loss', weights_feature_name='weights', reduction=SUM)(labels, scores, feats)
    assert val(got) == pytest.approx(-(2. * a + b), abs=1e-5)
    t10 = R.make_loss_fn('approx_ndcg_loss', params={'temperature': 10})(labels, scores, feats)
    t001 = R.make_loss_fn('approx_ndcg_loss', params={'temperature': 0.01})(labels, scores, feats)
    assert abs(val(t10) - val(t001)) > 1e-3
    # the same closed form through NeuralSort at temperature 0.1 (:939-987)
    got = R.make_loss_fn('neural_sort_ndcg_loss', params={'temperature': 0.1}, reduction=SUM)(labels, scores, feats)
    assert val(got) == pytest.approx(-(a + b), abs=1e-5)
    got = R.make_loss_fn('neural_sort_ndcg_loss', params={'temperature': 0.1}, weights_feature_name='weights', reduction=SUM)(labels, scores, feats)
    assert val(got) == pytest.approx(-(2. * a + b), abs=1e-5)
    labels = [[0., 0., 1.], [1., 0., 1.], [0., 0., 0.]]                                      # ApproxMRR (:856-900)
    assert val(R.make_loss_fn('approx_mrr_loss', reduction=SUM)(labels, scores, feats)) == pytest.approx(-(1. / 2. + 1. / 2. * (1. / 3. + 1.)), abs=1e-5)
    got = R.make_loss_fn('approx_mrr_loss', weights_feature_name='weights', reduction=SUM)(labels, scores, feats)
    assert val(got) == pytest.approx(-(2. * 1. / 2. + 1. / 2. * (1. / 3. + 1.)), abs=1e-5)


def test_make_gumbel_approx_ndcg_with_the_documented_sample():  # losses_test.py:796-854
    scores = [[1.4, -2.8, -0.4], [0., 1.8, 10.2], [1., 1.2, -3.2]]
    labels = [[0., 2., 1.], [1., 0., 3.], [1., 0., 0.]]
    sampled = [[-1.7508768e-1, -4.6947412, -1.887345], [-3.6629683e-1, -3.4472363, -1.2914587],
               [-7.654705, -8.3514204, -7.1014347e-4], [-10.080214, -8.7212124, -2.0500139e-4],
               [-2.0658800e-1, -1.678545, -46.035358], [-2.3852456e-1, -1.550176, -46.028168]]
    # Gumbel noise that yields them up to a constant per row (log-softmax is shift-invariant; the shift keeps the noise
    # inside what a uniform in (0, 1) can produce: at temperature 0.001 only the ORDER of each sampled row matters)
    g = torch.tensor(sampled).reshape(3, 2, 3) - torch.tensor(scores).unsqueeze(1)
    g = g - g.max(dim=2, keepdim=True).values
    uniform = torch.exp(-torch.exp(-g)).clamp(1e-30, 1. - 1e-7)
    feats = {'weights': [[2.], [1.], [1.]]}
    kw = dict(reduction=SUM, params={'temperature': 0.001}, gumbel_params={'sample_size': 2, 'seed': 1}, uniform=uniform)
    a = (2. / (3. / ln(2) + 1. / ln(3))) * (1. / ln(3) + 3. / ln(4))
    b = (1. / (7. / ln(2) + 1. / ln(3))) * (7. / ln(2) + 1. / ln(3)) + (1. / (7. / ln(2) + 1. / ln(3))) * (7. / ln(2) + 1. / ln(4))
    c = (2. / (1. / ln(2))) * (1. / ln(2))
    assert val(R.make_loss_fn('gumbel_approx_ndcg_loss', **kw)(labels, scores, feats)) == pytest.approx(-(a + b + c), abs=1e-4)
    got = R.make_loss_fn('gumbel_approx_ndcg_loss', weights_feature_name='weights', **kw)(labels, scores, feats)
    assert val(got) == pytest.approx(-(2. * a + b + c), abs=1e-4)


def test_make_neural_sort_cross_entropy_loss():  # losses_test.py:902-937
    scores = [[0.2, 0.5, 0.3], [0.2, 0.3, 0.5]]
    labels = [[0., 0., 1.], [0., 0., 1.]]
    feats = {'weights': [[2.], [1.]]}

    def smooth_perm(v):                                                                       # NeuralSort, temperature 1
        n = len(v)
        spread = [sum(abs(x - y) for y in v) for x in v]
        return [softmax([(n + 1 - 2 * (i + 1)) * x - s for x, s in zip(v, spread)]) for i in range(n)]

    def xent(p_true, p_pred):
        return sum(-t * ln(1e-20 + q) for rt, rp in zip(p_true, p_pred) for t, q in zip(rt, rp))
    a = xent(smooth_perm(labels[0]), smooth_perm(scores[0]))
    b = xent(smooth_perm(labels[1]), smooth_perm(scores[1]))
    assert val(R.make_loss_fn('neural_sort_cross_entropy_loss')(labels, scores, feats)) == pytest.approx((a + b) / 6., abs=1e-5)
    got = R.make_loss_fn('neural_sort_cross_entropy_loss', weights_feature_name='weights')(labels, scores, feats)
    assert val(got) == pytest.approx((2. * a + b) / 6., abs=1e-5)


@pytest.mark.parametrize('as_string', [False, True])
def test_make_loss_fn_combinations(as_string):  # losses_test.py:989-1172
    scores = [[0.2, 0.5, 0.3], [0.2, 0.3, 0.5]]
    labels = [[0., 0., 1.], [0., 0., 1.]]
    w2, w1, w3 = [[2.], [1.]], [2., 1.], [[[2.], [1.], [0.]], [[0.], [1.], [2.]]]
    feats = {'weights': w2, 'weights_1d': w1, 'weights_3d': w3}
    red = R.Reduction.SUM_BY_NONZERO_WEIGHTS
    hinge = lambda w: val(R.PairwiseHingeLoss().compute(labels, scores, w, red))
    mse = lambda w: val(R.MeanSquaredLoss().compute(labels, scores, w, red))
    keys = 'pairwise_hinge_loss:1.0,mean_squared_loss:1.0' if as_string else ['pairwise_hinge_loss', 'mean_squared_loss']
    assert val(R.make_loss_fn(keys)(labels, scores, feats)) == pytest.approx(hinge(None) + mse(None), abs=1e-5)
    for name in ('weights', 'weights_1d'):                                                   # 1-D weights are per list
        got = R.make_loss_fn(keys, weights_feature_name=name)(labels, scores, feats)
        assert val(got) == pytest.approx(hinge(w2) + mse(w2), abs=1e-5)
    item = [[2., 1., 0.], [0., 1., 2.]]                                                      # 3-D weights are per item
    got = R.make_loss_fn(keys, weights_feature_name='weights_3d')(labels, scores, feats)
    assert val(got) == pytest.approx(hinge(item) + mse(item), abs=1e-5)
    if as_string:
        both = R.make_loss_fn('pairwise_hinge_loss:3.0,mean_squared_loss:2.0', weights_feature_name='weights')
        with pytest.raises(ValueError, match='`loss_weights` has to be None when weights are encoded in `loss_keys`'):
            R.make_loss_fn(keys, [2.0])
    else:
        both = R.make_loss_fn(keys, [3., 2.], weights_feature_name='weights')
        with pytest.raises(ValueError, match='loss_keys cannot be None or empty.'):
            R.make_loss_fn([])
        with pytest.raises(ValueError, match='loss_keys cannot be None or empty.'):
            R.make_loss_fn('')
        with pytest.raises(ValueError, match='loss_keys and loss_weights must have the same size.'):
            R.make_loss_fn(keys, [2.0])
        with pytest.raises(ValueError, match='Invalid loss_key: invalid_key.'):
            R.make_loss_fn(['invalid_key'])(labels, scores, feats)
    assert val(both(labels, scores, feats)) == pytest.approx(3. * hinge(w2) + 2. * mse(w2), abs=1e-5)
    assert abs(val(R.make_loss_fn(keys, reduction=SUM)(labels, scores, feats)) - val(R.make_loss_fn(keys, reduction=MEAN)(labels, scores, feats))) > 1e-3


def test_lambda_weight_factories():  # losses_test.py:1174-1202
    got = R.create_ndcg_lambda_weight().pair_weights(torch.tensor([[2.0, 1.0]]), torch.tensor([[1, 2]])) / 2.
    max_dcg = 3.0 / ln(2.) + 1.0 / ln(3.)
    x = 2. * (1. / ln(2.) - 1. / ln(3.)) / max_dcg
    assert torch.allclose(got, torch.tensor([[[0., x], [x, 0.]]]), atol=1e-6)
    got = R.create_reciprocal_rank_lambda_weight().pair_weights(torch.tensor([[1.0, 2.0]]), torch.tensor([[1, 2]])) / 2.
    assert torch.allclose(got, torch.tensor([[[0., 1. / 2. / 2.5], [1. / 2. / 2.5, 0.]]]), atol=1e-6)
    got = R.create_p_list_mle_lambda_weight(2).individual_weights(torch.tensor([[1.0, 2.0]]), torch.tensor([[1, 2]]))
    assert torch.allclose(got, torch.tensor([[1.0, 0.0]]))
