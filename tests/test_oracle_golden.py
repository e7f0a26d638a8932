"""Pins the CPU oracle against the reference's own known-answer tests.

Every test is a transcription of a reference unit test (cited file:line under
/root/reference/tensorflow_ranking/python/); expectations are the reference's
closed-form expressions or literal decimals.  TF is not installable here, so
these literals are the only executable link to the reference.
"""
import math

import pytest
import torch

from oracle import tfr_ref as R

ln = math.log
RED = R.Reduction


def close(a, b, tol=1e-5):
    a = torch.as_tensor(a, dtype=torch.float64).reshape(-1)
    b = torch.as_tensor(b, dtype=torch.float64).reshape(-1)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert torch.allclose(a, b, rtol=tol, atol=tol), (a, b)


def logloss(x):
    return math.log(1. + math.exp(-x))


def softmax(values):
    total = sum(math.exp(v) for v in values)
    return [math.exp(v) / (1e-20 + total) for v in values]


def log2p1(x):
    return math.log2(1. + x)


# ---------------------------------------------------------------- utils_test.py
def test_sort_by_scores_2d():  # utils_test.py:64-81
    scores = [[1., 3., 2.], [1., 2., 3.]]
    positions = torch.tensor([[1, 2, 3], [4, 5, 6]])
    out, = R.sort_by_scores(scores, [positions])
    assert out.tolist() == [[2, 3, 1], [6, 5, 4]]
    out, = R.sort_by_scores(scores, [positions], topn=2)
    assert out.tolist() == [[2, 3], [6, 5]]


def test_sort_by_scores_3d():  # utils_test.py:83-102
    scores = [[1., 3., 2.], [1., 2., 3.]]
    feat = torch.tensor([[[1., 2., 3.], [4., 5., 6.], [7., 8., 9.]],
                         [[10., 20., 30.], [40., 50., 60.], [70., 80., 90.]]])
    out, = R.sort_by_scores(scores, [feat], topn=2)
    assert out.tolist() == [[[4., 5., 6.], [7., 8., 9.]], [[70., 80., 90.], [40., 50., 60.]]]


def test_sort_by_scores_ties_no_shuffle():  # utils_test.py:104-109
    names = torch.tensor([[0, 1, 2]])
    out, = R.sort_by_scores([[2., 1., 1.]], [names])
    assert out.tolist() == [[0, 1, 2]]


def test_sort_by_scores_with_mask():  # utils_test.py:114-126
    scores = [[0., math.inf, 2., -math.inf, 1.]]
    names = torch.tensor([[0, 1, 2, 3, 4]])  # a b c d e
    out, = R.sort_by_scores(scores, [names], mask=[[True, False, True, True, False]])
    assert out.tolist() == [[2, 0, 3, 1, 4]]
    out, = R.sort_by_scores(scores, [names], mask=[[False, True, False, True, True]])
    assert out.tolist() == [[1, 4, 3, 0, 2]]
    out, = R.sort_by_scores(scores, [names])
    assert out.tolist() == [[1, 2, 4, 0, 3]]


def test_sorted_ranks():  # utils_test.py:144-152
    assert R.sorted_ranks([[1., 3., 2.]]).tolist() == [[3, 1, 2]]
    assert R.sorted_ranks([[1., 2., 1.]]).tolist() == [[2, 1, 3]]


def test_padded_nd_indices_doc():  # utils.py:311-327 docstring
    idx = R.padded_nd_indices([[True, True, False]])
    assert idx.tolist() == [[0, 1, 0]]


# ---------------------------------------------------------- losses_impl_test.py
def test_approx_ranks():  # losses_impl_test.py:164-170
    logits = [[100., 300., 200., 0.], [400., 200., 150., 300.]]
    close(R.approx_ranks(logits), [[3., 1., 2., 4.], [1., 3., 4., 2.]])


def test_inverse_max_dcg():  # losses_impl_test.py:172-180
    labels = [[1., 4., 1., 0.], [4., 2., 0., 3.], [0., 0., 0., 0.]]
    close(R.inverse_max_dcg(labels), [[0.04297], [0.033139], [0.]], 1e-5)
    close(R.inverse_max_dcg(labels, topn=1), [[0.04621], [0.04621], [0.]], 1e-5)


def test_ndcg():  # losses_impl_test.py:182-196
    labels = [[1., 4., 1., 0.], [4., 2., 0., 3.], [0., 0., 0., 0.]]
    ranks = torch.tensor([[1, 2, 3, 4], [1, 3, 4, 2], [1, 2, 3, 4]])
    close(R.ndcg(labels), [[0.679685], [0.95176], [0.]])
    close(R.ndcg(labels, ranks), [[0.679685], [1.], [0.]])
    perm_mat = torch.tensor([[[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]],       # third assertion (:184-196):
                             [[1, 0, 0, 0], [0, 0, 0, 1], [0, 1, 0, 0], [0, 0, 1, 0]],       # the branch NeuralSortNDCG
                             [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]]], dtype=torch.float32)   # goes through
    close(R.ndcg(labels, perm_mat=perm_mat), [[0.679685], [1.], [0.]])
    with pytest.raises(ValueError):
        R.ndcg(labels, ranks, perm_mat)


def test_label_diff_lambda_weight():  # losses_impl_test.py:325-336
    w = R.LabelDiffLambdaWeight().pair_weights([[2.0, 1.0, 0.0]], None)
    close(w, [[[0., 1., 2.], [1., 0., 1.], [2., 1, 0.]]])


class TestDCGLambdaWeight:  # losses_impl_test.py:339-433
    labels = [[2.0, 1.0, 0.0]]
    ranks = torch.tensor([[1, 2, 3]])

    def test_default(self):
        w = R.DCGLambdaWeight().pair_weights(self.labels, self.ranks) / 3.
        close(w, [[[0., 1. / 2., 2. * 1. / 6.], [1. / 2., 0., 1. / 2.], [2. * 1. / 6., 1. / 2., 0.]]])

    def test_smooth_fraction(self):
        w = R.DCGLambdaWeight(smooth_fraction=1.0).pair_weights(self.labels, self.ranks) / 3.
        close(w, [[[0., 1. / 2., 2. * 2. / 3.], [1. / 2., 0., 1. / 6.], [2. * 2. / 3., 1. / 6., 0.]]])
        w = R.DCGLambdaWeight(topn=1, smooth_fraction=1.0).pair_weights(self.labels, self.ranks) / 3.
        close(w, [[[0., 1., 2.], [1., 0., 0.], [2., 0., 0.]]])

    def test_topn(self):
        w = R.DCGLambdaWeight(topn=1).pair_weights(self.labels, self.ranks) / 3.
        close(w, [[[0., 1. / 2., 1. / 3.], [1. / 2., 0., 0.], [1. / 3., 0., 0.]]])

    def test_invalid_labels(self):
        w = R.DCGLambdaWeight().pair_weights([[2.0, 1.0, -1.0]], self.ranks) / 3.
        close(w, [[[0., 1. / 2., 0.], [1. / 2., 0., 0.], [0., 0., 0.]]])

    def test_gain_and_discount(self):
        lw = R.DCGLambdaWeight(gain_fn=R.pow_minus_1, rank_discount_fn=R.log1p_inverse)
        w = lw.pair_weights([[2.0, 1.0]], torch.tensor([[1, 2]])) / 2.
        e = 2. * (1. / ln(2.) - 1. / ln(3.))
        close(w, [[[0., e], [e, 0.]]])

    def test_normalized(self):
        w = R.DCGLambdaWeight(normalized=True).pair_weights([[1.0, 2.0]], torch.tensor([[1, 2]])) / 2.
        close(w, [[[0., 1. / 2. / 2.5], [1. / 2. / 2.5, 0.]]])

    def test_individual_weights(self):
        w = R.DCGLambdaWeight(normalized=True).individual_weights([[1.0, 2.0]], torch.tensor([[1, 2]]))
        close(w, [[1. / 2.5 / 1., 2. / 2.5 / 2.]])

    def test_bad_smooth_fraction(self):  # losses_impl.py:329-331
        with pytest.raises(ValueError):
            R.DCGLambdaWeight(smooth_fraction=1.5)


def test_pairwise_compute_per_list():  # losses_impl_test.py:530-541
    scores = [[1., 3., 2.], [1., 2., 3.]]
    labels = [[0., 0., 1.], [0., 0., 2.]]
    w = [[2., 3., 4.], [1., 1., 1.]]
    losses, weights = R.PairwiseHingeLoss().compute_per_list(labels, scores, w)
    close(losses, [1., 0.])
    close(weights, [8., 2.])


def test_listwise_compute_per_list():  # losses_impl_test.py:543-554 (T NOT applied)
    scores = [[1., 3., 2.], [1., 2., 3.]]
    labels = [[0., 0., 1.], [0., 0., 2.]]
    w = [[2., 3., 4.], [1., 1., 1.]]
    losses, weights = R.ApproxNDCGLoss().compute_per_list(labels, scores, w)
    close(losses, [-0.63093, -0.796248])
    close(weights, [4., 1.])


RAGGED_SCORES = [[1., 3., 2.], [1., 3.]]
RAGGED_LABELS = [[0., 0., 1.], [0., 2.]]
RAGGED_W = [[2., 3., 4.], [1., 1.]]


@pytest.mark.parametrize('ctor,exp_l,exp_w', [   # losses_impl_test.py:556-580
    (R.SigmoidCrossEntropyLoss, [1.3644443, -0.8190755], [9., 2.]),
    (R.MeanSquaredLoss, [3.6666667, 1.], [9., 2.]),
    (R.PairwiseHingeLoss, [1., 0.], [8., 1.]),
    (R.PairwiseLogisticLoss, [0.813262, 0.126928], [8., 1.]),
    (R.SoftmaxLoss, [1.407606, 0.126928], [4., 2.]),
    (R.ApproxNDCGLoss, [-0.63093, -0.922917], [4., 1.]),
    (R.ApproxMRRLoss, [-0.5, -0.893493], [4., 1.]),
    (R.ListMLELoss, [3.534534, 0.126928], [4., 1.]),
    (R.UniqueSoftmaxLoss, [1.407606, 0.380784], [4., 1.]),
    (R.NeuralSortCrossEntropyLoss, [1.816267, 0.365334], [4., 1.]),
    (R.NeuralSortNDCGLoss, [-0.761571, -0.956006], [4., 1.]),
])
def test_compute_per_list_ragged(ctor, exp_l, exp_w):
    losses, weights = ctor(ragged=True).compute_per_list(RAGGED_LABELS, RAGGED_SCORES, RAGGED_W)
    close(losses, exp_l)
    close(weights, exp_w)


@pytest.mark.parametrize('ctor,expected', [   # losses_impl_test.py:582-611
    (R.SigmoidCrossEntropyLoss, [[1.313262, 3.048587, 0.126928], [1.313262, -2.951413, 0.]]),
    (R.MeanSquaredLoss, [[1., 9., 1.], [1., 1., 0.]]),
    (R.PairwiseHingeLoss, [[[0., 0., 0.], [0., 0., 0.], [0., 2., 0.]],
                           [[0., 0., 0.], [0., 0., 0.], [0., 0., 0.]]]),
    (R.PairwiseLogisticLoss, [[[0., 0., 0.], [0., 0., 0.], [0.313262, 1.313262, 0.]],
                              [[0., 0., 0.], [0.126928, 0., 0.], [0., 0., 0.]]]),
    (R.ApproxNDCGLoss, [[-0.63093], [-0.922917]]),
    (R.ApproxMRRLoss, [[-0.5], [-0.893493]]),
    (R.NeuralSortCrossEntropyLoss, [[1.816267], [0.365334]]),
    (R.NeuralSortNDCGLoss, [[-0.761571], [-0.956006]]),
])
def test_compute_unreduced_loss_ragged(ctor, expected):
    losses, weights = ctor(ragged=True).compute_unreduced_loss(RAGGED_LABELS, RAGGED_SCORES)
    close(losses * weights, expected)


@pytest.mark.parametrize('ctor,expected', [   # losses_impl_test.py:613-636
    (R.PairwiseLogisticLoss, [[[2.], [3.], [4.]], [[1.], [1.], [0.]]]),
    (R.SoftmaxLoss, [[4.], [1.]]),
    (R.ApproxNDCGLoss, [[4.], [1.]]),
])
def test_normalize_weights_ragged(ctor, expected):
    close(ctor(ragged=True).normalize_weights(RAGGED_LABELS, RAGGED_W), expected)


class TestPairwiseLogistic:  # losses_impl_test.py:639-724
    scores = [[1., 3., 2.], [1., 2., 3.]]
    labels = [[0., 0., 1.], [0., 0., 2.]]

    def test_plain(self):
        r = R.PairwiseLogisticLoss().compute(self.labels, self.scores, None, RED.MEAN)
        close(r, (logloss(3. - 2.) + logloss(1. - 2.) + logloss(3. - 1.) + logloss(3. - 2.)) / 4.)

    def test_list_weights(self):
        r = R.PairwiseLogisticLoss().compute(self.labels, self.scores, [[1.], [2.]], RED.MEAN)
        close(r, (1. * (logloss(3. - 2.) + logloss(1. - 2.))
                  + 2. * (logloss(3. - 2.) + logloss(3. - 1.))) / 6.)

    def test_example_weights(self):
        r = R.PairwiseLogisticLoss().compute(self.labels, self.scores,
                                             [[1., 1., 2.], [1., 1., 1.]], RED.MEAN)
        close(r, ((2. * logloss(3. - 2.) + 2. * logloss(1. - 2.))
                  + (logloss(3. - 1.) + logloss(3. - 2.))) / 6.)

    def test_lambda_weights(self):
        r = R.PairwiseLogisticLoss(lambda_weight=R.DCGLambdaWeight()).compute(
            self.labels, self.scores, None, RED.MEAN)
        e = (((3. / 2.) * logloss(3. - 2.) + (3. / 2.) * logloss(1. - 2.))
             + ((1. / 1.) * logloss(3. - 1.) + (3. / 1.) * logloss(3. - 2.))) / (
                 (3. / 2.) + (3. / 2.) + (1. / 1.) + (3. / 1.))
        close(r, e)

    def test_invalid_labels(self):
        r = R.PairwiseLogisticLoss().compute([[0., -1., 1.]], [[1., 3., 2.]], None, RED.MEAN)
        close(r, logloss(2. - 1.))

    def test_mask(self):
        r = R.PairwiseLogisticLoss().compute([[1., 0., 0.], [0., 0., 2.]], self.scores, None, RED.MEAN,
                                             mask=[[True, False, True], [True, True, True]])
        close(r, (logloss(1. - 2.) + logloss(3. - 1.) + logloss(3. - 2.)) / 3.)


class TestSoftmax:  # losses_impl_test.py:1085-1205
    scores = [[1., 3., 2.], [1., 2., 3.], [1., 2., 3.]]
    red = RED.SUM_BY_NONZERO_WEIGHTS

    def test_plain(self):
        labels = [[0., 0., 1.], [0., 0., 2.], [0., 0., 0.]]
        r = R.SoftmaxLoss().compute(labels, self.scores, None, self.red)
        close(r, -(ln(softmax(self.scores[0])[2]) + ln(softmax(self.scores[1])[2]) * 2.) / 2.)

    def test_example_weights(self):
        labels = [[0., 0., 1.], [1., 1., 2.], [0., 0., 0.]]
        w = [[1., 1., 1.], [1., 2., 3.], [1., 0., 1.]]
        p = [softmax(s) for s in self.scores]
        r = R.SoftmaxLoss().compute(labels, self.scores, w, self.red)
        close(r, -(ln(p[0][2]) * 1. + ln(p[1][0]) * 1. * 1. + ln(p[1][1]) * 1. * 2.
                   + ln(p[1][2]) * 2. * 3.) / 2.)

    def test_list_weights(self):
        labels = [[1., 2., 1.], [0., 0., 2.], [0., 0., 0.]]
        p = [softmax(s) for s in self.scores]
        r = R.SoftmaxLoss().compute(labels, self.scores, [[2.], [1.], [1.]], self.red)
        close(r, -(ln(p[0][0]) * 1. * 2. + ln(p[0][1]) * 2. * 2. + ln(p[0][2]) * 1. * 2.
                   + ln(p[1][2]) * 2. * 1.) / 2.)

    def test_lambda_weights(self):
        labels = [[0., 0., 1.], [0., 0., 2.], [0., 0., 0.]]
        lw = R.DCGLambdaWeight(rank_discount_fn=R.log1p_inverse)
        r = R.SoftmaxLoss(lambda_weight=lw).compute(labels, self.scores, None, self.red)
        close(r, -(ln(softmax(self.scores[0])[2]) / ln(1. + 2.)
                   + ln(softmax(self.scores[1])[2]) * 2. / ln(1. + 1.)) / 2.)

    def test_per_list(self):
        losses, weights = R.SoftmaxLoss().compute_per_list(
            [[0., 0., 1.], [0., 0., 2.]], [[1., 3., 2.], [1., 2., 3.]], [[2., 3., 4.], [1., 1., 1.]])
        close(losses, [1.407606, 0.407606])
        close(weights, [4., 2.])

    def test_invalid_labels(self):
        r = R.SoftmaxLoss().compute([[0., -1., 1.]], [[1., 3., 2.]], None, self.red)
        close(r, -(ln(softmax([1, 2])[1])))

    def test_mask(self):
        r = R.SoftmaxLoss().compute([[0., 1., 1.]], [[1., 2., 3.]], None, self.red,
                                    mask=[[True, False, True]])
        close(r, -(ln(softmax([1, 3])[1])))

    def test_padded_zero_labels(self):
        a = R.SoftmaxLoss().compute_unreduced_loss([[0., -1.]], [[0., 0.]])[0]
        b = R.SoftmaxLoss().compute_unreduced_loss([[0.]], [[0.]])[0]
        close(a, b)

    def test_fully_padded(self):
        r = R.SoftmaxLoss().compute_unreduced_loss([[-1., -1.]], [[0., 0.]])[0]
        close(r, [0.0])


class TestApproxNDCG:  # losses_impl_test.py:1662-1724
    def test_weights(self):
        scores = [[1.4, -2.8, -0.4], [0., 1.8, 10.2], [1., 1.2, -3.2]]
        labels = [[0., 2., 1.], [1., 0., -1.], [0., 0., 0.]]
        weights = [[2.], [1.], [1.]]
        example_weights = [[1., 2., 3.], [4., 5., 6.], [7., 8., 9.]]
        norm_weights = []
        for weight, label in zip(example_weights, labels):
            sum_label = sum(max(0, l) for l in label)
            norm_weights.append(sum(w * max(0, l) for w, l in zip(weight, label)) / sum_label
                                if sum_label else 0)
        loss = R.ApproxNDCGLoss(temperature=0.1)
        base = (1 / (3 / ln(2) + 1 / ln(3))) * (3 / ln(4) + 1 / ln(3))
        close(loss.compute(labels, scores, None, RED.SUM), -(base + ln(2) * (1 / ln(3))))
        close(loss.compute(labels, scores, weights, RED.SUM), -(2 * base + 1 * ln(2) * (1 / ln(3))))
        close(loss.compute(labels, scores, example_weights, RED.SUM),
              -(norm_weights[0] * base + norm_weights[1] * ln(2) * (1 / ln(3))))

    @pytest.mark.parametrize('big', [1., 1000.])
    def test_mask_and_extreme_labels(self, big):
        loss = R.ApproxNDCGLoss(temperature=1.)
        r = loss.compute([[0., 0., big]], [[1., 3., 2.]], None, RED.SUM_BY_NONZERO_WEIGHTS,
                         mask=[[True, False, True]])
        approxrank = 1. + 1. / (1. + math.exp(-(1. - 2.)))
        close(r, -(1. / math.log(1. + approxrank)) * math.log(2.))


def test_gumbel_structure_with_injected_scores():
    """losses_impl_test.py:198-236 lists TF-RNG sampled scores for seed=1.  The
    noise stream is TF-specific ("parity unpinned"), but the sampler's output is a
    log-softmax, so feeding noise G = sampled - scores must reproduce `sampled`
    (rows are normalised log-probabilities; the padded entry is ln(1e-20)-ish)."""
    scores = torch.tensor([[1.4, -2.8, -0.4], [0., 1.8, 10.2], [1., 1.2, -3.2]])
    labels = torch.tensor([[0., 0., 1.], [1., 0., 1.], [0., 0., -1.]])
    sampled = torch.tensor([[-1.7508768e-1, -4.6947412, -1.887345],
                            [-3.6629683e-1, -3.4472363, -1.2914587],
                            [-7.654705, -8.3514204, -7.1014347e-4],
                            [-10.080214, -8.7212124, -2.0500139e-4],
                            [-2.0658800e-1, -1.678545, -46.035358],
                            [-2.3852456e-1, -1.550176, -46.028168]])
    g = sampled.reshape(3, 2, 3) - scores.unsqueeze(1)
    g[2, :, 2] = 0.   # padded entry: any noise, it is overwritten by ln(1e-20)
    # log-softmax is shift invariant: move each row into the range the eps-clamped
    # Gumbel transform can represent (G >= -log(-log(1e-20)) ~ -3.83).
    g = g - g.min(dim=-1, keepdim=True).values - 3.0
    # invert G = -log(-log(u+eps)+eps) -> u
    u = torch.exp(-torch.exp(-g.double())).float()
    sampler = R.GumbelSampler(sample_size=2)
    gl, gs, gw = sampler.sample(labels, scores, [[2.], [1.], [1.]], uniform=u)
    assert gl.tolist() == [[0., 0., 1.], [0., 0., 1.], [1., 0., 1.], [1., 0., 1.],
                           [0., 0., -1.], [0., 0., -1.]]
    assert gw.tolist() == [[2.], [2.], [1.], [1.], [1.], [1.]]
    close(gs[:4], sampled[:4], 1e-3)
    close(gs[4:, :2], sampled[4:, :2], 1e-3)
    assert (gs[4:, 2] < -45.).all()


# ------------------------------------------------------------- keras doc values
def test_keras_doc_values():
    yt, yp = [[1., 0.]], [[0.6, 0.8]]
    ryt, ryp = [[1., 0.], [0., 1., 0.]], [[0.6, 0.8], [0.5, 0.8, 0.4]]
    # keras/losses.py:417-428
    close(R.keras_loss_call(R.PairwiseLogisticLoss(), yt, yp), 0.39906943, 1e-6)
    close(R.keras_loss_call(R.PairwiseLogisticLoss(ragged=True), ryt, ryp), 0.3109182, 1e-6)
    # keras/losses.py:770-781
    close(R.keras_loss_call(R.SoftmaxLoss(), yt, yp), 0.7981389, 1e-6)
    close(R.keras_loss_call(R.SoftmaxLoss(ragged=True), ryt, ryp), 0.83911896, 1e-6)
    # keras/losses.py:1183-1194
    close(R.keras_loss_call(R.ApproxNDCGLoss(), yt, yp), -0.655107, 1e-6)
    close(R.keras_loss_call(R.ApproxNDCGLoss(ragged=True), ryt, ryp), -0.80536866, 1e-6)


def test_keras_ragged_sum():  # keras/losses_test.py:769-790
    labels = [[0., 2., 1.], [1., 0.]]
    scores = [[1., 3., 2.], [1., 2.]]   # see losses_test ragged fixtures
    # structural check only: SUM over padded shape equals sum of per-list terms
    dense_l, dense_s, _, _ = R.ragged_to_dense(labels, scores, None)
    a = R.keras_loss_call(R.SoftmaxLoss(ragged=True), labels, scores, reduction=RED.KERAS_SUM)
    b = R.keras_loss_call(R.SoftmaxLoss(), dense_l, dense_s, reduction=RED.KERAS_SUM)
    close(a, b)


# --------------------------------------------------------- metrics_impl_test.py
class TestMRR:  # metrics_impl_test.py:27-136
    def test_single(self):
        close(R.MRRMetric().compute([[0., 0., 1.]], [[1., 3., 2.]])[0], [[0.5]])

    def test_no_rel(self):
        close(R.MRRMetric().compute([[0., 0., 0.]], [[1., 3., 2.]])[0], [[0.]])
        close(R.MRRMetric(topn=1).compute([[0., 0., 1.]], [[1., 3., 2.]])[0], [[0.]])

    def test_topn(self):
        scores = [[3., 2., 1.]] * 3
        labels = [[1., 0., 0.], [0., 1., 0.], [0., 0., 1.]]
        close(R.MRRMetric(topn=1).compute(labels, scores)[0], [[1.], [0.], [0.]])
        close(R.MRRMetric(topn=2).compute(labels, scores)[0], [[1.], [.5], [0.]])
        close(R.MRRMetric(topn=6).compute(labels, scores)[0], [[1.], [.5], [1. / 3.]])

    def test_padded_and_masked(self):
        close(R.MRRMetric().compute([[0., 1., -1.]], [[1., 2., 3.]])[0], [[1.]])
        close(R.MRRMetric().compute([[0., 1., 0.]], [[1., 2., 3.]], mask=[[True, True, False]])[0],
              [[1.]])

    def test_ragged(self):
        m = R.MRRMetric(ragged=True)
        close(m.compute([[0., 1., 0.], [0., 1.]], [[1., 2., 3.], [1., 2.]], None)[0], [[.5], [1.]])

    def test_weights(self):
        w = R.MRRMetric().compute([[1., 0., 0.], [0., 1., 1.]], [[1., 3., 2.], [1., 2., 3.]],
                                  [[2., 5., 1.], [1., 2., 3.]])[1]
        close(w, [[2.], [2.5]])
        w = R.MRRMetric().compute([[0., 0., 0.], [0., 0., 0.]], [[1., 3., 2.], [1., 3., 2.]],
                                  [[2., 5., 1.], [1., 1., 0.]])[1]
        close(w, [[1.], [1.]])
        w = R.MRRMetric(topn=2).compute([[1., 0., 1.], [0., 1., 1.]], [[3., 2., 1.], [1., 3., 2.]],
                                        [[2., 0., 5.], [1., 4., 2.]])[1]
        close(w, [[3.5], [3.]])


class TestNDCG:  # metrics_impl_test.py:643-839
    def test_single(self):
        close(R.NDCGMetric().compute([[0., 1., 0.]], [[3., 2., 1.]])[0],
              [[(1. / log2p1(2.)) / (1. / log2p1(1.))]])
        close(R.NDCGMetric().compute([[0., 0., 0.]], [[3., 2., 1.]])[0], [[0.]])

    def test_graded(self):
        dcg = (2. ** 3. - 1.) / log2p1(2.) + 1. / log2p1(3.)
        mx = (2. ** 3. - 1.) / log2p1(1.) + 1. / log2p1(2.)
        close(R.NDCGMetric().compute([[0., 3., 1., 0.]], [[4., 3., 2., 1.]])[0], [[dcg / mx]])

    def test_custom_fns(self):
        m = R.NDCGMetric(gain_fn=lambda l: l / 2.)
        dcg = (3. / 2.) / log2p1(2.) + (1. / 2.) / log2p1(3.)
        mx = (3. / 2.) / log2p1(1.) + (1. / 2.) / log2p1(2.)
        close(m.compute([[0., 3., 1., 0.]], [[4., 3., 2., 1.]])[0], [[dcg / mx]])
        m = R.NDCGMetric(rank_discount_fn=lambda r: 1.0 / (r + 10.0))
        dcg = (2. ** 3. - 1.) / (2. + 10.) + 1. / (3. + 10.)
        mx = (2. ** 3. - 1.) / (1. + 10.) + 1. / (2. + 10.)
        close(m.compute([[0., 3., 1., 0.]], [[4., 3., 2., 1.]])[0], [[dcg / mx]])

    def test_padded_masked(self):
        dcg = (2. ** 2. - 1.) / log2p1(3.) + 1. / log2p1(1.)
        mx = (2. ** 2. - 1.) / log2p1(1.) + 1. / log2p1(2.)
        close(R.NDCGMetric().compute([[2., -1., 1., 0.]], [[1., 4., 3., 2.]])[0], [[dcg / mx]])
        close(R.NDCGMetric().compute([[2., 2., 1., 0.]], [[1., 4., 3., 2.]],
                                     mask=[[True, False, True, True]])[0], [[dcg / mx]])

    def test_ragged(self):
        m = R.NDCGMetric(ragged=True)
        out = m.compute([[0., 1., 0.], [1., 1., 0., 0.]], [[3., 2., 1.], [4., 1., 2., 3.]], None)[0]
        dcg = [1. / log2p1(2.), 1. / log2p1(1.) + 1. / log2p1(4.)]
        mx = [1. / log2p1(1.), 1. / log2p1(1.) + 1. / log2p1(2.)]
        close(out, [[dcg[0] / mx[0]], [dcg[1] / mx[1]]])

    def test_topn(self):
        scores = [[3., 2., 1.]] * 3
        labels = [[1., 0., 2.], [0., 1., 0.], [0., 0., 1.]]
        mx1 = [(2. ** 2. - 1.) / log2p1(1.), 1. / log2p1(1.), 1. / log2p1(1.)]
        mx = [(2. ** 2. - 1.) / log2p1(1.) + 1. / log2p1(2.), 1. / log2p1(1.), 1. / log2p1(1.)]
        close(R.NDCGMetric(topn=1).compute(labels, scores)[0],
              [[(1. / log2p1(1.)) / mx1[0]], [0.], [0.]])
        close(R.NDCGMetric(topn=2).compute(labels, scores)[0],
              [[(1. / log2p1(1.)) / mx[0]], [(1. / log2p1(2.)) / mx[1]], [0.]])
        close(R.NDCGMetric(topn=6).compute(labels, scores)[0],
              [[(1. / log2p1(1.) + (2. ** 2. - 1.) / log2p1(3.)) / mx[0]],
               [(1. / log2p1(2.)) / mx[1]], [(1. / log2p1(3.)) / mx[2]]])

    def test_weights(self):
        w = R.NDCGMetric().compute([[1., 0., 2.]], [[1., 3., 2.]], [[3., 7., 9.]])[1]
        close(w, [[(1. * 3. + (2. ** 2. - 1.) * 9.) / (1. + (2. ** 2. - 1.))]])
        w = R.NDCGMetric().compute([[0., 0., 0.]], [[1., 3., 2.]], [[2., 4., 4.]])[1]
        close(w, [[1.]])
        out = R.NDCGMetric().compute([[1., 2., 3.]], [[1., 2., 3.]], [[4., 1., 1.]])[0]
        close(out, [[((2 ** 3. - 1.) / log2p1(1) + (2 ** 2. - 1.) / log2p1(2)
                      + (2 ** 1. - 1.) / log2p1(3) * 4.)
                     / ((2 ** 3. - 1.) / log2p1(1) + (2 ** 2. - 1.) / log2p1(3)
                        + (2 ** 1. - 1.) / log2p1(2) * 4.)]])
        out, w = R.NDCGMetric().compute([[1., 2., 3.]], [[1., 2., 3.]], [[0., 0., 0.]])
        close(out, [[0.]])
        close(w, [[0.]])


def test_keras_metric_doc_values():  # keras/metrics.py:218-229, 729-740
    yt, yp = [[0., 1., 1.]], [[3., 1., 2.]]
    close(R.keras_metric_mean(R.MRRMetric(), [(yt, yp, None)]), 0.5, 1e-6)
    close(R.keras_metric_mean(R.NDCGMetric(), [(yt, yp, None)]), 0.6934264, 1e-6)
    ryt, ryp = [[0., 1.], [1., 2., 0.]], [[2., 1.], [2., 5., 4.]]
    close(R.keras_metric_mean(R.MRRMetric(ragged=True), [(ryt, ryp, None)]), 0.75, 1e-6)
    close(R.keras_metric_mean(R.NDCGMetric(ragged=True), [(ryt, ryp, None)]), 0.7974351, 1e-6)


def test_per_list_weights_batch_mean():  # metrics_impl_test.py:108-126; SURVEY A11
    w = R._per_example_weights_to_per_list_weights(
        torch.tensor([[1., 2., 3.], [3., 3., 3.], [1., 1., 1.]]),
        torch.tensor([[0., 0., 1.], [1., 1., 0.], [0., 0., 0.]]))
    close(w, [[3.], [3.], [3.]])


# -------------------------------------------------------------- scorer pieces
def test_flatten_restore_doc():  # keras/layers.py:87-108,195-216
    ctx = torch.tensor([[1.], [2.]])
    ex = torch.tensor([[[1.], [0.], [-1.]], [[0.], [1.], [0.]]])
    mask = [[True, True, False], [True, False, False]]
    fc, fe = R.flatten_list(ctx, ex, mask)
    assert fc.reshape(-1).tolist() == [1., 1., 1., 2., 2., 2.]
    assert fe.reshape(-1).tolist() == [1., 0., 1., 0., 0., 0.]
    out = R.restore_list(torch.tensor([1., 2., 3., 4., 5., 6.]), mask)
    e = math.log(1e-10)
    close(out, [[1., 2., e], [4., e, e]])


def test_rolling_window_indices():  # model_test.py:52-73
    out = R.rolling_window_indices(3, 2, [3, 2, 1])
    assert out.tolist() == [[[0, 1], [1, 2], [2, 0]], [[0, 1], [1, 0], [0, 1]],
                            [[0, 0], [0, 0], [0, 0]]]


# ------------------------------------------------------------------ ApproxMRR
def test_approx_mrr_reference_literals():
    """losses_impl_test.py:1729-1755 (the ragged :568/601 literals are in the parametrised tests above), keras/losses.py:1113-1124 doc values,
    keras/losses_test.py:695-708."""
    scores = torch.tensor([[1.4, -2.8, -0.4], [0., 1.8, 10.2], [1., 1.2, -3.2]])
    labels = torch.tensor([[0., 0., 1.], [1., 0., 1.], [0., 0., 0.]])
    weights = torch.tensor([[2.], [1.], [1.]])
    loss = R.ApproxMRRLoss()
    assert abs(loss.compute(labels, scores, None, R.Reduction.SUM).item()
               + ((1 / 2.) + 1 / 2. * (1 / 3. + 1 / 1.))) < 1e-5
    assert abs(loss.compute(labels, scores, weights, R.Reduction.SUM).item()
               + (2 * 1 / 2. + 1 * 1 / 2. * (1 / 3. + 1 / 1.))) < 1e-5
    got = R.ApproxMRRLoss(temperature=1.).compute(torch.tensor([[0., 0., 1.]]), torch.tensor([[1., 3., 2.]]), None,
                                                  R.Reduction.SUM_BY_NONZERO_WEIGHTS,
                                                  mask=torch.tensor([[True, False, True]]))
    approxrank = 1. + 1. / (1. + math.exp(-(1. - 2.)))
    assert abs(got.item() + 1. / approxrank) < 1e-5
    assert abs(R.keras_loss_call(R.ApproxMRRLoss(), torch.tensor([[1., 0.]]), torch.tensor([[0.6, 0.8]])).item()
               + 0.53168947) < 1e-6
    assert abs(R.keras_loss_call(loss, labels, scores).item() + ((1 / 2.) + 1 / 2. * (1 / 3. + 1 / 1.)) / 3.) < 1e-5


# ------------------------------------------------------------------ hinge / soft zero-one
@pytest.mark.parametrize('ctor,fn', [
    (R.PairwiseHingeLoss, lambda x: max(0, 1. - x)),                   # losses_impl_test.py:729-768
    (R.PairwiseSoftZeroOneLoss, lambda x: 1 / (1 + math.exp(x))),     # losses_impl_test.py:819-858
])
def test_pairwise_hinge_and_soft_zero_one_reference_literals(ctor, fn):
    scores = torch.tensor([[1., 3., 2.], [1., 2., 3.]])
    labels = torch.tensor([[0., 0., 1.], [0., 0., 2.]])
    loss = ctor()
    got = loss.compute(labels, scores, None, R.Reduction.MEAN)
    assert abs(got.item() - (fn(3. - 2.) + fn(1. - 2.) + fn(3. - 1.) + fn(3. - 2.)) / 4.) < 1e-6
    got = loss.compute(labels, scores, torch.tensor([[1.], [2.]]), R.Reduction.MEAN)
    assert abs(got.item() - (1. * (fn(3. - 2.) + fn(1. - 2.)) + 2. * (fn(3. - 2.) + fn(3. - 1.))) / 6.) < 1e-6


def test_pairwise_keras_doc_values():
    yt, yp = torch.tensor([[1., 0.]]), torch.tensor([[0.6, 0.8]])
    assert abs(R.keras_loss_call(R.PairwiseHingeLoss(), yt, yp).item() - 0.6) < 1e-6              # keras/losses.py:350-354
    assert abs(R.keras_loss_call(R.PairwiseSoftZeroOneLoss(), yt, yp).item() - 0.274917) < 1e-6   # :484-488


# ------------------------------------------------------------------ more metrics
from tests.metric_cases import CASES as _METRIC_CASES


@pytest.mark.parametrize('case', _METRIC_CASES, ids=lambda c: '%s@%s:%d' % (c[0], c[1].get('topn'), c[8]))
def test_more_metrics_reference_literals(case):
    cls, kw, labels, scores, weights, mask, exp, exp_w, _line = case
    metric = getattr(R, cls)(**kw)
    out, w = metric.compute(torch.tensor(labels), torch.tensor(scores),
                            None if weights is None else torch.tensor(weights),
                            None if mask is None else torch.tensor(mask))
    if exp is not None:
        close(out, exp, 1e-6)
    if exp_w is not None:
        close(w, exp_w, 1e-6)


from tests.metric_cases import RAGGED_CASES as _RAGGED_METRIC_CASES


@pytest.mark.parametrize('case', _RAGGED_METRIC_CASES, ids=lambda c: '%s:%d' % (c[0], c[6]))
def test_more_metrics_ragged_reference_literals(case):
    cls, kw, labels, scores, exp, exp_w, _line = case
    out, w = getattr(R, cls)(**kw).compute(labels, scores, None)
    close(out, exp, 1e-6)
    if exp_w is not None:
        close(w, exp_w, 1e-6)
    with pytest.raises(ValueError):                                   # ragged inputs need ragged=True (metrics_impl.py:236-241)
        getattr(R, cls)(**{k: v for k, v in kw.items() if k != 'ragged'}).compute(labels, scores, None)


def test_bpref_is_zero_without_input_items():  # metrics_impl_test.py:1498-1506
    out, _ = R.BPrefMetric(topn=None).compute(torch.zeros(1, 0), torch.zeros(1, 0), None)
    close(out, [[0.]], 1e-6)


# ------------------------------------------------------------------ ListMLE
def test_list_mle_reference_literals():
    """losses_impl_test.py:1276-1328 (the tie test :1293-1302 depends on TF's shuffle: unpinned)."""
    ln = math.log
    scores = torch.tensor([[0., ln(3), ln(2)], [0., ln(2), ln(3)]])
    labels = torch.tensor([[0., 2., 1.], [1., 0., 2.]])
    red = R.Reduction.SUM_BY_NONZERO_WEIGHTS
    loss = R.ListMLELoss()
    want = -((ln(3. / 6) + ln(2. / 3) + ln(1. / 1)) + (ln(3. / 6) + ln(1. / 3) + ln(2. / 2))) / 2
    assert abs(loss.compute(labels, scores, None, red).item() - want) < 1e-5
    want = -(2 * (ln(3. / 6) + ln(2. / 3) + ln(1. / 1)) + (ln(3. / 6) + ln(1. / 3) + ln(2. / 2))) / 2
    assert abs(loss.compute(labels, scores, torch.tensor([[2.], [1.]]), red).item() - want) < 1e-5
    lw = R.ListMLELambdaWeight(rank_discount_fn=lambda rank: torch.pow(torch.tensor(2.), 3 - rank) - 1.)
    want = -((3 * ln(3. / 6) + 1 * ln(2. / 3) + 0) + (3 * ln(3. / 6) + 1 * ln(1. / 3) + 0)) / 2
    assert abs(R.ListMLELoss(lambda_weight=lw).compute(labels, scores, None, red).item() - want) < 1e-5
    got = R.ListMLELoss().compute(torch.tensor([[0., 0., 1.]]), torch.tensor([[0., ln(2), ln(3)]]), None, red,
                                  mask=torch.tensor([[True, False, True]]))
    assert abs(got.item() + (ln(3. / 4) + ln(1. / 1))) < 1e-5


# ------------------------------------------------------------------ UniqueSoftmax
def _softmax_py(v):
    m = [math.exp(x) for x in v]
    return [x / sum(m) for x in m]


def test_unique_softmax_reference_literals():
    """losses_impl_test.py:1231-1271, keras/losses.py:961-965."""
    scores = [[1., 3., 2.], [1., 2., 3.], [1., 2., 3.]]
    labels = torch.tensor([[0., 0., 1.], [0., 1., 2.], [0., 0., 0.]])
    red = R.Reduction.SUM_BY_NONZERO_WEIGHTS
    loss = R.UniqueSoftmaxLoss()
    want = -(math.log(_softmax_py(scores[0])[2]) + math.log(_softmax_py(scores[1][:2])[1])
             + math.log(_softmax_py(scores[1])[2]) * 3.) / 3.
    assert abs(loss.compute(labels, torch.tensor(scores), None, red).item() - want) < 1e-5
    want = -(math.log(_softmax_py(scores[0])[2]) * 2. + math.log(_softmax_py(scores[1][:2])[1]) * 1.
             + math.log(_softmax_py(scores[1])[2]) * 3. * 1.) / 2.
    assert abs(loss.compute(labels, torch.tensor(scores), torch.tensor([[2.], [1.], [1.]]), red).item() - want) < 1e-5
    losses, w = loss.compute_per_list(torch.tensor([[0., 0., 1.], [0., 0., 2.]]), torch.tensor([[1., 3., 2.], [1., 2., 3.]]),
                                      torch.tensor([[2., 3., 4.], [1., 1., 1.]]))
    close(losses, [1.407606, 1.222818]); close(w, [4., 1.])
    got = loss.compute(torch.tensor([[0., 1., 1., 0.]]), torch.tensor([[1., 2., 3., 2.]]), None, red,
                       mask=torch.tensor([[True, False, True, True]]))
    assert abs(got.item() + math.log(_softmax_py([1, 3, 2])[1])) < 1e-5
    assert abs(R.keras_loss_call(loss, torch.tensor([[1., 0.]]), torch.tensor([[0.6, 0.8]])).item() - 0.7981389) < 1e-6


class TestLambdaWeightV2YetiPrecision:  # losses_impl_test.py:436-511
    labels = [[2.0, 1.0, 0.0]]
    ranks = torch.tensor([[1, 2, 3]])

    def test_v2(self):
        close(R.DCGLambdaWeightV2().pair_weights(self.labels, self.ranks) / 3.,
              [[[0., 1. / 2., 2. / 6.], [1. / 2., 0., 1. / 2.], [2. / 6., 1. / 2., 0.]]])
        close(R.DCGLambdaWeightV2(topn=1).pair_weights(self.labels, self.ranks) / 3.,
              [[[0., 1., 1. / 2.], [1., 0., 3. / 4.], [1. / 2., 3. / 4., 0.]]])

    def test_yeti(self):
        close(R.YetiDCGLambdaWeight().pair_weights(self.labels, self.ranks) / 3.,
              [[[0., 1. / 2., 0.], [1. / 2., 0., 1. / 2.], [0., 1. / 2., 0.]]])
        close(R.YetiDCGLambdaWeight(topn=1).pair_weights(self.labels, self.ranks) / 3.,
              [[[0., 1., 0.], [1., 0., 3. / 4.], [0., 3. / 4., 0.]]])

    def test_precision(self):
        close(R.PrecisionLambdaWeight(topn=5).pair_weights(self.labels, self.ranks), [[[0.] * 3] * 3])
        close(R.PrecisionLambdaWeight(topn=1).pair_weights(self.labels, self.ranks),
              [[[0., 0., 1.], [0., 0., 0.], [1., 0., 0.]]])


def test_pairwise_mse_reference_literals():
    """losses_impl_test.py:906-1000, keras/losses.py:550-561."""
    scores = torch.tensor([[1., 3., 2.], [1., 2., 3.]])
    labels = torch.tensor([[0., 0., 1.], [0., 0., 2.]])
    red = R.Reduction.MEAN
    sq = lambda a, b: (a - b) ** 2
    want = 2 * (sq(2. - 3., 1.) + sq(2. - 1., 1.) + sq(3. - 1., 0.) + sq(3. - 2., 2.) + sq(3. - 1., 2.) + sq(2. - 1., 0.)) / 12.
    assert abs(R.PairwiseMSELoss().compute(labels, scores, None, red).item() - want) < 1e-5
    want = 2 * (sq(-1., 1.) + sq(1., 1.) + sq(2., 0.) + 2 * sq(1., 2.) + 2 * sq(2., 2.) + 2 * sq(1., 0.)) / 18.
    assert abs(R.PairwiseMSELoss().compute(labels, scores, torch.tensor([[1.], [2.]]), red).item() - want) < 1e-5
    want = ((3. * sq(-1., 1.) + 3. * sq(1., 1.) + 2. * sq(2., 0.)) + 2. * (sq(1., 2.) + sq(2., 2.) + sq(1., 0.))) / 14.
    got = R.PairwiseMSELoss().compute(labels, scores, torch.tensor([[1., 1., 2.], [1., 1., 1.]]), red)
    assert abs(got.item() - want) < 1e-5
    want = (1.5 * sq(-1., 1.) + 1.5 * sq(1., 1.) + 3. * sq(1., 2.) + 1. * sq(2., 2.)) / (1.5 + 1.5 + 3. + 1.)
    got = R.PairwiseMSELoss(lambda_weight=R.DCGLambdaWeight()).compute(labels, scores, None, red)
    assert abs(got.item() - want) < 1e-5
    got = R.PairwiseMSELoss().compute(torch.tensor([[0., -1., 1.]]), torch.tensor([[1., 3., 2.]]), None, red)
    assert abs(got.item() - 0.0) < 1e-5
    got = R.PairwiseMSELoss().compute(torch.tensor([[1., 0., 0.], [0., 0., 2.]]), scores, None, red,
                                      mask=torch.tensor([[True, False, True], [True, True, True]]))
    want = 2. * (sq(2. - 1., -1.) + sq(1., 2.) + sq(2., 2.) + sq(1., 0.)) / 8.
    assert abs(got.item() - want) < 1e-5
    assert abs(R.keras_loss_call(R.PairwiseMSELoss(), torch.tensor([[1., 0.]]), torch.tensor([[0.6, 0.8]])).item() - 1.44) < 1e-6


# ------------------------------------------------------------------ NeuralSort
def _neural_sort_py(logits, temperature=1.0):
    """losses_impl_test.py:126-146 (plain-Python neural sort of the reference's tests)."""
    out = []
    for row in logits:
        n = len(row)
        dsum = [sum(abs(m - l) for m in row) for l in row]
        perm = []
        for i in range(n):
            sc = n + 1 - 2 * (i + 1)
            p = [(sc * l - s) / temperature for l, s in zip(row, dsum)]
            mx = max(p)
            e = [math.exp(v - mx) for v in p]
            perm.append([v / sum(e) for v in e])
        out.append(perm)
    return out


def _softmax_ce_py(p_trues, p_preds):
    return sum(sum(-a * math.log(1e-20 + b) for a, b in zip(t, p)) for t, p in zip(p_trues, p_preds))


def test_neural_sort_reference_literals():
    """losses_impl_test.py:278-299."""
    scores = [[140., -280., -40.], [0., 180., 1020.], [100., 120., -320.]]
    want = [[[1, 0, 0], [0, 0, 1], [0, 1, 0]], [[0, 0, 1], [0, 1, 0], [1, 0, 0]], [[0, 1, 0], [1, 0, 0], [0, 0, 1]]]
    close(R.neural_sort(scores), want, 1e-3)
    got = R.neural_sort([[3.0, 1.0, -1.0, 1000.0, 5.0, 2.0]], mask=[[True, True, True, False, False, True]])
    close(got, [[[0.72140, 0.01321, 0.00000, 0., 0., 0.26539], [0.21183, 0.21183, 0.00053, 0., 0., 0.57581],
                 [0.01204, 0.65723, 0.08895, 0., 0., 0.24178], [0.00004, 0.11849, 0.87557, 0., 0., 0.0059],
                 [0., 0., 0., 0.5, 0.5, 0.], [0., 0., 0., 0.5, 0.5, 0.]]], 1e-4)


def test_neural_sort_losses_reference_literals():
    """losses_impl_test.py:1758-1850."""
    scores = torch.tensor([[1.4, -2.8, -0.4], [0., 1.8, 10.2], [1., 1.2, -3.2]])
    labels = torch.tensor([[0., 2., 1.], [1., 0., -3.], [0., 0., 0.]])
    weights = torch.tensor([[2.], [1.], [1.]])
    p_scores = _neural_sort_py([[1.4, -2.8, -0.4], [0., 1.8, -1000.], [1., 1.2, -3.2]])
    p_labels = _neural_sort_py([[0., 2., 1.], [1., 0., -1000.], [0., 0., 0.]])
    ce = R.NeuralSortCrossEntropyLoss()
    a = _softmax_ce_py(p_labels[0], p_scores[0]) / 3.
    b = _softmax_ce_py(p_labels[1][0:2], p_scores[1][0:2]) / 2.
    assert abs(ce.compute(labels, scores, None, R.Reduction.SUM).item() - (a + b)) < 1e-4
    assert abs(ce.compute(labels, scores, weights, R.Reduction.SUM).item() - (2 * a + b)) < 1e-4
    red = R.Reduction.SUM_BY_NONZERO_WEIGHTS
    want = _softmax_ce_py(_neural_sort_py([[0., 1.]])[0], _neural_sort_py([[1., 2.]])[0]) / 2.
    assert abs(ce.compute(torch.tensor([[0., -1., 1.]]), torch.tensor([[1., 3., 2.]]), None, red).item() - want) < 1e-5
    want = _softmax_ce_py(_neural_sort_py([[0., 1.]])[0], _neural_sort_py([[2., 3.]])[0]) / 2.
    got = ce.compute(torch.tensor([[0., 0., 1., 0., 1.]]), torch.tensor([[2., 4., 3., 3., -1e10]]), None, red,
                     mask=torch.tensor([[True, False, True, False, False]]))
    assert abs(got.item() - want) < 1e-5
    ln = math.log
    nd = R.NeuralSortNDCGLoss(temperature=0.1)
    a = (1 / (3 / ln(2) + 1 / ln(3))) * (3 / ln(4) + 1 / ln(3))
    b = (1 / (1 / ln(2))) * (1 / ln(3))
    assert abs(nd.compute(labels, scores, None, R.Reduction.SUM).item() + (a + b)) < 1e-4
    assert abs(nd.compute(labels, scores, weights, R.Reduction.SUM).item() + (2 * a + b)) < 1e-4
    assert abs(nd.compute(torch.tensor([[0., -1., 1.]]), torch.tensor([[1., 3., 2.]]), None, red).item() + 1.) < 1e-4
    got = nd.compute(torch.tensor([[0., 0., 1., 0., 1.]]), torch.tensor([[2., 4., 3., -5., 1000.0]]), None, red,
                     mask=torch.tensor([[True, False, True, False, False]]))
    assert abs(got.item() + 1.) < 1e-4


# ------------------------------------------------------------------ Circle loss
def _circle_py(labels, scores, gamma=64., margin=0.25):
    """losses_impl_test.py:32-87 (sum over preference pairs)."""
    tot = 0.
    for i in range(len(labels)):
        for j in range(len(labels)):
            if labels[i] > labels[j]:
                tot += math.exp(gamma * max(0., (1 + margin) - scores[i]) * ((1 - margin) - scores[i])
                                + gamma * max(0., scores[j] + margin) * (scores[j] - margin))
    return tot


def test_circle_loss_reference_literals():
    """losses_impl_test.py:1001-1083."""
    scores = [[0.1, 0.3, 0.2], [0.1, 0.2, 0.3]]
    labels = [[0., 0., 1.], [0., 1., 2.]]
    red = R.Reduction.MEAN
    l0, l1 = math.log1p(_circle_py(labels[0], scores[0])), math.log1p(_circle_py(labels[1], scores[1]))
    T = torch.tensor
    assert abs(R.CircleLoss().compute(T(labels), T(scores), None, red).item() - (l0 + l1) / 2) < 1e-5 * (l0 + l1)
    assert abs(R.CircleLoss().compute(T(labels), T(scores), T([[1.], [2.]]), red).item() - (l0 + 2 * l1) / 3) < 1e-5 * (l0 + l1)
    got = R.CircleLoss().compute(T(labels), T(scores), T([[1., 1., 2.], [1., 1., 1.]]), red)
    assert abs(got.item() - (2 * l0 + l1) / 3) < 1e-5 * (l0 + l1)
    labels2 = [[0., 0., 1.], [0., 0., 2.]]
    want = (math.log1p(_circle_py(labels2[0], scores[0], 4., 0.1)) + math.log1p(_circle_py(labels2[1], scores[1], 4., 0.1))) / 2
    assert abs(R.CircleLoss(gamma=4., margin=0.1).compute(T(labels2), T(scores), None, red).item() - want) < 1e-5
    want = math.log1p(_circle_py([0., 1.], [.1, .2]))
    assert abs(R.CircleLoss().compute(T([[0., -1., 1.]]), T([[.1, .3, .2]]), None, red).item() - want) < 1e-5 * want
    want = (math.log1p(_circle_py([1., 0.], [.1, .2])) + math.log1p(_circle_py(labels2[1], scores[1]))) / 2
    got = R.CircleLoss().compute(T([[1., 0., 0.], [0., 0., 2.]]), T(scores), None, red,
                                 mask=T([[True, False, True], [True, True, True]]))
    assert abs(got.item() - want) < 1e-5 * want


def test_diversity_metrics_ragged_reference_literals():
    """metrics_impl_test.py:1163-1176, 1316-1332."""
    from tests.metric_cases import log2p1
    scores = [[1., 3., 4., 2.], [1., 3., 2.]]
    labels = [[[0., 0.], [1., 0.], [1., 1.], [0., 1.]], [[0., 0.], [1., 0.], [0., 1.]]]
    out, _ = R.PrecisionIAMetric(topn=None, ragged=True).compute(labels, scores)
    close(out, [[1. / 2.], [2. / (2. * 3.)]], 1e-6)
    scores = [[1., 3., 2., 4.], [1., 3., 2.]]
    labels = [[[1., 0.], [1., 1.], [0., 1.], [1., 0.]], [[0., 0.], [1., 0.], [0., 1.]]]
    out, _ = R.AlphaDCGMetric(topn=None, ragged=True).compute(labels, scores)
    close(out, [[1. / log2p1(1.) + 1. / log2p1(2.) + 0.5 / log2p1(2.) + 0.5 / log2p1(3.) + 0.25 / log2p1(4.)],
                [1. / log2p1(1.) + 1. / log2p1(2.)]], 1e-6)


def test_mean_squared_loss_reference_literals():
    """losses_impl_test.py:1331-1371, keras/losses.py:1559-1570."""
    mse = lambda lab, sc: sum((a - b) ** 2 for a, b in zip(lab, sc))
    scores = [[0.2, 0.5, 0.3], [0.2, 0.3, 0.5], [0.2, 0.3, 0.5]]
    labels = [[0., 0., 1.], [0., 0., 2.], [0., 0., 0.]]
    red = R.Reduction.SUM_BY_NONZERO_WEIGHTS
    T = torch.tensor
    want = sum(mse(l, s) for l, s in zip(labels, scores)) / 9.
    assert abs(R.MeanSquaredLoss().compute(T(labels), T(scores), None, red).item() - want) < 1e-5
    want = (mse(labels[0], scores[0]) * 2. + mse(labels[1], scores[1]) + mse(labels[2], scores[2])) / 9.
    assert abs(R.MeanSquaredLoss().compute(T(labels), T(scores), T([[2.], [1.], [1.]]), red).item() - want) < 1e-5
    assert abs(R.MeanSquaredLoss().compute(T([[0., -1., 1.]]), T([[1., 3., 2.]]), None, red).item() - 1.) < 1e-5
    got = R.MeanSquaredLoss().compute(T([[0., 1., 1.]]), T([[1., 3., 2.]]), None, red, mask=T([[True, False, True]]))
    assert abs(got.item() - 1.) < 1e-5
    assert abs(R.keras_loss_call(R.MeanSquaredLoss(), T([[1., 0.]]), T([[0.6, 0.8]])).item() - 0.4) < 1e-6


def test_poly_one_softmax_reference_literal():
    """losses_impl_test.py:1208-1226."""
    scores = [[1., 3., 2.], [1., 2., 3.], [1., 2., 3.]]
    labels = torch.tensor([[0., 0., 1.], [0., 0., 2.], [0., 0., 0.]])
    got = R.PolyOneSoftmaxLoss(epsilon=3).compute(labels, torch.tensor(scores), None, R.Reduction.SUM_BY_NONZERO_WEIGHTS)
    s0, s1 = _softmax_py(scores[0])[2], _softmax_py(scores[1])[2]
    want = -((math.log(s0) - 3 * (1 - s0)) + (math.log(s1) - 3 * (1 - s1)) * 2.) / 2.
    assert abs(got.item() - want) < 1e-5


# ------------------------------------------------------------------ round 3: the remaining in-scope literals of
# losses_impl_test.py (hinge / soft zero-one with lambda weights, invalid labels and masks; sigmoid cross-entropy;
# the pointwise compute_per_list).  The reference tests of Ordinal / MultiClass / ClickEM / MixtureEM /
# CoupledRankDistil losses are out of scope ("--").
@pytest.mark.parametrize('ctor,fn', [
    (R.PairwiseHingeLoss, lambda x: max(0, 1. - x)),                   # losses_impl_test.py:772-788
    (R.PairwiseSoftZeroOneLoss, lambda x: 1 / (1 + math.exp(x))),     # losses_impl_test.py:862-878
])
def test_pairwise_hinge_and_soft_zero_one_with_lambda_weights(ctor, fn):
    scores = [[1., 3., 2.], [1., 2., 3.]]
    labels = [[0., 0., 1.], [0., 0., 2.]]
    got = ctor(lambda_weight=R.DCGLambdaWeight()).compute(labels, scores, None, R.Reduction.MEAN)
    want = (((3. / 2.) * fn(3. - 2.) + (3. / 2.) * fn(1. - 2.)) + ((1. / 1.) * fn(3. - 1.) + (3. / 1.) * fn(3. - 2.))) \
        / ((3. / 2.) + (3. / 2.) + (1. / 1.) + (3. / 1.))
    assert abs(got.item() - want) < 1e-6


@pytest.mark.parametrize('ctor,fn', [
    (R.PairwiseHingeLoss, lambda x: max(0, 1. - x)),                   # losses_impl_test.py:790-816
    (R.PairwiseSoftZeroOneLoss, lambda x: 1 / (1 + math.exp(x))),     # losses_impl_test.py:880-903
])
def test_pairwise_hinge_and_soft_zero_one_invalid_labels_and_mask(ctor, fn):
    got = ctor().compute([[0., -1., 1.]], [[1., 3., 2.]], None, R.Reduction.MEAN)
    assert abs(got.item() - fn(2. - 1.)) < 1e-5
    scores = [[1., 3., 2.], [1., 2., 3.]]
    labels = [[1., 0., 0.], [0., 0., 2.]]
    mask = [[True, False, True], [True, True, True]]
    got = ctor().compute(labels, scores, None, R.Reduction.MEAN, mask)
    assert abs(got.item() - (fn(1. - 2.) + fn(3. - 1.) + fn(3. - 2.)) / 3.) < 1e-6


def _sigmoid_ce(labels, logits):  # losses_impl_test.py:111-117
    return sum(max(x, 0) - x * y + math.log(1 + math.exp(-abs(x))) for y, x in zip(labels, logits))


def test_sigmoid_cross_entropy_reference_literals():  # losses_impl_test.py:1376-1418
    scores = [[0.2, 0.5, 0.3], [0.2, 0.3, 0.5], [0.2, 0.3, 0.5]]
    labels = [[0., 0., 1.], [0., 0., 2.], [0., 0., 0.]]
    red = R.Reduction.SUM_BY_NONZERO_WEIGHTS
    per = [_sigmoid_ce(l, s) for l, s in zip(labels, scores)]
    got = R.SigmoidCrossEntropyLoss().compute(labels, scores, None, red)
    assert abs(got.item() - sum(per) / 9.) < 1e-5
    got = R.SigmoidCrossEntropyLoss().compute(labels, scores, [[2.], [1.], [1.]], red)
    assert abs(got.item() - (2. * per[0] + per[1] + per[2]) / 9.) < 1e-5
    want = (math.log(1. + math.exp(-2.)) + math.log(1. + math.exp(1.))) / 2.
    got = R.SigmoidCrossEntropyLoss().compute([[0., -1., 1.]], [[1., 3., 2.]], None, red)          # :1395-1404
    assert abs(got.item() - want) < 1e-5
    got = R.SigmoidCrossEntropyLoss().compute([[0., 1., 1.]], [[1., 3., 2.]], None, red, [[True, False, True]])   # :1406-1418
    assert abs(got.item() - want) < 1e-5


def test_pointwise_compute_per_list():  # losses_impl_test.py:517-528
    scores = [[1., 3., 2.], [1., 2., 3.]]
    labels = [[0., 0., 1.], [0., 0., 2.]]
    losses, weights = R.SigmoidCrossEntropyLoss().compute_per_list(labels, scores, [[2., 3., 4.], [1., 1., 1.]])
    close(losses, [1.3644443, 0.16292572])
    close(weights, [9., 3.])
