"""The plain-C restatement of the headline path (oracle/approx_ndcg_c.c) against the reference's known answers and
against the torch restatement: two independent oracles that must agree (no GPU)."""
import math

import numpy as np
import pytest
import torch

from oracle import c_ref, tfr_ref as R
from tests.common import make_batch

ln = math.log


@pytest.mark.parametrize('variant,tol', [('f64', 2e-6), ('f32_fast', 2e-5)])
def test_reference_known_answers(variant, tol):
    # losses_impl_test.py:1665-1724 / keras/losses_test.py:576-602 (temperature 0.1: ranks are nearly hard)
    scores = [[1.4, -2.8, -0.4], [0., 1.8, 10.2], [1., 1.2, -3.2]]
    labels = [[0., 2., 1.], [1., 0., 3.], [0., 0., 0.]]
    loss, weight, _ = c_ref.approx_ndcg(scores, labels, temperature=0.1, variant=variant)
    n0 = (1 / (3 / ln(2) + 1 / ln(3))) * (3 / ln(4) + 1 / ln(3))
    n1 = (1 / (7 / ln(2) + 1 / ln(3))) * (7 / ln(2) + 1 / ln(4))
    assert abs(loss[0] + n0) < 1e-5 and abs(loss[1] + n1) < 1e-5 and loss[2] == 0.0
    assert weight.tolist() == [1.0, 1.0, 0.0]
    # losses_impl_test.py:543-554 (temperature NOT applied): per-list -0.63093 / -0.796248
    scores = [[1., 3., 2.], [1., 2., 3.]]
    labels = [[0., 0., 1.], [0., 0., 2.]]
    loss, _, _ = c_ref.approx_ndcg(scores, labels, temperature=1.0, variant=variant)
    want = R.ApproxNDCGLoss(temperature=1.0).compute_unreduced_loss(torch.tensor(labels), torch.tensor(scores))[0]
    assert np.abs(loss - want.reshape(-1).numpy()).max() < tol
    # an invalid label and an explicit mask (losses_impl_test.py:1708-1724)
    loss, weight, _ = c_ref.approx_ndcg([[1., 3., 2.]], [[0., 0., 1.]], mask=[[True, False, True]], temperature=1.0,
                                        variant=variant)
    approxrank = 1. + 1. / (1. + math.exp(-(1. - 2.)))
    assert abs(loss[0] + (1. / math.log(1. + approxrank)) * math.log(2.)) < 1e-5 and weight[0] == 1.0


@pytest.mark.parametrize('B,L', [(7, 1), (5, 3), (33, 50), (16, 200), (3, 1000)])
def test_c_and_torch_restatements_agree(B, L):
    labels, logits = make_batch(B, L, seed=900 + L)
    if B >= 3:
        labels[1] = -1.0                                    # an empty list
        labels[2] = torch.where(labels[2] >= 0, torch.zeros_like(labels[2]), labels[2])     # no relevant item
    lg = logits.clone().requires_grad_(True)
    o = R.ApproxNDCGLoss(temperature=0.1)
    l_t, w_t = o.compute_unreduced_loss(labels, o.get_logits(lg))
    l_t.sum().backward()
    want_g = lg.grad.numpy()
    for variant, tol in (('f64', 3e-6), ('f32_fast', 5e-5)):
        loss, weight, grad = c_ref.approx_ndcg(logits.numpy(), labels.numpy(), temperature=0.1, variant=variant)
        assert np.array_equal(weight, w_t.reshape(-1).numpy())
        live = weight > 0
        assert np.abs(loss - l_t.detach().reshape(-1).numpy())[live].max(initial=0.0) < tol
        assert np.abs(grad - want_g).max() <= tol * max(1.0, np.abs(want_g).max())
        assert not grad[labels.numpy() < 0].any()           # no gradient on padding
    assert c_ref.threads() >= 1
    with pytest.raises(ValueError):
        c_ref.approx_ndcg(logits.numpy(), labels.numpy(), temperature=0.0)


def test_softmax_c_known_answers_and_agreement():
    # keras/losses_test.py:284-308, 735-741: Keras AUTO = sum_b loss_b * weight_b / B
    scores = [[1., 3., 2.], [1., 2., 3.], [1., 2., 3.]]
    labels = [[0., 0., 1.], [0., 0., 2.], [0., 0., 0.]]
    pick = lambda s, k: math.exp(s[k]) / sum(math.exp(x) for x in s)
    loss, weight, _ = c_ref.softmax(scores, labels)
    assert abs(float((loss * weight).sum()) / 3. + (ln(pick(scores[0], 2)) + ln(pick(scores[1], 2)) * 2.) / 3.) < 1e-6
    loss, weight, _ = c_ref.softmax([[1., 3., 2.]], [[0., -1., 1.]])
    assert abs(loss[0] + ln(pick([1., 2.], 1))) < 1e-6 and weight[0] == 1.0
    for B, L, T in ((9, 7, 1.0), (33, 100, 0.5), (4, 1000, 2.0)):
        lb, lg = make_batch(B, L, seed=700 + L)
        lb[1] = -1.0
        lb[2] = torch.where(lb[2] >= 0, torch.zeros_like(lb[2]), lb[2])
        x = lg.clone().requires_grad_(True)
        l_t, w_t = R.SoftmaxLoss(temperature=T).compute_unreduced_loss(lb, x)
        l_t.sum().backward()
        loss, weight, grad = c_ref.softmax(lg.numpy(), lb.numpy(), temperature=T)
        assert np.abs(loss - l_t.detach().numpy()).max() < 2e-5 * max(1.0, float(l_t.detach().abs().max()))
        assert np.array_equal(weight, w_t.numpy())
        assert np.abs(grad - x.grad.numpy()).max() < 2e-6


def _py_pairwise_ndcg(labels, scores):
    """sum over pairs of the Keras NDCGLambdaWeight() weight times the logistic loss, in plain Python."""
    L = len(labels)
    order = sorted(range(L), key=lambda i: (-scores[i], i))
    rank = {i: p + 1 for p, i in enumerate(order)}
    gain = [2.0 ** l - 1.0 for l in labels]
    D = lambda r: math.log(2.0) / math.log1p(r)
    ideal = sum(g * D(p + 1) for p, g in enumerate(sorted(gain, reverse=True)))
    total = 0.0
    for i in range(L):
        for j in range(L):
            if labels[i] > labels[j]:
                rd = abs(rank[i] - rank[j])
                w = abs(gain[i] - gain[j]) / ideal * abs(D(rd) - D(rd + 1)) * L
                total += w * math.log1p(math.exp(-(scores[i] - scores[j])))
    return total


def test_pairwise_logistic_ndcg_c_against_plain_python_and_torch():
    scores = [[1., 3., 2., 0.5], [0.2, -0.4, 1.1, 2.0]]
    labels = [[0., 0., 1., 2.], [3., 0., 1., 1.]]
    out, _ = c_ref.pairwise_logistic_ndcg(scores, labels)
    for b in range(2):
        assert abs(out[b] - _py_pairwise_ndcg(labels[b], scores[b])) < 1e-5
    for B, L, T in ((9, 7, 1.0), (17, 50, 1.0), (5, 200, 0.5)):
        lb, lg = make_batch(B, L, seed=800 + L)
        lb[1] = -1.0
        lb[2] = torch.where(lb[2] >= 0, torch.ones_like(lb[2]), lb[2])                # all labels equal: no pair
        x = lg.clone().requires_grad_(True)
        o = R.PairwiseLogisticLoss(lambda_weight=R.NDCGLambdaWeight(), temperature=T)
        l_t, w_t = o.compute_unreduced_loss(lb, o.get_logits(x))        # the Keras __call__ order (keras/losses.py:277)
        tot = (l_t * w_t).sum(dim=(1, 2))
        tot.sum().backward()
        out, grad = c_ref.pairwise_logistic_ndcg(lg.numpy(), lb.numpy(), temperature=T)
        assert np.abs(out - tot.detach().numpy()).max() < 2e-5 * max(1.0, float(tot.detach().abs().max()))
        assert np.abs(grad - x.grad.numpy()).max() < 2e-5 * max(1.0, float(x.grad.abs().max()))
        assert out[1] == 0.0 and out[2] == 0.0


def test_ndcg_mrr_c_known_answers_and_agreement():
    # keras/metrics.py:218-229, 729-740 doc values; keras/metrics_test.py:296-396, 856-992
    ndcg, mrr = c_ref.ndcg_mrr([[3., 1., 2.]], [[0., 1., 1.]])
    assert abs(ndcg[0] - 0.6934264) < 1e-6 and abs(mrr[0] - 0.5) < 1e-7
    scores = [[1., 3., 2.], [1., 2., 3.], [3., 1., 2.]]
    labels = [[0., 0., 1.], [0., 1., 2.], [0., 1., 0.]]
    ndcg, mrr = c_ref.ndcg_mrr(scores, labels)
    assert np.allclose(mrr, [1 / 2., 1., 1 / 3.], atol=1e-7)
    dcg = lambda l, r: (2.0 ** l - 1.0) / math.log2(1.0 + r)
    assert abs(ndcg[0] - (dcg(0., 1) + dcg(1., 2) + dcg(0., 3)) / (dcg(1., 1) + dcg(0., 2) + dcg(0., 3))) < 1e-6
    assert abs(ndcg[1] - 1.0) < 1e-7
    _, mrr1 = c_ref.ndcg_mrr(scores, labels, topn=1)
    assert mrr1.tolist() == [0.0, 1.0, 0.0]
    for B, L, topn in ((9, 7, None), (33, 100, 10), (64, 200, 5), (4, 1000, 1)):
        lb, lg = make_batch(B, L, seed=600 + L)
        lb[1] = -1.0
        lb[2] = torch.where(lb[2] >= 0, torch.zeros_like(lb[2]), lb[2])
        want_n, _ = R.NDCGMetric(topn=topn).compute(lb, lg)
        want_m, _ = R.MRRMetric(topn=topn).compute(lb, lg)
        ndcg, mrr = c_ref.ndcg_mrr(lg.numpy(), lb.numpy(), topn=topn)
        assert np.abs(ndcg - want_n.reshape(-1).numpy()).max() < 2e-6
        assert np.abs(mrr - want_m.reshape(-1).numpy()).max() < 1e-7


def test_tie_rule_is_the_same_in_both_restatements():
    """Tied scores: descending score, ties -> lower index first, invalid items last (the deterministic rule of
    utils_test.py:108-109; the reference shuffles ties randomly otherwise).  Heavily tied integer scores must give the
    same pairwise lambda weights (through the ranks) and the same NDCG / MRR in the C and the torch restatement."""
    g = torch.Generator().manual_seed(11)
    B, L = 40, 30
    logits = torch.randint(-2, 3, (B, L), generator=g).float()           # 5 distinct values: ties everywhere
    labels = torch.randint(0, 4, (B, L), generator=g).float()
    n = torch.randint(L // 2, L + 1, (B,), generator=g)
    labels[torch.arange(L).unsqueeze(0) >= n.unsqueeze(1)] = -1.0
    o = R.PairwiseLogisticLoss(lambda_weight=R.NDCGLambdaWeight())
    l_t, w_t = o.compute_unreduced_loss(labels, logits)
    out, _ = c_ref.pairwise_logistic_ndcg(logits.numpy(), labels.numpy())
    tot = (l_t * w_t).sum(dim=(1, 2)).numpy()
    assert np.abs(out - tot).max() < 2e-5 * max(1.0, np.abs(tot).max())
    for topn in (None, 5):
        want_n, _ = R.NDCGMetric(topn=topn).compute(labels, logits)
        want_m, _ = R.MRRMetric(topn=topn).compute(labels, logits)
        ndcg, mrr = c_ref.ndcg_mrr(logits.numpy(), labels.numpy(), topn=topn)
        assert np.abs(ndcg - want_n.reshape(-1).numpy()).max() < 2e-6
        assert np.abs(mrr - want_m.reshape(-1).numpy()).max() < 1e-7


@pytest.mark.parametrize('B,L', [(5, 1), (6, 7), (9, 50), (4, 300)])
@pytest.mark.parametrize('with_lambda', [False, True])
def test_list_mle_c_against_the_torch_restatement(B, L, with_lambda):
    """oracle/listwise_c.c (fp64 double loops over the definition, losses_impl.py:1541-1576) vs oracle/tfr_ref.py
    (op-for-op torch + autograd): two independent restatements of ListMLE (+ ListMLELambdaWeight, :457-480)."""
    labels, logits = make_batch(B, L, seed=4100 + L)
    g = torch.Generator().manual_seed(L)                   # distinct labels: the reference shuffles ties at random
    labels = torch.where(labels >= 0, labels + torch.rand(labels.shape, generator=g) * 0.5, labels)
    if B >= 3:
        labels[1] = -1.0
    T_ = 0.7
    disc = (lambda rank: 1. / torch.log1p(rank)) if with_lambda else None
    oracle = R.ListMLELoss(lambda_weight=R.ListMLELambdaWeight(disc) if with_lambda else None, temperature=T_)
    lg = logits.clone().requires_grad_(True)
    want = oracle._compute_unreduced_loss_impl(labels, lg / T_)[0]
    want.sum().backward()
    pw = disc(torch.arange(1, L + 1, dtype=torch.float32)).numpy() if with_lambda else None
    loss, grad = c_ref.list_mle(logits.numpy(), labels.numpy(), pos_weight=pw, temperature=T_)
    scale = max(1.0, float(want.detach().abs().max()))
    assert np.abs(loss - want.detach().reshape(-1).numpy()).max() <= 2e-6 * scale * max(1, L) ** 0.5
    assert np.abs(grad - lg.grad.numpy()).max() <= 5e-6 * max(1.0, float(lg.grad.abs().max()))
    assert not grad[labels.numpy() < 0].any()
    with pytest.raises(ValueError):
        c_ref.list_mle(logits.numpy(), labels.numpy(), temperature=0.0)


@pytest.mark.parametrize('B,L', [(5, 1), (6, 7), (9, 50), (4, 300)])
def test_unique_softmax_c_against_the_torch_restatement(B, L):
    """UniqueSoftmaxLoss (losses_impl.py:1250-1281): the C double loop over `l_j < l_i` vs the torch restatement's
    [B, L, L + 1] tensor."""
    labels, logits = make_batch(B, L, seed=4300 + L)       # graded labels: plenty of tie groups
    if B >= 3:
        labels[1] = -1.0
        labels[0] = torch.where(labels[0] >= 0, torch.ones_like(labels[0]) * 2, labels[0])   # one single group
    T_ = 0.8
    oracle = R.UniqueSoftmaxLoss(temperature=T_)
    lg = logits.clone().requires_grad_(True)
    want = oracle._compute_unreduced_loss_impl(labels, lg / T_)[0]
    want.sum().backward()
    loss, grad = c_ref.unique_softmax(logits.numpy(), labels.numpy(), temperature=T_)
    scale = max(1.0, float(want.detach().abs().max()))
    assert np.abs(loss - want.detach().reshape(-1).numpy()).max() <= 5e-6 * scale
    assert np.abs(grad - lg.grad.numpy()).max() <= 2e-5 * max(1.0, float(lg.grad.abs().max()))
    assert not grad[labels.numpy() < 0].any()


def test_listwise_c_reference_known_answers():
    """The reference's own literals: ListMLE losses_impl_test.py:1276-1328, UniqueSoftmax :1231-1271 (per-list sums of
    the closed forms the tests average)."""
    ln = math.log
    scores = [[0., ln(3), ln(2)], [0., ln(2), ln(3)]]
    labels = [[0., 2., 1.], [1., 0., 2.]]
    loss, _ = c_ref.list_mle(scores, labels)
    want = [-(ln(3. / 6) + ln(2. / 3) + ln(1. / 1)), -(ln(3. / 6) + ln(1. / 3) + ln(2. / 2))]
    assert np.abs(loss - np.array(want, dtype=np.float32)).max() < 1e-6
    pw = [2. ** (3 - r) - 1. for r in (1, 2, 3)]             # ListMLELambdaWeight(rank_discount_fn = 2^(3 - rank) - 1)
    loss, _ = c_ref.list_mle(scores, labels, pos_weight=pw)
    want = [-(3 * ln(3. / 6) + 1 * ln(2. / 3)), -(3 * ln(3. / 6) + 1 * ln(1. / 3))]
    assert np.abs(loss - np.array(want, dtype=np.float32)).max() < 1e-6
    # (the masked case :1318-1328 is not a per-item identity: the reference leaves the masked item in the denominators
    #  with exp(log 1e-10); both restatements agree on it in test_list_mle_c_against_the_torch_restatement)

    def softmax(v):
        e = [math.exp(x) for x in v]
        return [x / sum(e) for x in e]
    scores = [[1., 3., 2.], [1., 2., 3.], [1., 2., 3.]]
    labels = [[0., 0., 1.], [0., 1., 2.], [0., 0., 0.]]
    loss, _ = c_ref.unique_softmax(scores, labels)
    want = [-ln(softmax(scores[0])[2]), -(ln(softmax(scores[1][:2])[1]) + 3. * ln(softmax(scores[1])[2])), 0.0]
    assert np.abs(loss - np.array(want, dtype=np.float32)).max() < 1e-6
    loss, _ = c_ref.unique_softmax([[1., 2., 3., 2.]], [[0., 1., 1., 0.]], mask=[[True, False, True, True]])
    assert abs(loss[0] + ln(softmax([1, 3, 2])[1])) < 1e-6


def _circle_py(labels, scores, gamma=64., margin=0.25):
    """losses_impl_test.py:32-87: the reference test suite's own helper (sum over preference pairs)."""
    tot = 0.
    for i in range(len(labels)):
        for j in range(len(labels)):
            if labels[i] > labels[j]:
                tot += math.exp(gamma * max(0., (1 + margin) - scores[i]) * ((1 - margin) - scores[i])
                                + gamma * max(0., scores[j] + margin) * (scores[j] - margin))
    return tot


def test_circle_c_known_answers_and_agreement():
    """oracle/listwise_c.c tfr_c_circle_f64: the reference's literals (losses_impl_test.py:1001-1083) and agreement
    with the torch restatement (fp64 run of it) on random batches, gradient included."""
    scores = [[0.1, 0.3, 0.2], [0.1, 0.2, 0.3]]
    labels = [[0., 0., 1.], [0., 1., 2.]]
    loss, has, _ = c_ref.circle(scores, labels)
    want = [math.log1p(_circle_py(labels[0], scores[0])), math.log1p(_circle_py(labels[1], scores[1]))]
    assert np.abs(loss - np.array(want, dtype=np.float32)).max() < 1e-5 * max(want) and has.all()
    loss, has, _ = c_ref.circle(scores, [[0., 0., 1.], [0., 1., 2.]], gamma=4., margin=0.1)
    want = [math.log1p(_circle_py(labels[0], scores[0], 4., 0.1)), math.log1p(_circle_py(labels[1], scores[1], 4., 0.1))]
    assert np.abs(loss - np.array(want, dtype=np.float32)).max() < 1e-6
    loss, has, _ = c_ref.circle([[.1, .3, .2]], [[0., -1., 1.]])                      # an invalid item
    assert abs(loss[0] - math.log1p(_circle_py([0., 1.], [.1, .2]))) < 1e-5 * loss[0]
    loss, has, grad = c_ref.circle([[.1, .3, .2]], [[1., 1., 1.]])                    # no preference pair
    assert loss[0] == 0.0 and not has[0] and not grad.any()
    for (B, L, gamma, margin, lo, hi) in ((6, 7, 64., 0.25, 0.2, 0.6), (9, 50, 4., 0.1, -0.3, 1.3), (4, 300, 16., 0.25, 0.0, 1.0)):
        lab, lg = make_batch(B, L, seed=4700 + L)
        lg = lo + (hi - lo) * torch.sigmoid(lg)                        # similarity scores, some outside [0, 1]
        lab[1] = -1.0
        oracle = R.CircleLoss(gamma=gamma, margin=margin)
        x = lg.double().clone().requires_grad_(True)
        want = oracle._compute_unreduced_loss_impl(lab.double(), oracle.get_logits(x))[0]
        want.sum().backward()
        loss, has, grad = c_ref.circle(lg.numpy(), lab.numpy(), gamma=gamma, margin=margin)
        w = want.detach().reshape(-1).numpy()
        assert np.abs(loss - w).max() <= 1e-6 * max(1.0, np.abs(w).max())
        assert np.abs(grad - x.grad.numpy()).max() <= 1e-5 * max(1.0, float(x.grad.abs().max()))
        assert not has[1] and has[0]
