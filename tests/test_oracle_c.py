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
