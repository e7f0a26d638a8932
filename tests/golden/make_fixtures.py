#!/usr/bin/env python
"""Generates tests/golden/oracle_fixtures.json: seeded small inputs and the CPU
oracle's outputs for every hot-path row.  The reference itself
cannot be imported (TensorFlow is absent), so these vectors come from the oracle
AFTER it has been pinned to the reference's literals (tests/test_oracle_golden.py);
they freeze the oracle (CPU test) and give the GPU tests committed targets.

    python tests/golden/make_fixtures.py
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import tfr_ref as R          # noqa: E402
from tests.common import make_batch, make_weights   # noqa: E402


def grad_of(fn, logits):
    lg = logits.clone().requires_grad_(True)
    out = fn(lg)
    out.sum().backward()
    return out.detach(), lg.grad


def build():
    fx = {}
    B, L, S = 5, 9, 3
    labels, logits = make_batch(B, L, seed=2024)
    labels[3] = torch.where(labels[3] >= 0, torch.zeros_like(labels[3]), labels[3])   # no relevant item
    w_item = make_weights(B, L, seed=1)
    w_list = make_weights(B, 1, seed=2)
    u = torch.rand((B, S, L), generator=torch.Generator().manual_seed(3))
    fx['inputs'] = dict(labels=labels.tolist(), logits=logits.tolist(), item_weights=w_item.tolist(),
                        list_weights=w_list.tolist(), uniform=u.tolist(), sample_size=S)
    mask = labels >= 0
    fx['ranks'] = R._compute_ranks(logits, mask).tolist()
    fx['order'] = R.sort_by_scores(logits, [torch.arange(L).expand(B, L)], mask=mask)[0].tolist()
    for k in (1, 3, 10, None):
        v, w = R.NDCGMetric(topn=k).compute(labels, logits, w_item)
        fx['ndcg_weighted@%s' % k] = dict(value=v.reshape(-1).tolist(), weight=w.reshape(-1).tolist())
        v, w = R.NDCGMetric(topn=k).compute(labels, logits)
        fx['ndcg@%s' % k] = dict(value=v.reshape(-1).tolist(), weight=w.reshape(-1).tolist())
        v, w = R.MRRMetric(topn=k).compute(labels, logits)
        fx['mrr@%s' % k] = dict(value=v.reshape(-1).tolist(), weight=w.reshape(-1).tolist())
    o = R.ApproxNDCGLoss(temperature=0.1)
    l, g = grad_of(lambda lg: o._compute_unreduced_loss_impl(labels, lg / 0.1)[0], logits)
    fx['approx_ndcg'] = dict(loss=l.reshape(-1).tolist(), dlogits=g.tolist(),
                             weight=o._compute_unreduced_loss_impl(labels, logits / 0.1)[1].reshape(-1).tolist())
    for name, lam in (('none', None), ('ndcg', R.NDCGLambdaWeight()),
                      ('ndcg_top3_smooth', R.NDCGLambdaWeight(topn=3, smooth_fraction=0.4))):
        o = R.PairwiseLogisticLoss(lambda_weight=lam, temperature=1.0)

        def rows(lg):
            losses, w = o._compute_unreduced_loss_impl(labels, lg, mask)
            return (losses * w * o._normalize_weights_impl(labels, w_item)).sum(dim=2)
        l, g = grad_of(rows, logits)
        fx['pairwise_%s' % name] = dict(row_loss=l.tolist(), dlogits=g.tolist())
    o = R.SoftmaxLoss(temperature=1.0)
    lg = logits.clone().requires_grad_(True)
    sl, sw = o.compute_per_list(labels, lg, w_list)
    (sl * sw).sum().backward()
    fx['softmax'] = dict(loss=sl.tolist(), weight=sw.tolist(), dlogits=lg.grad.tolist())
    gl, gs, _ = R.GumbelSampler(sample_size=S, temperature=1.0).sample(labels, logits, None, uniform=u)
    fx['gumbel'] = dict(sampled=gs.tolist())
    fx['keras'] = dict(
        approx_ndcg=R.keras_loss_call(R.ApproxNDCGLoss(), labels, logits, w_list).item(),
        pairwise_ndcg_lambda=R.keras_loss_call(R.PairwiseLogisticLoss(lambda_weight=R.NDCGLambdaWeight()),
                                               labels, logits, w_list).item(),
        softmax=R.keras_loss_call(R.SoftmaxLoss(), labels, logits, w_list).item(),
        gumbel_approx_ndcg=R.keras_loss_call(R.ApproxNDCGLoss(), labels, logits, w_list,
                                             gumbel_sampler=R.GumbelSampler(sample_size=S), uniform=u).item())
    return fx


if __name__ == '__main__':
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'oracle_fixtures.json')
    with open(path, 'w') as f:
        json.dump(build(), f, indent=1)
    print('wrote', path)
