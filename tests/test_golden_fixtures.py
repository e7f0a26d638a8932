"""tests/golden/oracle_fixtures.json: (CPU) the oracle still reproduces it;
(GPU) the HIP path matches the committed vectors."""
import json
import os

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
FX = json.load(open(os.path.join(HERE, 'golden', 'oracle_fixtures.json')))
T = lambda x, **kw: torch.tensor(x, **kw)


def _flat_close(a, b, tol):
    a = torch.as_tensor(a, dtype=torch.float64).reshape(-1)
    b = torch.as_tensor(b, dtype=torch.float64).reshape(-1)
    assert a.shape == b.shape
    assert (a - b).abs().max().item() <= tol * max(1.0, b.abs().max().item()), (a - b).abs().max()


def test_oracle_reproduces_fixtures():
    import importlib.util
    spec = importlib.util.spec_from_file_location('mk', os.path.join(HERE, 'golden', 'make_fixtures.py'))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    now = mk.build()

    def walk(a, b, path=''):
        if isinstance(a, dict):
            assert a.keys() == b.keys(), path
            for k in a:
                walk(a[k], b[k], path + '/' + k)
        elif isinstance(a, (list, float)):
            _flat_close(a, b, 1e-6)
        else:
            assert a == b, path
    walk(now, FX)


@pytest.mark.gpu
def test_hip_path_matches_fixtures():
    import ranking_amd as ra
    from ranking_amd import _ops
    dev = 'cuda'
    inp = FX['inputs']
    labels, logits = T(inp['labels'], device=dev), T(inp['logits'], device=dev)
    w_item, w_list = T(inp['item_weights'], device=dev), T(inp['list_weights'], device=dev)
    u = T(inp['uniform'], device=dev)
    S = inp['sample_size']
    ranks, order = _ops.sort_ranks(logits, labels, None, None)
    assert ranks.cpu().tolist() == FX['ranks'] and order.cpu().tolist() == FX['order']
    mi = ra.metrics_impl
    for k in (1, 3, 10, None):
        v, w = mi.NDCGMetric(None, k).compute(labels, logits, w_item)
        assert v.reshape(-1).cpu().tolist() == FX['ndcg_weighted@%s' % k]['value']       # bit-exact
        _flat_close(w.cpu(), FX['ndcg_weighted@%s' % k]['weight'], 1e-6)
        v, w = mi.NDCGMetric(None, k).compute(labels, logits)
        assert v.reshape(-1).cpu().tolist() == FX['ndcg@%s' % k]['value']
        v, w = mi.MRRMetric(None, k).compute(labels, logits)
        assert v.reshape(-1).cpu().tolist() == FX['mrr@%s' % k]['value']
        _flat_close(w.cpu(), FX['mrr@%s' % k]['weight'], 1e-6)
    loss, weight, d = _ops.approx_ndcg(logits, labels, None, None, 0.1)
    _flat_close(loss.cpu(), FX['approx_ndcg']['loss'], 1e-5)
    assert weight.cpu().tolist() == FX['approx_ndcg']['weight']
    _flat_close(d.cpu(), FX['approx_ndcg']['dlogits'], 1e-5)
    K = ra.keras.losses
    for name, lam in (('none', None), ('ndcg', K.NDCGLambdaWeight()),
                      ('ndcg_top3_smooth', K.NDCGLambdaWeight(topn=3, smooth_fraction=0.4))):
        loss_obj = ra.losses_impl.PairwiseLogisticLoss(None, lambda_weight=lam)
        lg = logits.clone().requires_grad_(True)
        list_loss, row_loss, _, _ = loss_obj._fused(labels, lg, w_item, None)
        list_loss.sum().backward()
        _flat_close(row_loss.cpu(), FX['pairwise_%s' % name]['row_loss'], 1e-5)
        _flat_close(lg.grad.cpu(), FX['pairwise_%s' % name]['dlogits'], 1e-5)
    lg = logits.clone().requires_grad_(True)
    sl, sw = ra.losses_impl.SoftmaxLoss(None).compute_per_list(labels, lg, w_list)
    (sl * sw).sum().backward()
    _flat_close(sl.cpu(), FX['softmax']['loss'], 1e-5)
    _flat_close(sw.cpu(), FX['softmax']['weight'], 1e-6)
    _flat_close(lg.grad.cpu(), FX['softmax']['dlogits'], 1e-5)
    gs = _ops.gumbel_sample(logits, labels, None, u, 0, 0, S, 1.0)
    _flat_close(gs.cpu(), FX['gumbel']['sampled'], 1e-5)
    k = FX['keras']
    assert abs(K.ApproxNDCGLoss()(labels, logits, w_list).item() - k['approx_ndcg']) < 1e-5
    assert abs(K.PairwiseLogisticLoss(lambda_weight=K.NDCGLambdaWeight())(labels, logits, w_list).item()
               - k['pairwise_ndcg_lambda']) < 1e-5 * max(1., abs(k['pairwise_ndcg_lambda']))
    assert abs(K.SoftmaxLoss()(labels, logits, w_list).item() - k['softmax']) < 1e-5 * max(1., abs(k['softmax']))
    assert abs(K.GumbelApproxNDCGLoss(sample_size=S)(labels, logits, w_list, uniform=u).item()
               - k['gumbel_approx_ndcg']) < 1e-5
