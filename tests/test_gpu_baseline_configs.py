"""The BASELINE.json configurations at their FULL sizes against fp64 arbiters (VERDICT r2 "Next round" #2).

north_star's bars: fp32 per-list loss within 1e-5, gradient within 1e-5 * max|g|, integer ranks / NDCG@k bit-exact.
Every check below
  * compares the HIP path with an fp64 evaluation (the plain-C arbiters of oracle/*.c where one exists, the torch
    restatement run in float64 otherwise) on the whole batch bench.py times -- not a slice;
  * evaluates the fp32 ORACLE (the op-for-op restatement of the reference, i.e. what the reference's own fp32 arithmetic
    achieves) against the same fp64 numbers, chunk by chunk;
  * records both maxima (tests/margins.py -> the "parity margins" section of the pytest log), and asserts the kernel
    within 1e-5 -- or, where fp32 itself cannot hold 1e-5 against fp64 (sums of ~n^2 / 2 terms at list_size 1000), within
    twice the fp32 oracle's own error, with that figure printed next to it.
"""
import math

import numpy as np
import pytest
import torch

from oracle import tfr_ref as R
from tests.common import make_batch
from tests.margins import record_margin

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def ra():
    import ranking_amd
    return ranking_amd


def _c():
    from oracle import c_ref
    try:
        c_ref.build()
    except Exception as e:  # pragma: no cover
        pytest.skip('plain-C arbiters unavailable: %s' % e)
    return c_ref


def _oracle32_chunks(fn, labels, logits, chunk, stride=1):
    """fp32 oracle per-list losses and gradients of sum(losses), `chunk` lists at a time (the [B, L, L] tensors), on
    every `stride`-th list (the context figure does not need all of them; the kernel check covers the whole batch)."""
    labels, logits = labels[::stride], logits[::stride]
    losses, grads = [], []
    for s in range(0, labels.shape[0], chunk):
        lg = logits[s:s + chunk].clone().requires_grad_(True)
        out = fn(labels[s:s + chunk], lg).reshape(-1)
        out.sum().backward()
        losses.append(out.detach()); grads.append(lg.grad)
    return torch.cat(losses), torch.cat(grads)


def _check(name, got_loss, got_grad, want_loss64, want_grad64, o32_loss, o32_grad, loss_bar=1e-5, grad_bar=1e-5, stride=1):
    """Kernel and fp32 oracle against fp64; bars relative to max(1, |loss|) and to max|g| of the batch."""
    w = torch.as_tensor(want_loss64, dtype=torch.float64).reshape(-1)
    g = torch.as_tensor(want_grad64, dtype=torch.float64)
    den = torch.clamp(w.abs(), min=1.0)
    e_k = ((got_loss.detach().cpu().double().reshape(-1) - w).abs() / den).max().item()
    e_o = ((o32_loss.double().reshape(-1) - w[::stride]).abs() / den[::stride]).max().item() if o32_loss is not None else float('nan')
    gs = g.abs().max().item()
    g_k = (got_grad.detach().cpu().double() - g).abs().max().item() / gs
    g_o = (o32_grad.double() - g[::stride]).abs().max().item() / gs if o32_grad is not None else float('nan')
    record_margin('%s: per-list loss, kernel vs fp64' % name, e_k, loss_bar, pin=True)
    record_margin('%s: gradient / max|g|, kernel vs fp64' % name, g_k, grad_bar, pin=True)
    if o32_loss is not None:
        record_margin('%s: per-list loss, fp32 ORACLE vs fp64 (context)' % name, e_o, loss_bar, pin=True)
        record_margin('%s: gradient / max|g|, fp32 ORACLE vs fp64 (context)' % name, g_o, grad_bar, pin=True)
    print('\n[%s] loss: kernel %.3e, fp32 oracle %.3e (bar %.0e) | grad/max|g|: kernel %.3e, fp32 oracle %.3e (bar %.0e)'
          % (name, e_k, e_o, loss_bar, g_k, g_o, grad_bar))
    return e_k, e_o, g_k, g_o


def test_headline_full_batch_1e5():
    """Row H: ApproxNDCGLoss(T = 0.1), 16384 x 200, seed 4 (the batch bench.py times), launch order on."""
    c = _c()
    from ranking_amd import _ops
    B, L = 16384, 200
    labels, logits = make_batch(B, L, seed=4)
    lb, lg = labels.to(DEV), logits.to(DEV)
    loss, _, d = _ops.approx_ndcg(lg, lb, None, None, 0.1, 0, True)
    w_loss, _, w_grad = c.approx_ndcg(logits.numpy(), labels.numpy(), temperature=0.1)
    o = R.ApproxNDCGLoss(temperature=0.1)
    o_loss, o_grad = _oracle32_chunks(lambda l, x: o._compute_unreduced_loss_impl(l, x / 0.1)[0], labels, logits, 256, stride=8)
    e_k, e_o, g_k, g_o = _check('headline ApproxNDCG 16384x200', loss, d, w_loss, w_grad, o_loss, o_grad, stride=8)
    assert e_k <= 1e-5 and g_k <= 1e-5, (e_k, g_k)
    assert bool((d[lb < 0] == 0).all())


def test_config2_full_batch_1e5():
    """Config 2 (loss part): SoftmaxLoss, 4096 x 100 and the 65 536 x 100 batch of bench.py's `softmax_hbm` line (the packed
    kernel): per-list loss, list weight and the gradient of weight * loss against the fp64 plain-C arbiter."""
    c = _c()
    from ranking_amd import _ops
    for B, L, tag in ((4096, 100, 'config 2 softmax 4096x100'), (65536, 100, 'softmax_hbm batch 65536x100 (packed kernel)')):
        labels, logits = make_batch(B, L, seed=4)
        lb, lg = labels.to(DEV), logits.to(DEV)
        loss, weight, d = _ops.softmax_loss(lg, lb, None, None, temperature=1.0, want_grad=True)
        w_loss, w_weight, w_grad = c.softmax(logits.numpy(), labels.numpy(), temperature=1.0)
        want_grad = torch.from_numpy(w_grad).double() * torch.from_numpy(w_weight).double().unsqueeze(1)   # kernel: d(weight * loss)
        e_k, _, g_k, _ = _check(tag, loss, d, w_loss, want_grad, None, None)
        assert e_k <= 1e-5 and g_k <= 1e-5, (tag, e_k, g_k)
        assert (weight.cpu().double() - torch.from_numpy(w_weight).double()).abs().max().item() <= 1e-6 * max(1.0, float(w_weight.max()))


def test_config3_full_batch_1e5():
    """Config 3: PairwiseLogisticLoss + NDCGLambdaWeight, 4096 x 200: per-list sums of w_ij * loss_ij and gradient."""
    c = _c()
    from ranking_amd import _ops
    K = ra().keras.losses
    B, L = 4096, 200
    labels, logits = make_batch(B, L, seed=4)
    lb, lg = labels.to(DEV), logits.to(DEV)
    lam = ra().losses_impl._lambda_kernel_args(K.NDCGLambdaWeight(), lb, L, torch.device(DEV))
    ones = torch.ones(B, device=DEV)
    _, _, _, d, lst = _ops.pairwise_logistic(lg, lb, None, None, ones, want_rows=False, want_aux=False, want_list=True, **lam)
    w_out, w_grad = c.pairwise_logistic_ndcg(logits.numpy(), labels.numpy())
    o = R.PairwiseLogisticLoss(lambda_weight=R.NDCGLambdaWeight())

    def per_list(l, x):
        losses, w = o._compute_unreduced_loss_impl(l, x, l >= 0)
        return (losses * w).sum(dim=(1, 2))
    o_loss, o_grad = _oracle32_chunks(per_list, labels, logits, 128, stride=4)
    e_k, e_o, g_k, g_o = _check('config 3 pairwise + NDCG lambda 4096x200', lst, d, w_out, w_grad, o_loss, o_grad, stride=4)
    assert e_k <= 1e-5 and g_k <= 1e-5, (e_k, g_k)


def test_config4_full_batch():
    """Config 4 shard: ApproxNDCGLoss(T = 0.1), 512 x 1000 -- ALL 512 lists (round 2 checked four)."""
    c = _c()
    from ranking_amd import _ops
    B, L = 512, 1000
    labels, logits = make_batch(B, L, seed=5)
    lb, lg = labels.to(DEV), logits.to(DEV)
    loss, _, d = _ops.approx_ndcg(lg, lb, None, None, 0.1)
    w_loss, _, w_grad = c.approx_ndcg(logits.numpy(), labels.numpy(), temperature=0.1)
    o = R.ApproxNDCGLoss(temperature=0.1)
    o_loss, o_grad = _oracle32_chunks(lambda l, x: o._compute_unreduced_loss_impl(l, x / 0.1)[0], labels, logits, 16, stride=4)
    e_k, e_o, g_k, g_o = _check('config 4 ApproxNDCG 512x1000', loss, d, w_loss, w_grad, o_loss, o_grad, stride=4)
    # north_star's bar, flat (round 6: measured 1.8e-7 / 6.1e-7; the `max(1e-5, 2 x oracle error)` relaxation was never needed)
    assert e_k <= 1e-5 and g_k <= 1e-5, (e_k, e_o, g_k, g_o)


def test_config5_full_batch_gumbel_injected_uniform():
    """Config 5 loss: GumbelApproxNDCGLoss(S = 8, T_g = 1, T = 0.1), 512 x 50, the uniform draws injected: Keras AUTO
    value and the gradient w.r.t. the ORIGINAL logits (through the sampler) against the restatement run in float64."""
    B, L, S = 512, 50, 8
    labels, logits = make_batch(B, L, seed=6)
    u = torch.rand((B, S, L), generator=torch.Generator().manual_seed(66))
    K = ra().keras.losses
    mine = K.GumbelApproxNDCGLoss(sample_size=S, gumbel_temperature=1.0, temperature=0.1, seed=3)
    value, d = mine.loss_and_grad(labels.to(DEV), logits.to(DEV), None, uniform=u.to(DEV))

    def ref(dtype):
        lg = logits.to(dtype).clone().requires_grad_(True)
        v = R.keras_loss_call(R.ApproxNDCGLoss(temperature=0.1), labels.to(dtype), lg,
                              gumbel_sampler=R.GumbelSampler(sample_size=S, temperature=1.0), uniform=u.to(dtype))
        v.backward()
        return v.detach(), lg.grad
    v64, g64 = ref(torch.float64)
    v32, g32 = ref(torch.float32)
    e_k, e_o, g_k, g_o = _check('config 5 GumbelApproxNDCG 512x50xS8 (batch mean)', value.reshape(1), d, v64.reshape(1), g64,
                                v32.reshape(1), g32)
    assert e_k <= 1e-5 and g_k <= max(1e-5, 2 * g_o), (e_k, g_k, g_o)


def test_ndcg_at_10_bit_exact_on_the_whole_headline_batch():
    """NDCG@10 of all 16384 lists bit-equal to the oracle (round 2: a 256-list slice), MRR@10 too, and both within
    5e-6 / 1e-6 of the independent fp64 C arbiter."""
    c = _c()
    B, L = 16384, 200
    labels, logits = make_batch(B, L, seed=4)
    mi = ra().metrics_impl
    ndcg, _ = mi.NDCGMetric(None, 10).compute(labels.to(DEV), logits.to(DEV))
    mrr, _ = mi.MRRMetric(None, 10).compute(labels.to(DEV), logits.to(DEV))
    want_n, _ = R.NDCGMetric(topn=10).compute(labels, logits)
    want_m, _ = R.MRRMetric(topn=10).compute(labels, logits)
    assert torch.equal(ndcg.cpu().reshape(-1), want_n.reshape(-1))
    assert torch.equal(mrr.cpu().reshape(-1), want_m.reshape(-1))
    w_ndcg, w_mrr = c.ndcg_mrr(logits.numpy(), labels.numpy(), topn=10)
    e_n = float(np.abs(ndcg.cpu().double().numpy().reshape(-1) - w_ndcg.reshape(-1)).max())
    e_m = float(np.abs(mrr.cpu().double().numpy().reshape(-1) - w_mrr.reshape(-1)).max())
    record_margin('headline NDCG@10 (16384 lists) vs fp64 C', e_n, 5e-6, pin=True)
    record_margin('headline MRR@10 (16384 lists) vs fp64 C', e_m, 1e-6, pin=True)
    assert e_n <= 5e-6 and e_m <= 1e-6, (e_n, e_m)
