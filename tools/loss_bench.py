#!/usr/bin/env python
"""Times every fused loss (forward + backward, one launch each) and every metric kernel of SURVEY 8f at one
shape, next to the CPU oracle on a bounded sample: the measurement row of the widened scope.
usage (through gpurun):  python tools/loss_bench.py [B] [L]     -> table on stdout (copied to profiles/)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                   # noqa: E402
from ranking_amd import _ops                   # noqa: E402
import ranking_amd as ra                       # noqa: E402
from oracle import tfr_ref as R                # noqa: E402  (CPU baseline column only)


def gpu_ms(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def cpu_lists_per_s(fn, n_lists, budget=2.0):
    fn()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < budget:
        fn(); n += 1
    return n * n_lists / (time.perf_counter() - t0)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    dev = 'cuda'
    labels, logits = bench.make_inputs(B, L, 0, dev)
    sim = torch.sigmoid(logits)                                        # Circle loss wants similarity scores
    nb = 256                                                           # CPU sample
    cl, cg = labels[:nb].cpu(), logits[:nb].cpu()
    Li, K = ra.losses_impl, ra.keras.losses
    lam = Li._lambda_kernel_args(K.NDCGLambdaWeight(), labels, L, torch.device(dev))
    tab = _ops.rank_table(lambda r: 1. / torch.log1p(r), L, torch.device(dev))
    div_labels = (torch.rand(B, L, 3, device=dev) < 0.3).float()
    div_labels[labels < 0] = -1.0

    def orc(loss):
        def f():
            lg = cg.clone().requires_grad_(True)
            loss.compute(cl, lg, None, R.Reduction.SUM).backward()
        return f
    rows = [
        ('ApproxNDCG', lambda: _ops.approx_ndcg(logits, labels), orc(R.ApproxNDCGLoss())),
        ('ApproxMRR', lambda: _ops.approx_mrr(logits, labels), orc(R.ApproxMRRLoss())),
        ('PairwiseLogistic + NDCG lambda', lambda: _ops.pairwise_logistic(logits, labels, want_aux=False, **lam),
         orc(R.PairwiseLogisticLoss(lambda_weight=R.NDCGLambdaWeight()))),
        ('PairwiseLogistic (no lambda)', lambda: _ops.pairwise_logistic(logits, labels, want_aux=False), orc(R.PairwiseLogisticLoss())),
        ('PairwiseHinge', lambda: _ops.pairwise_logistic(logits, labels, loss_kind=_ops.PAIR_HINGE), orc(R.PairwiseHingeLoss())),
        ('PairwiseSoftZeroOne', lambda: _ops.pairwise_logistic(logits, labels, loss_kind=_ops.PAIR_SOFT_ZERO_ONE), orc(R.PairwiseSoftZeroOneLoss())),
        ('PairwiseMSE', lambda: _ops.pairwise_logistic(logits, labels, loss_kind=_ops.PAIR_MSE), orc(R.PairwiseMSELoss())),
        ('Softmax', lambda: _ops.softmax_loss(logits, labels), orc(R.SoftmaxLoss())),
        ('PolyOneSoftmax', lambda: _ops.softmax_loss(logits, labels, poly_epsilon=1.0), orc(R.PolyOneSoftmaxLoss())),
        ('UniqueSoftmax', lambda: _ops.unique_softmax(logits, labels), orc(R.UniqueSoftmaxLoss())),
        ('ListMLE', lambda: _ops.list_mle(logits, labels), orc(R.ListMLELoss())),
        ('Circle', lambda: _ops.circle_loss(sim, labels), None),
        ('NeuralSortNDCG', lambda: _ops.neural_sort_loss(_ops.NEURAL_SORT_NDCG, logits, labels), orc(R.NeuralSortNDCGLoss())),
        ('NeuralSortCrossEntropy', lambda: _ops.neural_sort_loss(_ops.NEURAL_SORT_CE, logits, labels), orc(R.NeuralSortCrossEntropyLoss())),
        ('SigmoidCrossEntropy', lambda: _ops.pointwise_loss(_ops.POINT_SIGMOID_CE, logits, labels), orc(R.SigmoidCrossEntropyLoss())),
        ('MeanSquared', lambda: _ops.pointwise_loss(_ops.POINT_MSE, logits, labels), orc(R.MeanSquaredLoss())),
        ('GumbelSampler S=8 (fwd)', lambda: _ops.gumbel_sample(logits, labels, seed=1, sample_size=8), None),
        ('sort_ranks', lambda: _ops.sort_ranks(logits, None, labels >= 0, None, want_ranks=True, want_order=False), None),
        ('NDCG@{1,3,5,10,all}', lambda: ra.metrics_impl.NDCGMetric(None, None).compute_multi(labels, logits, None, None, [1, 3, 5, 10, None]),
         lambda: [R.NDCGMetric(topn=k).compute(cl, cg, None) for k in (1, 3, 5, 10, None)]),
        ('MRR', lambda: ra.metrics_impl.MRRMetric(None, None).compute(labels, logits), lambda: R.MRRMetric().compute(cl, cg, None)),
        ('MAP@10', lambda: ra.metrics_impl.MeanAveragePrecisionMetric(None, 10).compute(labels, logits),
         lambda: R.MeanAveragePrecisionMetric(topn=10).compute(cl, cg, None)),
        ('OPA', lambda: ra.metrics_impl.OPAMetric(None).compute(labels, logits), lambda: R.OPAMetric().compute(cl, cg, None)),
        ('BPref', lambda: ra.metrics_impl.BPrefMetric(None, None).compute(labels, logits), lambda: R.BPrefMetric().compute(cl, cg, None)),
        ('alpha-DCG@10 (3 subtopics)', lambda: ra.metrics_impl.AlphaDCGMetric(None, 10).compute(div_labels, logits),
         lambda: R.AlphaDCGMetric(topn=10).compute(div_labels[:nb].cpu(), cg, None)),
    ]
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    print('# B=%d L=%d (valid length U{L/2..L}); GPU: ms per call (fwd+bwd for losses, eager launches incl. the ordering'
          ' launches where automatic); CPU: oracle/tfr_ref.py on %d lists, %d threads' % (B, L, nb, torch.get_num_threads()))
    print('%-34s %10s %14s %14s %10s' % ('kernel path', 'GPU ms', 'GPU lists/s', 'CPU lists/s', 'ratio'))
    for name, g, c in rows:
        ms = gpu_ms(g)
        gl = B / (ms * 1e-3)
        cl_s = cpu_lists_per_s(c, nb) if c is not None else None
        print('%-34s %10.4f %14.3e %14s %10s' % (name, ms, gl, '%.3e' % cl_s if cl_s else '-',
                                                 '%.0fx' % (gl / cl_s) if cl_s else '-'))


if __name__ == '__main__':
    main()
