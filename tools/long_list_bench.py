#!/usr/bin/env python3
"""Times the entry points whose list-size ceiling went to 8192 in round 5, just below and above the size at which their
working arrays leave LDS for the caller's workspace (ListMLE / UniqueSoftmax / Circle / rank + diversity metrics: 4096;
NeuralSort: 2048, where the wave kernel hands over to the workgroup form).  HIP-event timing of eager launches, median of
`--iters`; prints one line per (op, list size): ms per launch and microseconds per list.  Developer / evidence tool
(profiles/r05_long_lists.txt); nothing here is on a bench.py line."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ranking_amd as ra                      # noqa: E402
from ranking_amd import _ops                  # noqa: E402


def timed(fn, iters):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--lists', type=int, default=256)
    ap.add_argument('--iters', type=int, default=7)
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    g = torch.Generator().manual_seed(5)
    mi = ra.metrics_impl
    print('%-34s %6s %6s %12s %12s' % ('op', 'lists', 'L', 'ms/launch', 'us/list'))
    for L in (2048, 2049, 4096, 4097, 8192):
        B = args.lists
        labels = torch.randint(0, 5, (B, L), generator=g).float()
        n = torch.randint(L // 2, L + 1, (B, 1), generator=g)
        labels = torch.where(torch.arange(L).unsqueeze(0) < n, labels, torch.full_like(labels, -1.0)).to(dev)
        logits = torch.randn((B, L), generator=g).to(dev)
        sub = (torch.rand((B, L, 2), generator=g) < 0.3).float().to(dev)
        disc = _ops.rank_table(lambda r: 1. / torch.log1p(r), L, dev)
        ops = [
            ('list_mle fwd+bwd', B, lambda: _ops.list_mle(logits, labels)),
            ('unique_softmax fwd+bwd', B, lambda: _ops.unique_softmax(logits, labels)),
            ('circle fwd+bwd', B, lambda: _ops.circle_loss(torch.sigmoid(logits), labels)),
            ('MAP@{10,all}', B, lambda: mi.MeanAveragePrecisionMetric(None, None).compute_multi(labels, logits, None, None, [10, None])),
            ('alpha-DCG@{10,all} (S=2)', B, lambda: _ops.div_metric(_ops.DIV_ALPHA_DCG, sub, logits, None, None, [10, None], discount=disc)),
        ]
        nb = max(1, min(B, 16))                           # NeuralSort is O(L^2) per list: a few lists
        ops += [
            ('neural_sort NDCG fwd+bwd', nb, lambda: _ops.neural_sort_loss(_ops.NEURAL_SORT_NDCG, logits[:nb], labels[:nb])),
            ('neural_sort CE fwd+bwd', nb, lambda: _ops.neural_sort_loss(_ops.NEURAL_SORT_CE, logits[:nb], labels[:nb])),
        ]
        for name, nl, fn in ops:
            if name.startswith('neural_sort') and L in (4096, 4097):
                continue
            if not name.startswith('neural_sort') and L in (2048, 2049):
                continue
            ms = timed(fn, args.iters)
            print('%-34s %6d %6d %12.4f %12.2f' % (name, nl, L, ms, 1e3 * ms / nl))


if __name__ == '__main__':
    main()
