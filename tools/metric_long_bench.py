#!/usr/bin/env python
"""Times the long-list (workgroup-kernel) metrics: list_size 600 ... 4096, several batch sizes."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ranking_amd as ra  # noqa: E402
from ranking_amd.synthetic import make_batch  # noqa: E402


def main():
    mi = ra.metrics_impl
    for B, L in ((512, 1000), (1024, 1000), (2048, 1000), (4096, 1000), (4096, 600), (512, 3000)):
        labels, preds = make_batch(B, L, seed=5)
        lb, pr = labels.cuda(), preds.cuda()
        cases = (('NDCG@{1,3,5,10,all}', lambda: mi.NDCGMetric(None, None).compute_multi(lb, pr, None, None, [1, 3, 5, 10, None])),
                 ('NDCG@10', lambda: mi.NDCGMetric(None, 10).compute(lb, pr)),
                 ('MAP@{1,10,all}', lambda: mi.MeanAveragePrecisionMetric(None, None).compute_multi(lb, pr, None, None, [1, 10, None])),
                 ('MRR', lambda: mi.MRRMetric(None, None).compute(lb, pr)))
        for name, fn in cases:
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 20
            print('metric %-20s B=%-5d L=%-5d %.3f ms  %.3g lists/s' % (name, B, L, dt * 1e3, B / dt))


if __name__ == '__main__':
    main()
