#!/usr/bin/env python
"""Developer aid / test helper: outputs of the LambdaRank group kernel on a fixed set of cases, written to a file.
tests/test_gpu_parity.py runs it twice (TFR_LAMBDARANK_GRADED=1 / 0, read once per process by the library) and compares
the files bit for bit: the graded builder of round 6 (lambdarank_group.h: grp_build_graded) against the general one.
Usage: python tools/lgraded_check.py <out.pt>"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ranking_amd as ra  # noqa: E402
from ranking_amd import _ops  # noqa: E402
from ranking_amd.synthetic import make_batch  # noqa: E402

DEV = 'cuda'


def cases():
    g = torch.Generator().manual_seed(11)
    out = []
    for B, L, seed in ((4096, 200, 1), (1024, 100, 2), (777, 37, 3), (600, 255, 4), (512, 64, 5), (520, 130, 6), (512, 256, 7)):
        labels, logits = make_batch(B, L, seed=seed)
        out.append(('synthetic %dx%d' % (B, L), labels, logits, {}))
    labels, logits = make_batch(1024, 200, seed=8)
    labels[0] = -1.0                                            # no valid item
    labels[1] = torch.where(labels[1] >= 0, torch.full_like(labels[1], 3.0), labels[1])      # one grade
    labels[2, 5:] = -1.0                                        # five items
    labels[3] = torch.where(labels[3] >= 0, torch.randint(0, 13, labels[3].shape, generator=g).float(), labels[3])   # 13 grades: tail segment
    labels[4] = torch.where(labels[4] >= 0, labels[4] + 0.5, labels[4])                      # not integers
    labels[5] = torch.where(labels[5] >= 0, torch.randint(0, 8, labels[5].shape, generator=g).float(), labels[5])    # exactly 8 grades
    labels[6] = torch.where(labels[6] >= 0, torch.randint(0, 9, labels[6].shape, generator=g).float(), labels[6])    # 9 grades
    labels[7] = torch.where(labels[7] >= 0, labels[7] * 7 + 3, labels[7])                    # grades 3 .. 31
    labels[8] = torch.where(labels[8] >= 0, labels[8] * 8 + 3, labels[8])                    # a grade of 35: not "small"
    logits[9] = torch.round(logits[9] * 2) / 2                                               # tied scores
    logits[10] = logits[10] * 60                                                             # range > 80: the per-pair exp body
    logits[11, 7] = 500.0                                                                    # an outlier
    out.append(('edge lists', labels, logits, {}))
    labels, logits = make_batch(2048, 200, seed=9)
    out.append(('list weights, T = 0.7', labels, logits, dict(weights=True, temperature=0.7)))
    out.append(('DCG (not normalised), identity gain', labels, logits, dict(dcg=True)))
    return out


def main():
    k = ra.keras.losses
    res = {}
    for name, labels, logits, opt in cases():
        lb, lg = labels.to(DEV), logits.to(DEV)
        B, L = lb.shape
        if opt.get('dcg'):
            lw_obj = ra.losses_impl.DCGLambdaWeight()
        else:
            lw_obj = k.NDCGLambdaWeight()
        lam = ra.losses_impl._lambda_kernel_args(lw_obj, lb, L, torch.device(DEV))
        w = (torch.rand(B, generator=torch.Generator().manual_seed(3)) + 0.5).to(DEV) if opt.get('weights') else None
        outs = _ops.pairwise_logistic(lg, lb, list_weights=w, temperature=opt.get('temperature', 1.0), want_aux=False,
                                      want_rows=True, want_list=True, **lam)
        res[name] = [None if t is None else t.cpu() for t in outs]
    torch.save(res, sys.argv[1])
    print('wrote %d cases to %s' % (len(res), sys.argv[1]))


if __name__ == '__main__':
    main()
