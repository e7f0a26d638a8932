#!/usr/bin/env python
"""Times the fp32 Dense products (csrc/gemm_f32.hip) at BASELINE config 2's row count (4096 lists x 100 items):
HIP events around 10 launches each, after 3 warm-up launches.  Output -> profiles/r03_gemm_f32.txt."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ranking_amd import _tower_ops as T

dev = 'cuda'
M = 409600


def t(fn, n=10):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


print('# fp32 Dense on v_mfma_f32_32x32x2_f32 (peak 157.3 TFLOP/s), M = %d rows; ms per launch, TFLOP/s' % M)
SHAPES = ((512, 512),) if os.environ.get('GEMM_QUICK') else ((512, 136), (512, 512), (1, 512))
for N, K in SHAPES:
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev)
    b = torch.randn(N, device=dev)
    dy = torch.randn(M, N, device=dev)
    fl = 2.0 * M * N * K
    for name, fn in (('forward  y = x W^T + b', lambda: T.dense_f32(x, w, b)),
                     ('dgrad    dx = dy W', lambda: T.dense_f32_dgrad(dy, w)),
                     ('wgrad    dW = dy^T x', lambda: T.dense_f32_wgrad(dy, x)),
                     ('bias     db = colsum(dy)', lambda: T.colsum_f32(dy)),
                     ('torch    F.linear (library GEMM, for reference)', lambda: torch.nn.functional.linear(x, w, b)))[:3 if os.environ.get('GEMM_QUICK') else 5]:
        ms = t(fn)
        print('N = %3d K = %3d  %-50s %8.3f ms  %7.1f TFLOP/s' % (N, K, name, ms, 0.0 if 'bias' in name else fl / ms / 1e9))
