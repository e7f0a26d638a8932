#!/usr/bin/env python
"""Micro-benchmark of the tower GEMM kernel against torch (hipBLASLt) on the scorer shapes."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ranking_amd import _tower_ops as t  # noqa: E402


def timeit(fn, iters=20, warm=3):
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


def main():
    dev = 'cuda'
    M = int(os.environ.get('M', 409600))
    for (N, K) in [(512, 512), (512, 136)]:
        A = torch.randn((M, K), device=dev).to(torch.bfloat16)
        W = (torch.randn((N, K), device=dev) * 0.05).to(torch.bfloat16)
        bias = torch.randn(N, device=dev)
        sc = torch.rand(K, device=dev) + 0.5
        sh = torch.randn(K, device=dev) * 0.1
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        flops = 2.0 * M * N * K
        byts = 2.0 * (M * K + M * N + N * K)
        for name, fn in [
            ('torch bf16 matmul', lambda: torch.matmul(A, W.t(), out=out)),
            ('tower plain', lambda: t.gemm(A, W, N, K, out=out)),
            ('tower bias+stats', lambda: t.gemm(A, W, N, K, bias=bias, epilogue=t.EPI_STATS, out=out)),
            ('tower bnrelu+bias+stats', lambda: t.gemm(A, W, N, K, prologue=2, a_scale=sc, a_shift=sh, bias=bias,
                                                       epilogue=t.EPI_STATS, out=out)),
        ]:
            ms = timeit(fn)
            print('M=%d N=%d K=%d %-26s %8.3f ms  %7.1f TFLOP/s  %6.0f GB/s' % (
                M, N, K, name, ms, flops / ms / 1e9, byts / ms / 1e6))
    z = torch.randn((M, 512), device=dev).to(torch.bfloat16)
    w = torch.randn((1, 512), device=dev) * 0.05
    b = torch.zeros(1, device=dev)
    sc = torch.rand(512, device=dev) + 0.5; sh = torch.randn(512, device=dev) * 0.1
    ms = timeit(lambda: t.out_layer(z, 512, 2, sc, sh, w, b))
    print('out layer M=%d K=512: %.3f ms %.0f GB/s' % (M, ms, M * 512 * 2 / ms / 1e6))
    x = torch.randn((M, 136), device=dev)
    ms = timeit(lambda: t.cast_rows(x))
    print('cast rows M=%d F=136: %.3f ms %.0f GB/s' % (M, ms, M * 136 * 6 / ms / 1e6))


if __name__ == '__main__':
    main()
