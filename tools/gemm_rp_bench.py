#!/usr/bin/env python
"""Same-process A/B of the tower GEMM forms: round-5 kernels (TFR_GEMM_RP=0) against the resident-panel kernel (TFR_GEMM_RP=1).
    M=512000 python tools/gemm_rp_bench.py"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ranking_amd import _tower_ops as t  # noqa: E402


def timeit(fn, iters=30, warm=5):
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
    N = K = 512
    for M in [int(x) for x in os.environ.get('M', '512000,409600').split(',')]:
        A = torch.randn((M, K), device=dev).to(torch.bfloat16)
        W = (torch.randn((N, K), device=dev) * 0.05).to(torch.bfloat16)
        Zp = torch.randn((M, N), device=dev).to(torch.bfloat16)
        bias = torch.randn(N, device=dev)
        sc = torch.rand(K, device=dev) + 0.5; sh = torch.randn(K, device=dev) * 0.1
        es = torch.rand(N, device=dev) + 0.5; eh = torch.randn(N, device=dev) * 0.1
        em = torch.randn(N, device=dev) * 0.1; er = torch.rand(N, device=dev) + 0.5
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        aout = torch.empty((M, K), dtype=torch.bfloat16, device=dev)
        d = t.Dropout.make(0.5, 7)
        forms = [
            ('plain', lambda: t.gemm(A, W, N, K, out=out)),
            ('bias + stats', lambda: t.gemm(A, W, N, K, bias=bias, epilogue=t.EPI_STATS, out=out)),
            ('forward (BN+ReLU, stats)', lambda: t.gemm(A, W, N, K, prologue=2, a_scale=sc, a_shift=sh, bias=bias, epilogue=t.EPI_STATS, out=out)),
            ('forward + Dropout 0.5 + a_out', lambda: t.gemm(A, W, N, K, prologue=2, a_scale=sc, a_shift=sh, bias=bias, epilogue=t.EPI_STATS,
                                                            out=out, pro_dropout=d, a_out=aout)),
            ('dgrad (ReLU bwd)', lambda: t.gemm(A, W, N, K, epilogue=t.EPI_RELU_BWD, Zp=Zp, e_scale=es, e_shift=eh, e_mean=em, e_rstd=er, out=out)),
            ('dgrad + Dropout', lambda: t.gemm(A, W, N, K, epilogue=t.EPI_RELU_BWD, Zp=Zp, e_scale=es, e_shift=eh, e_mean=em, e_rstd=er, out=out,
                                               epi_dropout=d)),
        ]
        flops = 2.0 * M * N * K
        for name, fn in forms:
            row = []
            for rp in ('0', '1', '0', '1'):
                os.environ['TFR_GEMM_RP'] = rp
                row.append(timeit(fn))
            os.environ['TFR_GEMM_RP'] = '0'
            os.environ['TFR_GEMM_BS'] = '1'
            bs = [timeit(fn), timeit(fn)]
            os.environ['TFR_GEMM_BS'] = '0'
            print('      weight-stationary (TFR_GEMM_BS=1; only the forms without a prologue take it): %.4f / %.4f ms  (%.0f TFLOP/s)' % (
                bs[0], bs[1], flops / min(bs) / 1e9))
            os.environ['TFR_GEMM_RP_ROT'] = '0'
            norot = timeit(fn)
            os.environ.pop('TFR_GEMM_RP_ROT')
            stag = []
            for u in os.environ.get('STAGGERS', '0,4').split(','):
                os.environ['TFR_GEMM_RP_STAGGER'] = u
                stag.append('%s: %.4f' % (u, timeit(fn)))
            os.environ.pop('TFR_GEMM_RP_STAGGER')
            print('      stagger units -> ms   ' + '   '.join(stag))
            print('M=%d %-32s RP=0 %.4f / %.4f ms   RP=1 %.4f / %.4f ms [same k order: %.4f]  (%.0f -> %.0f TFLOP/s, x%.2f)' % (
                M, name, row[0], row[2], row[1], row[3], norot, flops / min(row[0], row[2]) / 1e9, flops / min(row[1], row[3]) / 1e9,
                min(row[0], row[2]) / min(row[1], row[3])), flush=True)


if __name__ == '__main__':
    main()
