#!/usr/bin/env python
"""Race screen + timing of the persistent tower GEMM's two k-loop forms (TFR_GEMM_PP=0 / 1; the switch is read once per
process, so every form runs in its own process and leaves fingerprints + timings under <out>):

    TFR_GEMM_PP=0 python tools/gemm_pp_check.py <out> a      # fingerprints of every form's outputs, REPS launches each
    TFR_GEMM_PP=1 python tools/gemm_pp_check.py <out> b
    python tools/gemm_pp_check.py <out> compare              # the two loops must agree BIT FOR BIT (same k order per
                                                             # accumulator) on every launch; prints the timing table

A stale stage / a buffer re-staged under a read shows up as a fingerprint that differs between launches or between the
two forms.  Forms: hidden-layer forward (BN + ReLU prologue, bias + statistics epilogue; Dropout 0 and 0.5, with the
written operand), its dgrad (ReLU-backward epilogue, Dropout 0 and 0.5), the plain product; M = 512 000 and 25 600."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REPS = 6


def fingerprint(t):
    v = t.contiguous().view(torch.int16 if t.dtype == torch.bfloat16 else torch.int32).reshape(-1).to(torch.int64)
    idx = torch.arange(v.numel(), device=v.device, dtype=torch.int64) % 65521 + 1
    return [int(v.sum().item()), int((v * idx).sum().item())]


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def run(out_dir, tag):
    from ranking_amd import _tower_ops as T
    dev = 'cuda'
    res = {'pp': os.environ.get('TFR_GEMM_PP', '(default)'), 'forms': {}}
    for M in (512000, 25600):
        g = torch.Generator(device=dev).manual_seed(1 + M)
        N = K = 512
        A = torch.randn((M, K), generator=g, device=dev).to(torch.bfloat16)
        W = (torch.randn((N, K), generator=g, device=dev) * 0.05).to(torch.bfloat16)
        Zp = torch.randn((M, N), generator=g, device=dev).to(torch.bfloat16)
        sc = torch.rand(K, generator=g, device=dev) + 0.5
        sh = torch.randn(K, generator=g, device=dev) * 0.1
        bias = torch.randn(N, generator=g, device=dev)
        mean = torch.randn(N, generator=g, device=dev) * 0.1
        rstd = torch.rand(N, generator=g, device=dev) + 0.5
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        aout = torch.empty((M, K), dtype=torch.bfloat16, device=dev)
        drop = T.Dropout.make(0.5, 12345)
        forms = {
            'plain': lambda: T.gemm(A, W, N, K, out=out),
            'forward': lambda: T.gemm(A, W, N, K, prologue=T.PRO_AFFINE_RELU, a_scale=sc, a_shift=sh, bias=bias,
                                      epilogue=T.EPI_STATS, out=out),
            'forward_drop': lambda: T.gemm(A, W, N, K, prologue=T.PRO_AFFINE_RELU, a_scale=sc, a_shift=sh, bias=bias,
                                           epilogue=T.EPI_STATS, out=out, pro_dropout=drop, a_out=aout),
            'dgrad': lambda: T.gemm(A, W, N, K, epilogue=T.EPI_RELU_BWD, Zp=Zp, e_scale=sc, e_shift=sh, e_mean=mean,
                                    e_rstd=rstd, out=out),
            'dgrad_drop': lambda: T.gemm(A, W, N, K, epilogue=T.EPI_RELU_BWD, Zp=Zp, e_scale=sc, e_shift=sh, e_mean=mean,
                                         e_rstd=rstd, out=out, epi_dropout=drop),
        }
        for name, fn in forms.items():
            fps = []
            for _ in range(REPS):
                out.zero_()
                C, stats = fn()
                fp = fingerprint(C)
                if stats is not None:
                    fp += fingerprint(stats)
                if name == 'forward_drop':
                    fp += fingerprint(aout)
                fps.append(fp)
            ms = timeit(fn)
            res['forms']['%s M=%d' % (name, M)] = {'fingerprints': fps, 'ms': ms,
                                                   'tflops': 2.0 * M * N * K / ms / 1e9}
            print('%-24s M=%-7d %8.3f ms %7.1f TFLOP/s  stable=%s' % (name, M, ms, 2.0 * M * N * K / ms / 1e9,
                                                                   all(f == fps[0] for f in fps)), flush=True)
    json.dump(res, open(os.path.join(out_dir, 'gemm_pp_%s.json' % tag), 'w'))


def compare(out_dir):
    runs = []
    for tag in 'abcdef':
        f = os.path.join(out_dir, 'gemm_pp_%s.json' % tag)
        if os.path.exists(f):
            runs.append(json.load(open(f)))
    ok = True
    print('%-26s' % 'form' + ''.join('  PP=%-9s' % r['pp'] for r in runs) + '  bits')
    for k in runs[0]['forms']:
        fps = [f for r in runs for f in r['forms'][k]['fingerprints']]
        same = all(f == fps[0] for f in fps)
        ok = ok and same
        print('%-26s' % k + ''.join('  %8.3f ms' % r['forms'][k]['ms'] for r in runs) + '  ' + ('identical' if same else 'DIFFERENT'))
    print('ALL IDENTICAL' if ok else 'MISMATCH')
    return 0 if ok else 1


if __name__ == '__main__':
    if sys.argv[2] == 'compare':
        sys.exit(compare(sys.argv[1]))
    run(sys.argv[1], sys.argv[2])
