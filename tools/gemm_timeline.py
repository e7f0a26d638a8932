#!/usr/bin/env python
"""Developer aid: per-workgroup timeline of the 256 x 256 tower GEMM from in-kernel s_memtime stamps
(tower.hip built with -DTFR_GEMM_ABLATE=16 | 48 by `build`): phase durations and the idle gap between consecutive
workgroups on one CU.  Config-2 hidden-layer shape (M = 409600, N = K = 512)."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, 'tools', '_ablate')
SRC = os.path.join(ROOT, 'ranking_amd', 'csrc', 'tower.hip')
MASKS = [int(m) for m in os.environ.get('MASKS', '16').split(',')]


def lib_path(mask):
    return os.path.join(OUT, 'libtower_tl%d.so' % mask)


def build():
    os.makedirs(OUT, exist_ok=True)
    procs = [subprocess.Popen(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC',
                               '-DTFR_GEMM_ABLATE=%d' % m, SRC, '-o', lib_path(m)]) for m in MASKS]
    assert all(p.wait() == 0 for p in procs)


def run():
    import torch
    dev = 'cuda'
    M, N, K = 409600, 512, 512
    A = torch.randn((M, K), device=dev).to(torch.bfloat16)
    W = (torch.randn((N, K), device=dev) * 0.05).to(torch.bfloat16)
    C = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    Zp = torch.randn((M, N), device=dev).to(torch.bfloat16)
    vec = lambda v: torch.full((max(N, K),), v, device=dev)
    sc, sh, mean, rstd, bias = vec(1.0), vec(0.1), vec(0.0), vec(1.0), vec(0.01)
    stats = torch.zeros(((M + 63) // 64, 2, N), device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    n_wg = ((M + 255) // 256 + 7) // 8 * 8 * (N // 256)
    for mask in MASKS:
        lib = ctypes.CDLL(lib_path(mask))
        f = lib.tfr_tower_gemm_bf16
        f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long,
                      ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                      ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long] + \
            [ctypes.c_void_p] * 7
        buf = torch.zeros((n_wg, 8), dtype=torch.int64, device=dev)
        lib.tfr_prof_set_buffer_gemm(ctypes.c_void_p(buf.data_ptr()))
        for name, pro, epi in (('fwd hidden (2, 1)', 2, 1), ('plain (0, 0)', 0, 0), ('dgrad (0, 2)', 0, 2)):
            call = lambda: f(p(A), K, p(W), K, p(C), N, M, N, K, pro, p(sc), p(sh), (None if epi == 2 else p(bias)), epi, p(stats), p(Zp), N,
                             p(sc), p(sh), p(mean), p(rstd), None, None, st)
            for _ in range(3):
                assert call() == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); call(); e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3
            d = buf.cpu()
            os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
            import numpy as np
            np.save(os.path.join(ROOT, 'gpurun_out', 'gemm_tl_%d_%d%d.npy' % (mask, pro, epi)), d.numpy())
            print('%s mask %d: launch %.1f us' % (name, mask, us))
            analyse(d.numpy(), us, mask)


def analyse(d, us, mask):
    import numpy as np
    d = d[d[:, 0] > 0]
    t = d[:, :6].astype(np.float64)
    hw, xcc = d[:, 6], d[:, 7] & 0xf
    # the s_memtime counters of the XCDs are not synchronised: everything per XCD
    spans = [t[xcc == x, 5].max() - t[xcc == x, 0].min() for x in np.unique(xcc)]
    tk = np.mean(spans) / us
    print('   %d workgroups, %d XCDs, per-XCD stamp span %.0f..%.0f ticks -> %.1f ticks/us' % (
        len(d), len(spans), min(spans), max(spans), tk))
    names = ['prologue (first tile staged)', 'k loop', 'epilogue half 0', 'epilogue half 1',
             'stores acknowledged' if mask & 32 else '(end)']
    for i, nm in enumerate(names):
        dt = t[:, i + 1] - t[:, i]
        print('   %-30s mean %7.2f us   p10 %7.2f   p90 %7.2f' % (nm, dt.mean() / tk, np.quantile(dt, 0.1) / tk,
                                                                     np.quantile(dt, 0.9) / tk))
    print('   %-30s mean %7.2f us' % ('workgroup lifetime', (t[:, 5] - t[:, 0]).mean() / tk))
    cu_key = (xcc << 16) | (hw & 0xff00) | ((hw >> 13) & 7)
    gaps, per_cu = [], []
    for key in np.unique(cu_key):
        tt = t[cu_key == key]
        tt = tt[np.argsort(tt[:, 0])]
        per_cu.append(len(tt))
        if len(tt) > 1:
            gaps.append(tt[1:, 0] - tt[:-1, 5])
    g = np.concatenate(gaps)
    print('   distinct CU keys %d, workgroups per CU %.1f; gap end -> next start on one CU: mean %.2f us, p10 %.2f, '
          'p90 %.2f' % (len(per_cu), np.mean(per_cu), g.mean() / tk, np.quantile(g, 0.1) / tk, np.quantile(g, 0.9) / tk))
    fr = []
    for x in np.unique(xcc):
        tx = t[xcc == x]
        t0 = tx[:, 0].min()
        grid = np.linspace(0.1, 0.9, 200) * (tx[:, 5].max() - t0)
        inep = np.array([(((tx[:, 2] - t0) <= v) & ((tx[:, 5] - t0) > v)).sum() for v in grid])
        live = np.array([(((tx[:, 0] - t0) <= v) & ((tx[:, 5] - t0) > v)).sum() for v in grid])
        fr.append(inep / np.maximum(live, 1))
    fr = np.concatenate(fr)
    print('   share of an XCD\'s live workgroups that are in their epilogue: mean %.2f, p10 %.2f, p90 %.2f '
          '(lockstep -> bimodal)' % (fr.mean(), np.quantile(fr, 0.1), np.quantile(fr, 0.9)))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'build':
        build()
    else:
        run()
