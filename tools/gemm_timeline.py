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
    """Per-CU timeline (the s_memtime counters are not synchronised across XCDs / shader engines, so everything is
    computed per CU: HW_ID + XCC_ID identify it): ticks per microsecond from the CU's busy span against the launch
    duration, phase means, idle gap between consecutive workgroups (round-1 kernel) or tiles (persistent kernel)."""
    import numpy as np
    d = d[d[:, 0] > 0]
    t = d[:, :6].astype(np.float64)
    hw, xcc = d[:, 6], d[:, 7] & 0xf
    key = (xcc << 16) | (hw & 0xff00) | ((hw >> 13) & 7)
    spans, gaps = [], []
    for k in np.unique(key):
        tt = t[key == k]
        tt = tt[np.argsort(tt[:, 0])]
        spans.append(tt[-1, 5] - tt[0, 0])
        if len(tt) > 1:
            gaps.append(tt[1:, 0] - tt[:-1, 5])
    tk = np.median(spans) / us
    g = np.concatenate(gaps) if gaps else np.zeros(1)
    print('   %d tiles on %d CUs, %.0f ticks/us' % (len(d), len(spans), tk))
    persistent = bool((t[:, 1] == t[:, 0]).all())           # the persistent kernel stamps 0, 2, 3, 4, 5 only
    phases = ([(0, 2, 'k loop'), (2, 3, 'epilogue half 0'), (3, 4, 'epilogue half 1'), (4, 5, 'stats / end')] if persistent else
              [(0, 1, 'first tile staged'), (1, 2, 'k loop'), (2, 3, 'epilogue half 0'), (3, 4, 'epilogue half 1'),
               (4, 5, 'stores acknowledged' if mask & 32 else 'end')])
    for i0, i1, nm in phases:
        dt = t[:, i1] - t[:, i0]
        print('   %-22s mean %6.2f us   p10 %6.2f   p90 %6.2f' % (nm, dt.mean() / tk, np.quantile(dt, .1) / tk,
                                                                  np.quantile(dt, .9) / tk))
    print('   tile %.2f us, gap to the next one on the same CU %.2f us' % ((t[:, 5] - t[:, 0]).mean() / tk, g.mean() / tk))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'build':
        build()
    elif len(sys.argv) > 4 and sys.argv[1] == 'analyse':      # analyse <file.npy> <launch us> <mask>: offline
        import numpy as np
        analyse(np.load(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4]))
    else:
        run()
