#!/usr/bin/env python
"""Developer aid: where does a resident-panel tower GEMM launch (csrc/tower_gemm_rp.h) spend its time?
`build` (no GPU) compiles csrc/tower.hip into tools/_ablate/libtower_rp<mask>.so per ablation mask (-DTFR_RP_ABLATE: 1 no
activation loads in the k loop, 2 no epilogue, 4 no MFMAs, 8 no panel fragment reads); `run` times plain / forward / dgrad at
M = 512000, N = K = 512.  Ablated variants compute garbage: timing only."""
import ctypes
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'tools', '_ablate')
SRC = os.path.join(ROOT, 'ranking_amd', 'csrc', 'tower.hip')
MASKS = [int(m) for m in os.environ.get('MASKS', '0,1,2,3,8,11,6,16').split(',')]
NAMES = {16: 'full + stamps', 0: 'full', 1: 'no A loads', 2: 'no epilogue', 3: 'no A loads, no epilogue', 8: 'no panel reads', 11: 'MFMA only',
         4: 'no MFMA', 6: 'loads + panel reads only', 9: 'MFMA + epilogue', 10: 'loads + MFMA'}


TAG = os.environ.get('TAG', '')
EXTRA = os.environ.get('EXTRA', '').split()


def lib_path(mask):
    return os.path.join(OUT, 'libtower_rp%d%s.so' % (mask, TAG))


def build():
    os.makedirs(OUT, exist_ok=True)

    def one(mask):
        cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-shared', '-fPIC', '-I', os.path.join(ROOT, 'include'),
               '-DTFR_RP_ABLATE=%d' % mask, *EXTRA, SRC, '-o', lib_path(mask)]
        subprocess.run(cmd, check=True)
        return mask
    with ThreadPoolExecutor(int(os.environ.get('JOBS', '4'))) as ex:
        for m in ex.map(one, MASKS):
            print('built', lib_path(m), flush=True)


def run():
    import torch
    dev = 'cuda'
    M, N, K = int(os.environ.get('M', 512000)), 512, 512
    A = torch.randn((M, K), device=dev).to(torch.bfloat16)
    W = (torch.randn((N, K), device=dev) * 0.05).to(torch.bfloat16)
    C = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    Zp = torch.randn((M, N), device=dev).to(torch.bfloat16)
    vec = lambda v: torch.full((max(N, K),), v, device=dev)
    sc, sh, mean, rstd, bias = vec(1.0), vec(0.1), vec(0.0), vec(1.0), vec(0.01)
    stats = torch.zeros(((M + 63) // 64, 2, N), device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    forms = [('plain (0, 0)', 0, 0), ('fwd hidden (pro 2, epi 1)', 2, 1), ('dgrad (0, 2)', 0, 2)]
    print('%-28s' % 'variant' + ''.join('%28s' % f[0] for f in forms))
    for mask in MASKS:
        if not os.path.exists(lib_path(mask)):
            continue
        lib = ctypes.CDLL(lib_path(mask))
        f = lib.tfr_tower_gemm_bf16
        f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long,
                      ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                      ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long] + \
            [ctypes.c_void_p] * 7
        row = '%-28s' % ('%2d %s' % (mask, NAMES.get(mask, '')))
        buf = None
        if mask & 16:
            buf = torch.zeros((256, 8, 64, 4), dtype=torch.int64, device=dev)
            lib.tfr_prof_set_buffer_rp(ctypes.c_void_p(buf.data_ptr()))
        for _, pro, epi in forms:
            call = lambda: f(p(A), K, p(W), K, p(C), N, M, N, K, pro, p(sc), p(sh), (None if epi == 2 else p(bias)), epi, p(stats), p(Zp), N,
                             p(sc), p(sh), p(mean), p(rstd), None, None, st)
            for _ in range(3):
                rc = call()
            assert rc == 0, rc
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                call()
            e1.record()
            torch.cuda.synchronize()
            row += '%25.1f us' % (e0.elapsed_time(e1) * 100)
            if buf is not None:
                buf.zero_(); call(); torch.cuda.synchronize()
                t = buf.cpu().double()                       # [wg, wave, pass, 4]
                ok = t[..., 2] > 0
                t0 = t[..., 0][ok].min()
                kl = (t[..., 1] - t[..., 0])[ok]; ep = (t[..., 2] - t[..., 1])[ok]
                npass = ok.sum(dim=2).double()
                end = torch.where(ok, t[..., 2], torch.zeros_like(t[..., 2])).amax(dim=2)
                act = end > 0
                print('   [%s] stamps (s_memtime ticks): passes per wave %.1f (min %d max %d); k loop per pass mean %.0f p10 %.0f p90 %.0f; epilogue mean %.0f '
                      'p10 %.0f p90 %.0f; wave end - first start: min %.0f mean %.0f max %.0f; first pass start spread %.0f' % (
                          _, npass[act].mean(), npass[act].min(), npass[act].max(), kl.mean(), kl.quantile(0.1), kl.quantile(0.9), ep.mean(),
                          ep.quantile(0.1), ep.quantile(0.9), (end[act] - t0).min(), (end[act] - t0).mean(), (end[act] - t0).max(),
                          (t[..., 0, 0][act].max() - t0)), flush=True)
                # the k loop of a pass by its position in the launch (does it slow down under load / over time?)
                for lo, hi in ((0, 4), (4, 12), (12, 24), (24, 40)):
                    sel = ok[:, :, lo:hi]
                    if sel.any():
                        print('      passes %2d-%2d: k loop %.0f, epilogue %.0f' % (lo, hi - 1, (t[..., 1] - t[..., 0])[:, :, lo:hi][sel].mean(),
                                                                               (t[..., 2] - t[..., 1])[:, :, lo:hi][sel].mean()), flush=True)
        print(row, flush=True)


if __name__ == '__main__':
    {'build': build, 'run': run}[sys.argv[1]]()
