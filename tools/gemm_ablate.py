#!/usr/bin/env python
"""Developer aid: where does a 256 x 256 tower GEMM launch spend its time?

`build` (no GPU needed) compiles ranking_amd/csrc/tower.hip into tools/_ablate/libtower_ab<mask>.so once per ablation
mask (-DTFR_GEMM_ABLATE=mask: 1 no MFMA block, 2 no global->LDS staging in the k loop, 4 no epilogue, 8 MFMAs on
registers without ds_read); `run` times every variant on the config-2 hidden-layer shape (M = 409600, N = K = 512) for
the forward (prologue 2, epilogue 1), plain (0, 0) and dgrad (0, 2) forms.  Ablated variants compute garbage: timing only.
"""
import ctypes
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'tools', '_ablate')
SRC = os.path.join(ROOT, 'ranking_amd', 'csrc', 'tower.hip')
MASKS = [int(m) for m in os.environ.get('MASKS', '0,1,2,4,6,14,5,3').split(',')]
NAMES = {0: 'full', 1: 'no MFMA', 2: 'no staging', 4: 'no epilogue', 6: 'LDS reads + MFMA only', 14: 'MFMA only',
         5: 'staging + barriers only', 3: 'epilogue only'}


def lib_path(mask, tag=''):
    return os.path.join(OUT, 'libtower_ab%d%s.so' % (mask, tag))


def build(extra=(), tag=''):
    os.makedirs(OUT, exist_ok=True)

    def one(mask):
        cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-DTFR_GEMM_ABLATE=%d' % mask,
               *extra, SRC, '-o', lib_path(mask, tag)]
        subprocess.run(cmd, check=True)
        return mask
    with ThreadPoolExecutor(8) as ex:
        for m in ex.map(one, MASKS):
            print('built', lib_path(m, tag))


def run(tag=''):
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
    forms = [('fwd hidden (pro 2, epi 1)', 2, 1), ('plain (0, 0)', 0, 0), ('dgrad (0, 2)', 0, 2)]
    print('%-28s' % 'variant' + ''.join('%28s' % f[0] for f in forms))
    for mask in MASKS:
        if not os.path.exists(lib_path(mask, tag)):
            continue
        lib = ctypes.CDLL(lib_path(mask, tag))
        f = lib.tfr_tower_gemm_bf16
        f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long,
                      ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                      ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long] + \
            [ctypes.c_void_p] * 7
        row = '%-28s' % ('%2d %s' % (mask, NAMES[mask]))
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
        print(row, flush=True)


if __name__ == '__main__':
    mode = sys.argv[1] if len(sys.argv) > 1 else 'run'
    tag = sys.argv[2] if len(sys.argv) > 2 else ''
    if mode == 'build':
        build(tuple(sys.argv[3:]), tag)
    else:
        run(tag)
