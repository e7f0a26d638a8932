#!/usr/bin/env python
"""Developer aid: per-tile timeline of the weight-stationary tower GEMM (csrc/tower_gemm_bs.h) from in-kernel s_memtime stamps.
`build` (no GPU) compiles csrc/tower.hip with -DTFR_BS_STAMPS into tools/_ablate/libtower_bs_tl.so; `run` times plain / dgrad at
M = 512000 and prints where a tile's ticks go: waiting for a stage (+ barrier), multiplying, epilogue."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'tools', '_ablate')
LIB = os.path.join(OUT, 'libtower_bs_tl.so')


def build():
    os.makedirs(OUT, exist_ok=True)
    subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-shared', '-fPIC', '-I', os.path.join(ROOT, 'include'),
                    '-DTFR_BS_STAMPS', os.path.join(ROOT, 'ranking_amd', 'csrc', 'tower.hip'), '-o', LIB], check=True)


def run():
    import torch
    os.environ['TFR_GEMM_BS'] = '1'; os.environ['TFR_GEMM_BS_MIN_TILES'] = '1'
    dev = 'cuda'
    M, N, K = int(os.environ.get('M', 512000)), 512, 512
    A = torch.randn((M, K), device=dev).to(torch.bfloat16)
    W = (torch.randn((N, K), device=dev) * 0.05).to(torch.bfloat16)
    C = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    Zp = torch.randn((M, N), device=dev).to(torch.bfloat16)
    vec = lambda v: torch.full((max(N, K),), v, device=dev)
    sc, sh, mean, rstd = vec(1.0), vec(0.1), vec(0.0), vec(1.0)
    stats = torch.zeros(((M + 63) // 64, 2, N), device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    lib = ctypes.CDLL(LIB)
    f = lib.tfr_tower_gemm_bf16
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long,
                  ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                  ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long] + [ctypes.c_void_p] * 7
    buf = torch.zeros((256, 48, 10), dtype=torch.int64, device=dev)
    lib.tfr_prof_set_buffer_bs(ctypes.c_void_p(buf.data_ptr()))
    for name, epi in (('plain', 0), ('dgrad', 2)):
        call = lambda: f(p(A), K, p(W), K, p(C), N, M, N, K, 0, None, None, None, epi, p(stats), p(Zp), N, p(sc), p(sh), p(mean), p(rstd), None, None, st)
        for _ in range(3):
            assert call() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            call()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        buf.zero_(); call(); torch.cuda.synchronize()
        t = buf.cpu().double()[:, 4:44, :]                      # tiles 4 .. 43 of every workgroup (steady state)
        wait = sum((t[..., 2 * k + 1] - t[..., 2 * k]) for k in range(4))
        # multiply = from after a barrier to the next stage's wait (ks 0..2), and from the last barrier to the epilogue start
        mul = sum((t[..., 2 * k + 2] - t[..., 2 * k + 1]) for k in range(3)) + (t[..., 8] - t[..., 7])
        epi_t = t[..., 9] - t[..., 8]
        tile = t[:, 1:, 0] - t[:, :-1, 0]
        print('%s: %.1f us per launch (stamped build); per tile ticks: tile-to-tile %.0f | waits + barriers %.0f (per stage %s) | '
              'between barriers (fragment reads + 64 MFMAs per stage) %.0f | epilogue %.0f' % (
                  name, us, tile.mean(), wait.mean(), ' '.join('%.0f' % (t[..., 2 * k + 1] - t[..., 2 * k]).mean() for k in range(4)),
                  mul.mean(), epi_t.mean()), flush=True)
        lifetime = (buf.cpu().double()[:, :, 9].amax(dim=1) - buf.cpu().double()[:, 0, 0])
        print('   ticks per microsecond (workgroup lifetime / launch time, 62.5 tiles per workgroup, 48 stamped): tile-to-tile x 62.5 = %.0f ticks ~ %.1f us' % (
            tile.mean() * 62.5, us))


if __name__ == '__main__':
    {'build': build, 'run': run}[sys.argv[1]]()
