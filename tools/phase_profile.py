#!/usr/bin/env python
"""Developer aid: per-phase cycle breakdown of the ApproxNDCG wave kernel from in-kernel
s_memtime stamps (libtfr_hip_prof.so = the product sources + -DTFR_PROFILE_STAMPS)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ranking_amd import _lib  # noqa: E402
from ranking_amd.synthetic import make_batch  # noqa: E402


def main():
    if not os.path.exists(_lib.PROF_LIB_PATH):
        _lib.build_profiling()
    lib = ctypes.CDLL(_lib.PROF_LIB_PATH)
    B, L = 16384, 200
    labels, logits = make_batch(B, L, seed=4)
    order = os.environ.get('ORDER', '')
    if order:
        nv = (labels >= 0).sum(1)
        perm = torch.argsort(nv, descending=(order == 'desc'))
        labels, logits = labels[perm].contiguous(), logits[perm].contiguous()
    dev = 'cuda'
    labels, logits = labels.to(dev), logits.to(dev)
    r = torch.arange(1, L + 1, dtype=torch.float32)
    inv = (1.0 / torch.log1p(r)).to(dev)
    loss = torch.empty(B, device=dev); wout = torch.empty(B, device=dev); dl = torch.empty((B, L), device=dev)
    buf = torch.zeros((B, 8), dtype=torch.int64, device=dev)
    lib.tfr_prof_set_buffer(ctypes.c_void_p(buf.data_ptr()))
    f = lib.tfr_approx_ndcg_f32
    f.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 2 + [ctypes.c_float] + [ctypes.c_int] + [ctypes.c_void_p] * 5
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        rc = f(logits.data_ptr(), labels.data_ptr(), None, inv.data_ptr(), None, B, L, 0.1, 0, loss.data_ptr(),
               wout.data_ptr(), dl.data_ptr(), None, st)
    torch.cuda.synchronize()
    assert rc == 0
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f(logits.data_ptr(), labels.data_ptr(), None, inv.data_ptr(), None, B, L, 0.1, 0, loss.data_ptr(),
          wout.data_ptr(), dl.data_ptr(), None, st)
    e1.record(); torch.cuda.synchronize()
    print('ORDER=%r kernel %.4f ms' % (order, e0.elapsed_time(e1) / 20))
    d = buf.cpu()
    t = d[:, :6].double()
    names = ['load+stats', 'gains+idealDCG', 'compact+exp', 'fwd sweep', 'bwd sweep']
    tot = (t[:, 5] - t[:, 0]).mean().item()
    print('mean ticks per list-wave: total %.0f' % tot)
    for i, nme in enumerate(names):
        dt = (t[:, i + 1] - t[:, i]).mean().item()
        print('  %-16s %8.0f  %5.1f %%' % (nme, dt, 100 * dt / tot))
    import numpy as np
    st_, en_ = t[:, 0].numpy(), t[:, 5].numpy()
    lo = st_.min()
    span_ = en_.max() - lo
    grid = np.linspace(0, span_, 21)
    act = [(int(((st_ - lo) <= g_) .sum() - ((en_ - lo) <= g_).sum())) for g_ in grid]
    print('active waves at 0,5,..,100 %% of the span (%.0f cycles): %s' % (span_, act))
    span = (t[:, 5].max() - t[:, 0].min()).item()
    print('kernel span %.0f ticks; sum of wave lifetimes / span = %.1f concurrent waves (chip)' % (
        span, (t[:, 5] - t[:, 0]).sum().item() / span))
    n = d[:, 7].double()
    print('mean n_valid %.1f, mean n^2 %.0f' % (n.mean().item(), (n * n).mean().item()))
    # concurrency per SIMD: decode HW_ID (wave_id[3:0], simd_id[5:4], cu_id[11:8], sh_id[12], se_id[15:13]), xcc in upper bits
    hw = d[:, 6]
    key = (hw >> 4) & 0xfffffff
    print('distinct simd keys:', len(set(key.tolist())))


def main_pairwise():
    if not os.path.exists(_lib.PROF_LIB_PATH):
        _lib.build_profiling()
    lib = ctypes.CDLL(_lib.PROF_LIB_PATH)
    B, L = int(os.environ.get('B', '4096')), 200
    labels, logits = make_batch(B, L, seed=4)
    dev = 'cuda'
    labels, logits = labels.to(dev), logits.to(dev)
    r = torch.arange(1, L + 2, dtype=torch.float32)
    import math
    disc = (math.log(2.) / torch.log1p(r)).to(dev)
    row_loss = torch.empty((B, L), device=dev); dl = torch.empty((B, L), device=dev)
    buf = torch.zeros((B, 8), dtype=torch.int64, device=dev)
    lib.tfr_prof_set_buffer_pw(ctypes.c_void_p(buf.data_ptr()))
    f = lib.tfr_pairwise_logistic_f32
    f.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 2 + [ctypes.c_float] + [ctypes.c_int] * 2 + \
        [ctypes.c_void_p] * 2 + [ctypes.c_int] * 2 + [ctypes.c_float] + [ctypes.c_void_p] * 5
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    call = lambda: f(logits.data_ptr(), labels.data_ptr(), None, None, None, 2, 0, 0.0, 1, 1, None, disc.data_ptr(),
                     B, L, 1.0, row_loss.data_ptr(), None, None, dl.data_ptr(), st)
    for _ in range(3):
        rc = call()
    torch.cuda.synchronize()
    assert rc == 0, rc
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        call()
    e1.record(); torch.cuda.synchronize()
    print('pairwise (NDCG lambda) kernel %.4f ms' % (e0.elapsed_time(e1) / 20))
    t = buf.cpu()[:, :5].double()
    names = (['load+compact', 'rank count', 'grade order + ideal DCG + records', 'pair sweeps'] if os.environ.get('TFR_PAIRWISE_LEAN', '1') != '0'
             else ['load+compact', 'ideal DCG', 'rank count + re-home', 'pair sweep'])
    tot = (t[:, 4] - t[:, 0]).mean().item()
    print('mean ticks per list-wave: total %.0f' % tot)
    for i, nme in enumerate(names):
        dt = (t[:, i + 1] - t[:, i]).mean().item()
        print('  %-36s %8.0f  %5.1f %%' % (nme, dt, 100 * dt / tot))
    n = buf.cpu()[:, 7].double()
    lab = labels.cpu()
    cnt = torch.stack([(lab == g).sum(dim=1).double() for g in range(5)], dim=1)
    higher = torch.flip(torch.cumsum(torch.flip(cnt, [1]), 1), [1]) - cnt
    print('mean n_valid %.1f, mean n^2 %.0f, mean active ordered pairs %.0f' % (
        n.mean().item(), (n * n).mean().item(), (cnt * higher).sum(dim=1).mean().item()))


def main_group():
    """LambdaRank group kernel (lambdarank_group.h): per-wave stamps [table, load+compact, rank count, grade order +
    ideal DCG, records, (barrier wait), sweeps]."""
    if not os.path.exists(_lib.PROF_LIB_PATH):
        _lib.build_profiling()
    lib = ctypes.CDLL(_lib.PROF_LIB_PATH)
    B, L = int(os.environ.get('B', '4096')), 200
    W = int(os.environ.get('TFR_LAMBDARANK_WAVES', '8'))
    labels, logits = make_batch(B, L, seed=4)
    dev = 'cuda'
    labels, logits = labels.to(dev), logits.to(dev)
    import math
    r = torch.arange(1, L + 2, dtype=torch.float32)
    disc = (math.log(2.) / torch.log1p(r)).to(dev)
    dl = torch.empty((B, L), device=dev); lst = torch.empty((B,), device=dev)
    lw = torch.full((B,), 1.0 / (B * L), device=dev)
    Wt_ = W + (max(0, int(os.environ['TFR_LAMBDARANK_HELPERS'])) if 'TFR_LAMBDARANK_HELPERS' in os.environ else W // 2)   # = grp_geometry()
    Wt_ = min(Wt_, 16)
    nwaves = ((B + W - 1) // W) * Wt_
    buf = torch.zeros((nwaves, 12), dtype=torch.int64, device=dev)
    lib.tfr_prof_set_buffer_pw(ctypes.c_void_p(buf.data_ptr()))
    from ranking_amd import _ops
    order = _ops.list_order(labels) if os.environ.get('ORDER', '1') != '0' else None
    f = lib.tfr_pairwise_loss_f32
    f.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 5 + [ctypes.c_int] * 2 + [ctypes.c_float] + [ctypes.c_int] * 2 + \
        [ctypes.c_void_p] * 2 + [ctypes.c_int] * 2 + [ctypes.c_float] + [ctypes.c_void_p] * 6 + [ctypes.c_uint32, ctypes.c_void_p]
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    call = lambda: f(0, logits.data_ptr(), labels.data_ptr(), None, None, lw.data_ptr(), 2, 0, 0.0, 1, 1, None,
                     disc.data_ptr(), B, L, 1.0, None, None, None, dl.data_ptr(),
                     None if order is None else order.data_ptr(), lst.data_ptr(), 0, st)
    stop = int(os.environ.get('STOP', '0'))
    if stop:
        # counter runs (under rocprofv3 --pmc): the kernel truncated after build phase `stop` (1 .. 5) or with one of the
        # two sweeps only (6 = hi, 7 = lo); differences of SQ_INSTS_* between successive runs = that phase's instructions
        lib.tfr_prof_set_stop_pw.argtypes = [ctypes.c_int]
        assert lib.tfr_prof_set_stop_pw(stop) == 0
        for _ in range(6):
            rc = call()
        torch.cuda.synchronize()
        assert rc == 0, rc
        print('STOP=%d: 6 launches' % stop)
        return
    for _ in range(3):
        rc = call()
    torch.cuda.synchronize()
    assert rc == 0, rc
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        call()
    e1.record(); torch.cuda.synchronize()
    print('lambdarank group kernel (stamped build) B=%d, %d lists + %d sweep-only waves per workgroup: %.4f ms' % (B, W, Wt_ - W, e0.elapsed_time(e1) / 20))
    d = buf.cpu()
    t = d[:, :8].double()
    build = t[:, 2] > 0                                        # builder waves (the sweep-only waves skip phase 1)
    names = ['loads + table + barrier', 'gains + compaction', 'rank count', 'grade order + ideal DCG', 'records + publish']
    tot = (t[:, 7] - t[:, 0]).mean().item()
    print('mean ticks per wave: total %.0f   (%d waves, %d builders)' % (tot, t.shape[0], int(build.sum())))
    tb = t[build]
    for i, nme in enumerate(names):
        dt = (tb[:, i + 1] - tb[:, i])
        print('  %-28s mean %8.0f  %5.1f %%   max %8.0f' % (nme, dt.mean().item(), 100 * dt.mean().item() / tot, dt.max().item()))
    sw = (t[:, 7] - torch.where(build, t[:, 5], t[:, 1]))
    print('  %-28s mean %8.0f  %5.1f %%   max %8.0f' % ('sweeps (+ waiting for lists)', sw.mean().item(), 100 * sw.mean().item() / tot, sw.max().item()))
    print('mean n_valid %.1f, passes per wave mean %.2f max %d, sleeps per wave mean %.1f max %d' % (
        d[build, 8].double().mean().item(), d[:, 9].double().mean().item(), int(d[:, 9].max()),
        d[:, 10].double().mean().item(), int(d[:, 10].max())))
    print('sweep ticks per pass: %.0f' % (sw.sum().item() / max(1, d[:, 9].sum().item())))
    # per-workgroup lifetime (waves of a workgroup share the XCD clock)
    Wt = Wt_
    tw = t.reshape(-1, Wt, 8)
    life = (tw[:, :, 7].max(dim=1).values - tw[:, :, 0].min(dim=1).values)
    print('workgroup lifetime: mean %.0f  max %.0f  min %.0f ticks' % (life.mean().item(), life.max().item(), life.min().item()))
    # placement: HW_ID (cu 11:8, sh 12, se 15:13) and XCC_ID of wave 0 of every workgroup
    hw = d[:, 11].reshape(-1, Wt)[:, 0]
    xcc = (hw >> 32) & 0xf
    cu = ((hw & 0xffffffff) >> 8) & 0xff                      # cu | sh | se bits
    key = (xcc * 256 + cu).tolist()
    from collections import defaultdict
    per_cu = defaultdict(list)
    for i, k in enumerate(key):
        per_cu[k].append(i)
    cnts = [len(v) for v in per_cu.values()]
    print('distinct (XCC, SE/SH/CU) keys %d; workgroups per key: min %d max %d; histogram %s' % (
        len(per_cu), min(cnts), max(cnts), {c: cnts.count(c) for c in sorted(set(cnts))}))
    st0 = tw[:, :, 0].min(dim=1).values
    en0 = tw[:, :, 7].max(dim=1).values
    for x in range(8):
        sel = (xcc == x)
        if sel.any():
            print('  XCC %d: %3d workgroups, lifetime mean %.0f max %.0f, start spread %.0f, last end - first start %.0f' % (
                x, int(sel.sum()), life[sel].mean().item(), life[sel].max().item(),
                (st0[sel].max() - st0[sel].min()).item(), (en0[sel].max() - st0[sel].min()).item()))
    # lifetime by how many workgroups shared the CU key
    for c in sorted(set(cnts)):
        idx = [i for v in per_cu.values() if len(v) == c for i in v]
        print('  CUs hosting %d workgroups: mean lifetime %.0f (n = %d)' % (c, life[idx].mean().item(), len(idx)))

    # what the lifetime of a workgroup follows: its own work (passes swept by its waves = passes of its lists), the
    # builds (the longest build of the workgroup), or the partner workgroup on the same CU
    import numpy as np
    passes_wg = d[:, 9].reshape(-1, Wt).double().sum(dim=1)
    bt = (t[:, 5] - t[:, 0]).reshape(-1, Wt)[:, :W]
    build_max = bt.max(dim=1).values
    lf = life.numpy()
    print('per workgroup: passes mean %.1f min %d max %d; corr(lifetime, passes) %.3f; corr(lifetime, longest build) %.3f' % (
        passes_wg.mean().item(), int(passes_wg.min()), int(passes_wg.max()),
        np.corrcoef(lf, passes_wg.numpy())[0, 1], np.corrcoef(lf, build_max.numpy())[0, 1]))
    print('longest build of a workgroup: mean %.0f min %.0f max %.0f; first list published (min build): mean %.0f' % (
        build_max.mean().item(), build_max.min().item(), build_max.max().item(), bt.min(dim=1).values.mean().item()))
    sweep_span = en0 - (tw[:, :W, 5].min(dim=1).values)
    print('first publish -> workgroup end: mean %.0f min %.0f max %.0f ticks; per pass of the workgroup: %.0f' % (
        sweep_span.mean().item(), sweep_span.min().item(), sweep_span.max().item(), (sweep_span / passes_wg).mean().item()))
    pairs = [v for v in per_cu.values() if len(v) == 2]
    if pairs:
        a = np.array([lf[v[0]] for v in pairs]); b2 = np.array([lf[v[1]] for v in pairs])
        dst = np.array([abs(st0[v[0]].item() - st0[v[1]].item()) for v in pairs])
        cu_end = np.array([max(en0[v[0]].item(), en0[v[1]].item()) - min(st0[v[0]].item(), st0[v[1]].item()) for v in pairs])
        pw = np.array([passes_wg[v[0]].item() + passes_wg[v[1]].item() for v in pairs])
        print('CU pairs: corr(lifetime A, lifetime B) %.3f; start offset between the two: mean %.0f max %.0f; CU busy span mean %.0f min %.0f max %.0f; corr(span, passes of both) %.3f'
              % (np.corrcoef(a, b2)[0, 1], dst.mean(), dst.max(), cu_end.mean(), cu_end.min(), cu_end.max(), np.corrcoef(cu_end, pw)[0, 1]))
        q = np.percentile(cu_end, [5, 25, 50, 75, 95])
        print('CU busy span percentiles 5/25/50/75/95: %s' % ' '.join('%.0f' % x for x in q))
    dec = np.percentile(lf, [5, 25, 50, 75, 95])
    print('workgroup lifetime percentiles 5/25/50/75/95: %s' % ' '.join('%.0f' % x for x in dec))
    for x in range(8):
        sel = (xcc == x)
        if sel.any():
            print('  XCC %d: lifetime p5 %.0f p50 %.0f p95 %.0f' % ((x,) + tuple(np.percentile(lf[sel.numpy()], [5, 50, 95]))))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'group':
        main_group()
    elif len(sys.argv) > 1 and sys.argv[1] == 'pairwise':
        main_pairwise()
    else:
        main()
