#!/usr/bin/env python
"""Assembles profiles/r03_*.{txt,json} from one consolidated GPU visit under gpurun_out/<tag>/ (tools/gpu_r03.sh):
bench.py lines, rocprofv3 --kernel-trace --stats tables, the separate --pmc passes and the traffic figures bench.py
reports as `roofline.traffic` (dominant kernel of every profiled workload, BASELINE configs 4 and 5 included)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKLOADS = ['pairwise_lambda', 'softmax', 'gumbel_approx_ndcg', 'ndcg_metric', 'approx_ndcg_l1000', 'e2e_softmax',
             'e2e_pairwise_lambda', 'e2e_approx_ndcg_l1000', 'e2e_groupwise_gumbel']
CLIP = 4000        # characters of a bench line kept in the text table (the full lines stay in gpurun_out/)


def last_line(path):
    if not os.path.exists(path):
        return '(missing: %s)' % path
    with open(path) as f:
        lines = [l for l in f.read().strip().splitlines() if l.startswith('{')]
    return lines[-1] if lines else '(no JSON line: see %s)' % path


def pmc_mean(path, kernel_sub, counter):
    if not os.path.exists(path):
        return None
    for line in open(path):
        if kernel_sub in line and counter in line:
            return float(line.split(counter)[1].split()[1])
    return None


def main(tag):
    R = os.path.join(ROOT, 'gpurun_out', tag)
    out = ['# Round 3, consolidated GPU visit %s (one MI355X, fresh box): bash tools/gpu_r03.sh %s bench all prof:... pmc:...\n'
           '# bench.py lines (graph replay; dominant-kernel time = HIP events around graph-replayed launches of that kernel),\n'
           '# rocprofv3 --kernel-trace --stats tables of the same commands.  e2e workloads run at the reference dropout 0.5;\n'
           '# `dropout_0` in their lines is the same step without Dropout.\n' % (tag, tag)]
    t = os.path.join(R, 'bench_default.time')
    wall = open(t).read().strip().replace('\n', '  ') if os.path.exists(t) else ''
    out.append('## python bench.py   (the driver\'s invocation: headline + `also` workloads, N = 1)   wall: %s\n%s\n'
               % (wall, last_line(os.path.join(R, 'bench_default.json'))))
    for w in WORKLOADS:
        p = os.path.join(R, 'bench_%s.json' % w)
        if os.path.exists(p):
            out.append('## python bench.py --workload %s --steps 50 --warmup 5 --no-cpu-baseline\n%s\n' % (w, last_line(p)[:CLIP]))
    for f in sorted(os.listdir(R)):
        m = re.match(r'stats_(.+)\.txt', f)
        if m:
            out.append('## rocprofv3 --kernel-trace --stats -- python bench.py --workload %s --steps 50 --warmup 5 '
                       '--no-cpu-baseline --also none\n%s\n' % (m.group(1), open(os.path.join(R, f)).read().rstrip()))
    open(os.path.join(ROOT, 'profiles', 'r03_all_workloads.txt'), 'w').write('\n'.join(out))

    pm = ['# Round 3 PMC passes (visit %s): separate rocprofv3 --pmc runs as MI355X_MICROARCH.md prescribes (FETCH_SIZE / WRITE_SIZE in\n'
          '# KiB per dispatch; FETCH_SIZE x 2 on gfx950 for wide coalesced reads).  Columns: mean counter value per dispatch, avg ns.\n' % tag]
    traffic = {}
    for w in ('approx_ndcg', 'pairwise_lambda', 'e2e_approx_ndcg_l1000', 'e2e_groupwise_gumbel', 'e2e_softmax'):
        for c in ('fetch', 'write', 'sq'):
            p = os.path.join(R, 'pmc_%s_%s.txt' % (c, w))
            if os.path.exists(p) and not open(p).read().startswith('Traceback'):
                body = [l for l in open(p).read().rstrip().splitlines()
                        if not l.startswith('void at::') and 'rocclr' not in l]          # (torch's own tiny kernels)
                pm.append('## rocprofv3 --pmc <%s counters> -- python bench.py --workload %s --steps 20 --warmup 2 '
                          '--no-cpu-baseline --also none\n%s\n' % (c, w, '\n'.join(body)))
    for w, sub, B, L in (('approx_ndcg', 'approx_ndcg_wave_kernel', 16384, 200),
                         ('pairwise_lambda', 'lambdarank_group_kernel', 4096, 200)):
        f = pmc_mean(os.path.join(R, 'pmc_fetch_%s.txt' % w), sub, 'FETCH_SIZE')
        wr = pmc_mean(os.path.join(R, 'pmc_write_%s.txt' % w), sub, 'WRITE_SIZE')
        if f and wr:
            traffic[w] = dict(B=B, L=L, kernel=sub, algorithmic_bytes=(12 * L + 12) * B, fetch_kib=f, write_kib=wr,
                              traffic_bytes=int(round((f * 2 + wr) * 1024)))
    # e2e: the dominant kernel = the hidden-layer forward GEMM (BN + ReLU + Dropout prologue, bias + statistics epilogue)
    for w, B, L in (('e2e_approx_ndcg_l1000', 512, 1000), ('e2e_groupwise_gumbel', 512, 50), ('e2e_softmax', 4096, 100)):
        pf, pw = os.path.join(R, 'pmc_fetch_%s.txt' % w), os.path.join(R, 'pmc_write_%s.txt' % w)
        M = B * L
        unit = M * 512 * 2                                       # one [M, 512] bf16 matrix
        names = {'tower_gemm256p_kernel<2, 1, true>': ('forward hidden layer (BN + ReLU + Dropout prologue): reads z, writes z', 2 * unit),
                 'tower_gemm256p_kernel<0, 2, true>': ('dgrad (ReLU / Dropout backward epilogue): reads dz and Zp, writes dy', 3 * unit),
                 'tower_wgrad256_kernel<2>': ('weight gradient: reads dz and z; writes fp32 split slabs of [512, 512]', 2 * unit),
                 'tower_bn_bwd_apply_kernel': ('dz = p dy + q z + r in place', 3 * unit)}
        ent = {}
        for k, (note, alg) in names.items():
            f, wr = pmc_mean(pf, k, 'FETCH_SIZE'), pmc_mean(pw, k, 'WRITE_SIZE')
            if f is not None and wr is not None:
                ent[k] = dict(note=note, fetch_kib=f, write_kib=wr, algorithmic_bytes=alg,
                              traffic_bytes=int(round((2 * f + wr) * 1024)))
        k0 = 'tower_gemm256p_kernel<2, 1, true>'
        if k0 in ent:
            traffic[w] = dict(B=B, L=L, kernel=k0 + ' (' + ent[k0]['note'] + ')', fetch_kib=ent[k0]['fetch_kib'],
                              write_kib=ent[k0]['write_kib'], traffic_bytes=ent[k0]['traffic_bytes'],
                              algorithmic_bytes=ent[k0]['algorithmic_bytes'], others={k: v for k, v in ent.items() if k != k0})
    open(os.path.join(ROOT, 'profiles', 'r03_pmc.txt'), 'w').write('\n'.join(pm))
    tp = os.path.join(ROOT, 'profiles', 'r03_traffic.json')
    old = {}
    old['_comment'] = ('HBM bytes per launch of the dominant kernel from separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB per '
                       'dispatch, mean over dispatches; visit %s, tables in profiles/r03_pmc.txt), corrected as MI355X_MICROARCH.md '
                       'prescribes for gfx950 (FETCH_SIZE x 2 for wide coalesced reads).  bench.py copies the entry that matches its '
                       'workload and batch into roofline.traffic and says so in roofline.traffic_source.  e2e entries: the '
                       'hidden-layer forward GEMM bench.py names as the dominant kernel, at the reference dropout 0.5.' % tag)
    old.update(traffic)
    json.dump(old, open(tp, 'w'), indent=1)
    for w, v in traffic.items():
        print(w, 'traffic / algorithmic = %.3f' % (v['traffic_bytes'] / v['algorithmic_bytes']))


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'r03z')
