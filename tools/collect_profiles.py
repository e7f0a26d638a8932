#!/usr/bin/env python
"""Assembles profiles/r02_*.{txt,json} from one consolidated GPU visit under gpurun_out/<tag>/ (tools/gpu_r02.sh):
bench.py lines, rocprofv3 --kernel-trace --stats tables, the separate --pmc passes and the traffic figures bench.py
reports as `roofline.traffic`."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKLOADS = ['pairwise_lambda', 'softmax', 'gumbel_approx_ndcg', 'ndcg_metric', 'approx_ndcg_l1000', 'e2e_softmax',
             'e2e_pairwise_lambda', 'e2e_approx_ndcg_l1000', 'e2e_groupwise_gumbel']


def last_line(path):
    with open(path) as f:
        lines = [l for l in f.read().strip().splitlines() if l.startswith('{')]
    return lines[-1] if lines else '(no JSON line: see %s)' % path


def pmc_mean(path, kernel_sub, counter):
    for line in open(path):
        if kernel_sub in line and counter in line:
            return float(line.split(counter)[1].split()[1])
    return None


def main(tag):
    R = os.path.join(ROOT, 'gpurun_out', tag)
    out = ['# Round 2, consolidated GPU visit %s (one MI355X, fresh box): bash tools/gpu_r02.sh %s tests all prof:... pmc:...\n'
           '# bench.py lines (graph replay; dominant-kernel time = HIP events around graph-replayed launches of that kernel),\n'
           '# rocprofv3 --kernel-trace --stats tables of the same commands.\n' % (tag, tag)]
    t = os.path.join(R, 'bench_default.time')
    wall = open(t).read().strip().replace('\n', '  ') if os.path.exists(t) else ''
    out.append('## python bench.py   (the driver\'s invocation: headline + `also` workloads, N = 1)   wall: %s\n%s\n'
               % (wall, last_line(os.path.join(R, 'bench_default.json'))))
    for w in WORKLOADS:
        p = os.path.join(R, 'bench_%s.json' % w)
        if os.path.exists(p):
            out.append('## python bench.py --workload %s --steps 50 --warmup 5\n%s\n' % (w, last_line(p)))
    for f in sorted(os.listdir(R)):
        m = re.match(r'stats_(.+)\.txt', f)
        if m:
            out.append('## rocprofv3 --kernel-trace --stats -- python bench.py --workload %s --steps 50 --warmup 5 '
                       '--no-cpu-baseline --also none\n%s\n' % (m.group(1), open(os.path.join(R, f)).read().rstrip()))
    open(os.path.join(ROOT, 'profiles', 'r02_all_workloads.txt'), 'w').write('\n'.join(out))
    pm = ['# Round 2 PMC passes (visit %s): separate rocprofv3 --pmc runs as MI355X_MICROARCH.md prescribes (FETCH_SIZE / WRITE_SIZE in\n'
          '# KiB per dispatch; FETCH_SIZE x 2 on gfx950 for wide coalesced reads).  Columns: mean counter value per dispatch, avg ns.\n' % tag]
    traffic = {}
    for w, sub, B, L in (('approx_ndcg', 'approx_ndcg_wave_kernel', 16384, 200), ('pairwise_lambda', 'pairwise_lean_kernel', 4096, 200)):
        vals = {}
        for c in ('fetch', 'write', 'sq'):
            p = os.path.join(R, 'pmc_%s_%s.txt' % (c, w))
            if not os.path.exists(p):
                continue
            pm.append('## rocprofv3 --pmc <%s counters> -- python bench.py --workload %s --steps 20 --warmup 2 --no-cpu-baseline '
                      '--also none\n%s\n' % (c, w, open(p).read().rstrip()))
            if c == 'fetch':
                vals['fetch_kib'] = pmc_mean(p, sub, 'FETCH_SIZE')
            if c == 'write':
                vals['write_kib'] = pmc_mean(p, sub, 'WRITE_SIZE')
        if vals.get('fetch_kib') and vals.get('write_kib'):
            traffic[w] = dict(B=B, L=L, kernel=sub, algorithmic_bytes=(12 * L + 12) * B,
                              traffic_bytes=int(round((vals['fetch_kib'] * 2 + vals['write_kib']) * 1024)), **vals)
    e2e = None
    pf, pw = os.path.join(R, 'pmc_fetch_e2e_softmax.txt'), os.path.join(R, 'pmc_write_e2e_softmax.txt')
    if os.path.exists(pf) and os.path.exists(pw):
        for c, p in (('fetch', pf), ('write', pw)):
            pm.append('## rocprofv3 --pmc <%s counters> -- python bench.py --workload e2e_softmax --steps 20 --warmup 2 '
                      '--no-cpu-baseline --also none\n%s\n' % (c, open(p).read().rstrip()))
        M, W = 4096 * 100, 512
        unit = M * W * 2                                         # one [M, 512] bf16 matrix
        names = {'tower_gemm256p_kernel<0, 2, false>': ('dgrad, the longest kernel of the step: reads dz and Zp, writes dy', 3 * unit),
                 'tower_gemm256p_kernel<2, 1, false>': ('forward hidden layer: reads z, writes z', 2 * unit),
                 'tower_wgrad256_kernel<2>': ('weight gradient: reads dz and z; writes 64 fp32 split slabs of [512, 512]', 2 * unit),
                 'tower_bn_bwd_apply_kernel': ('dz = p dy + q z + r in place', 3 * unit),
                 'tower_out_bwd_kernel<2, 1, 64, 1>': ('last layer backward, sums only', unit),
                 'tower_out_bwd_kernel<2, 1, 64, 2>': ('last layer backward, recompute dy, write dz', 2 * unit)}
        ent = {}
        for k, (note, alg) in names.items():
            f, w = pmc_mean(pf, k, 'FETCH_SIZE'), pmc_mean(pw, k, 'WRITE_SIZE')
            if f is not None and w is not None:
                ent[k] = dict(note=note, fetch_kib=f, write_kib=w, algorithmic_bytes=alg,
                              traffic_bytes=int(round((2 * f + w) * 1024)))
        k0 = 'tower_gemm256p_kernel<0, 2, false>'
        if k0 in ent:
            e2e = dict(B=4096, L=100, kernel=k0 + ' (' + ent[k0]['note'] + ')', fetch_kib=ent[k0]['fetch_kib'],
                       write_kib=ent[k0]['write_kib'], traffic_bytes=ent[k0]['traffic_bytes'],
                       algorithmic_bytes=ent[k0]['algorithmic_bytes'], others={k: v for k, v in ent.items() if k != k0})
    open(os.path.join(ROOT, 'profiles', 'r02_pmc.txt'), 'w').write('\n'.join(pm))
    tp = os.path.join(ROOT, 'profiles', 'r02_traffic.json')
    old = json.load(open(tp)) if os.path.exists(tp) else {}
    old['_comment'] = ('HBM bytes per launch of the dominant kernel from separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB per '
                       'dispatch, mean over dispatches; visit %s, tables in profiles/r02_pmc.txt), corrected as MI355X_MICROARCH.md '
                       'prescribes for gfx950 (FETCH_SIZE x 2 for wide coalesced reads).  bench.py copies the entry that matches its '
                       'workload and batch into roofline.traffic and says so in roofline.traffic_source.' % tag)
    for w, v in traffic.items():
        old[w] = v
    if e2e is not None:
        old['e2e_softmax'] = e2e
    json.dump(old, open(tp, 'w'), indent=1)
    for w, v in traffic.items():
        print(w, 'traffic / algorithmic = %.3f' % (v['traffic_bytes'] / v['algorithmic_bytes']))


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'r02z')
