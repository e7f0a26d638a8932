#!/usr/bin/env python
"""Assembles profiles/r06_*.{txt,json} from one consolidated GPU visit under gpurun_out/<tag>/ (bash tools/gpu_r06.sh
<tag> profiles): bench.py lines, rocprofv3 --kernel-trace --stats tables of the same commands, the separate --pmc
passes (FETCH_SIZE / WRITE_SIZE) and the traffic figures bench.py reports as `roofline.traffic`."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLIP = 4000


def last_line(path):
    if not os.path.exists(path):
        return '(missing: %s)' % path
    lines = [l for l in open(path).read().strip().splitlines() if l.startswith('{')]
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
    out = ['# Round 6, consolidated GPU visit %s (one MI355X, fresh box): bash tools/gpu_r06.sh %s profiles\n'
           '# bench.py lines (graph replay; dominant-kernel time = HIP events around graph-replayed launches of that kernel),\n'
           '# rocprofv3 --kernel-trace --stats tables of the same commands.  e2e workloads run at the reference dropout 0.5;\n'
           '# `dropout_0` in their lines is the same step without Dropout.  `*_hbm` = the same kernels on a cycled working\n'
           '# set of ~1.3 GB (beyond the 256 MB Infinity Cache).\n' % (tag, tag)]
    for f in sorted(os.listdir(R)):
        m = re.match(r'one_(.+)\.out', f)
        if m:
            out.append('## python bench.py --workload %s --also none --no-cpu-baseline --steps 50 --warmup 5\n%s\n'
                       % (m.group(1), last_line(os.path.join(R, f))[:CLIP]))
    for f in sorted(os.listdir(R)):
        m = re.match(r'stats_(.+)\.txt', f)
        if m:
            out.append('## rocprofv3 --kernel-trace --stats -- python bench.py --workload %s --steps 50 --warmup 5 '
                       '--no-cpu-baseline --also none\n%s\n' % (m.group(1), open(os.path.join(R, f)).read().rstrip()))
    open(os.path.join(ROOT, 'profiles', 'r06_all_workloads.txt'), 'w').write('\n'.join(out))

    pm = ['# Round 6 PMC passes (visit %s): separate rocprofv3 --pmc runs as MI355X_MICROARCH.md prescribes (FETCH_SIZE / WRITE_SIZE in\n'
          '# KiB per dispatch; FETCH_SIZE x 2 on gfx950 for wide coalesced reads).  Columns: mean counter value per dispatch, avg ns.\n' % tag]
    for f in sorted(os.listdir(R)):
        m = re.match(r'pmc_(fetch|write|sq)_(.+)\.txt', f)
        if m and not open(os.path.join(R, f)).read().startswith('Traceback'):
            body = [l for l in open(os.path.join(R, f)).read().rstrip().splitlines()
                    if not l.startswith('void at::') and 'rocclr' not in l]
            what = 'SQ_* (waves, wave / busy cycles in units of 4 cycles, instruction counts)' if m.group(1) == 'sq' else m.group(1).upper() + '_SIZE'
            how = ' --no-graph --kernel-timing none (eager launches: rocprofv3 --pmc died on the 128-launch graphs)' if m.group(2).endswith('_hbm') else ''
            pm.append('## rocprofv3 --pmc %s -- python bench.py --workload %s --no-cpu-baseline --also none%s\n%s\n'
                      % (what, m.group(2), how, '\n'.join(body)))
    open(os.path.join(ROOT, 'profiles', 'r06_pmc.txt'), 'w').write('\n'.join(pm))

    traffic = {}
    for w, sub, B, L, alg in (('approx_ndcg', 'approx_ndcg_wave_kernel', 16384, 200, (12 * 200 + 12) * 16384),
                              ('pairwise_lambda', 'lambdarank_group_kernel', 4096, 200, (12 * 200 + 12) * 4096),
                              ('softmax_hbm', 'softmax_pack_kernel', 65536, 100, (12 * 100 + 12) * 65536),
                              ('ndcg_metric_hbm', 'ndcg_lean_kernel', 16384, 200, (8 * 200 + 24) * 16384)):
        f = pmc_mean(os.path.join(R, 'pmc_fetch_%s.txt' % w), sub, 'FETCH_SIZE')
        wr = pmc_mean(os.path.join(R, 'pmc_write_%s.txt' % w), sub, 'WRITE_SIZE')
        if f is not None and wr is not None:
            traffic[w] = dict(B=B, L=L, kernel=sub, algorithmic_bytes=alg, fetch_kib=f, write_kib=wr,
                              traffic_bytes=int(round((f * 2 + wr) * 1024)))
            # VALU pipe utilisation from the SQ pass: a wave64 VALU instruction occupies its SIMD for 4 cycles
            # (SQ_ACTIVE_INST_VALU counts those quads), SQ_BUSY_CYCLES is summed over the 32 shader engines
            sq = os.path.join(R, 'pmc_sq_%s.txt' % w)
            act, busy = pmc_mean(sq, sub, 'SQ_ACTIVE_INST_VALU'), pmc_mean(sq, sub, 'SQ_BUSY_CYCLES')
            nv, ns = pmc_mean(sq, sub, 'SQ_INSTS_VALU'), pmc_mean(sq, sub, 'SQ_INSTS_SALU')
            if act and busy:
                traffic[w].update(valu_busy_frac=act * 4.0 / 1024.0 / (busy / 32.0), valu_insts_per_list=nv / B,
                                  salu_insts_per_list=(ns or 0.0) / B)
    for w, B, L in (('e2e_approx_ndcg_l1000', 512, 1000), ('e2e_groupwise_gumbel', 512, 50), ('e2e_softmax', 4096, 100)):
        pf, pw = os.path.join(R, 'pmc_fetch_%s.txt' % w), os.path.join(R, 'pmc_write_%s.txt' % w)
        unit = B * L * 512 * 2                                   # one [M, 512] bf16 matrix
        f = wr = k0 = None
        for k0 in ('tower_gemm256p_kernel<2, 1, 2, 0>', 'tower_gemm256p_kernel<2, 1, 2, 1>', 'tower_gemm256p_kernel<2, 1, 2, 2>', 'tower_gemm256p_kernel<2, 1, 2>'):
            f, wr = pmc_mean(pf, k0, 'FETCH_SIZE'), pmc_mean(pw, k0, 'WRITE_SIZE')
            if f is not None:
                break
        if f is not None and wr is not None:
            traffic[w] = dict(B=B, L=L, kernel=k0 + ' (forward hidden layer, BN + ReLU + Dropout prologue: reads z, writes z; in the training step also the transformed operand for the weight gradient)',
                              fetch_kib=f, write_kib=wr, traffic_bytes=int(round((2 * f + wr) * 1024)), algorithmic_bytes=2 * unit)
    doc = {'_comment': ('HBM bytes per launch of the dominant kernel from separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB per '
                        'dispatch, mean over dispatches; visit %s, tables in profiles/r06_pmc.txt), corrected as MI355X_MICROARCH.md '
                        'prescribes for gfx950 (FETCH_SIZE x 2 for wide coalesced reads).  bench.py copies the entry that matches its '
                        'workload and batch into roofline.traffic and says so in roofline.traffic_source.' % tag)}
    doc.update(traffic)
    json.dump(doc, open(os.path.join(ROOT, 'profiles', 'r06_traffic.json'), 'w'), indent=1)
    for w, v in traffic.items():
        print(w, 'traffic / algorithmic = %.3f' % (v['traffic_bytes'] / v['algorithmic_bytes']))


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'r06p')
