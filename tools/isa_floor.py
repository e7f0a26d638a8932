#!/usr/bin/env python
"""Instruction-mix issue floors of the O(L^2) pair sweeps, derived from the COMPILED kernels (VERDICT r2 "Next round" #5:
`valu_frac`'s floor was a constant measured once by hand and did not move with the loop).

    python tools/isa_floor.py            # recompiles the two sources to assembly, writes ranking_amd/csrc/isa_floor.json

For each hot kernel the gfx950 assembly (hipcc --cuda-device-only -S of the product source, product flags) is scanned for
its innermost loops (a backward branch to a label with no other loop inside).  A sweep loop is recognised by what it
evaluates per pair: a pair evaluation issues one v_rcp_f32 (the sigmoid) -- half of one in ApproxNDCG's forward sweep
since round 4, where two columns share a reciprocal (the loop with the v_pk_mul_f32 of their product) --, so
    cycles per 64 pair evaluations = sum of the issue costs of the VALU instructions of the loop body / #pairs in it
with the per-instruction issue costs measured on MI355X by tools/ubench.hip (cycles per wave-instruction and SIMD at the
2.4 GHz the roofline uses: plain VALU 2.38, v_pk_* 4.56, transcendental 8.5; profiles/r03_ubench.txt).  LDS / SALU /
waitcnt instructions issue from other ports and are not counted: this is a FLOOR of the sweeps' VALU issue time.
`trans_*` is the floor from the transcendental instructions alone.

bench.py reads the JSON (`roofline.valu_floor_ms`, `valu_frac`, `trans_floor_ms`, `trans_frac`); a CPU test checks that its
fingerprint matches the sources, so the figure cannot go stale silently.
"""
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'ranking_amd', 'csrc')
OUT = os.path.join(CSRC, 'isa_floor.json')
COST = {'plain': 2.38, 'pk': 4.56, 'trans': 8.5}        # tools/ubench.hip on MI355X (profiles/r03_ubench.txt)
TRANS = ('v_rcp_f32', 'v_log_f32', 'v_exp_f32', 'v_sqrt_f32', 'v_rsq_f32', 'v_sin_f32', 'v_cos_f32', 'v_rcp_iflag_f32')
KERNELS = {
    # name in the JSON: (source, mangled-name fragment of the instantiation bench.py runs, loop selectors)
    'approx_ndcg': ('approx_ndcg.hip', 'approx_ndcg_wave_kernelILi4E', 'two_largest_rcp'),
    'pairwise': ('pairwise.hip', 'lambdarank_group_kernelILi4ELb0ELb0E', 'hi_lo'),
}
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC']


def fingerprint():
    h = hashlib.sha256(json.dumps(COST, sort_keys=True).encode() + ' '.join(FLAGS).encode())
    for f in sorted(os.listdir(CSRC)):
        if f.endswith(('.hip', '.h')):
            with open(os.path.join(CSRC, f), 'rb') as fh:
                h.update(f.encode()); h.update(fh.read())
    with open(os.path.abspath(__file__), 'rb') as fh:
        h.update(fh.read())
    return h.hexdigest()


def assemble(src):
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, 'k.s')
        cmd = [hipcc] + FLAGS + ['--cuda-device-only', '-S', '-I', os.path.join(ROOT, 'include'), os.path.join(CSRC, src), '-o', out]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError('hipcc -S failed:\n' + res.stderr[-2000:])
        with open(out) as f:
            return f.read().splitlines()


def function_body(lines, fragment):
    start = None
    for i, ln in enumerate(lines):
        if start is None and re.match(r'^_Z\w*%s\w*:' % re.escape(fragment), ln):
            start = i
        elif start is not None and 's_endpgm' in ln:
            return lines[start:i + 1]
    raise RuntimeError('kernel %s not found in the assembly' % fragment)


def innermost_loops(body):
    labels = {}
    for i, ln in enumerate(body):
        m = re.match(r'^(\.LBB\d+_\d+):', ln)
        if m:
            labels[m.group(1)] = i
    loops = []
    for i, ln in enumerate(body):
        m = re.search(r'\bs_cbranch_\w+\s+(\.LBB\d+_\d+)', ln) or re.search(r'\bs_branch\s+(\.LBB\d+_\d+)', ln)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    inner = [lp for lp in loops if not any(o != lp and lp[0] <= o[0] and o[1] <= lp[1] for o in loops)]
    return inner


def classify(body, lo, hi):
    """Instruction classes of the loop body [lo, hi].  Blocks that a forward conditional branch inside the loop jumps
    over (the once-per-segment flush of the LambdaRank sweeps) are NOT counted: the floor is the straight-line trip."""
    c = {'plain': 0, 'pk': 0, 'trans': 0, 'lds': 0, 'rcp': 0, 'log': 0, 'exp': 0, 'pk_mul': 0}
    labels = {}
    for i in range(lo, hi + 1):
        m = re.match(r'^(\.LBB\d+_\d+):', body[i])
        if m:
            labels[m.group(1)] = i
    skipped = set()
    for i in range(lo, hi):
        m = re.search(r'\bs_cbranch_\w+\s+(\.LBB\d+_\d+)', body[i])
        if m and m.group(1) in labels and labels[m.group(1)] > i:
            skipped.update(range(i + 1, labels[m.group(1)]))
    for i in range(lo, hi + 1):
        if i in skipped:
            continue
        ln = body[i]
        t = ln.strip().split()
        if not t or t[0].startswith(('.', ';')) or t[0].endswith(':'):
            continue
        op = re.sub(r'_(e32|e64|dpp|sdwa)$', '', t[0])
        if op.startswith('ds_'):
            c['lds'] += 1
        elif op.startswith('v_'):
            if op in TRANS:
                c['trans'] += 1
                c['rcp'] += op.startswith('v_rcp')
                c['log'] += op == 'v_log_f32'
                c['exp'] += op == 'v_exp_f32'
            elif op.startswith('v_pk_'):
                c['pk'] += 1
                c['pk_mul'] += op == 'v_pk_mul_f32'
            else:
                c['plain'] += 1
    c['cycles'] = c['plain'] * COST['plain'] + c['pk'] * COST['pk'] + c['trans'] * COST['trans']
    c['trans_cycles'] = c['trans'] * COST['trans']
    return c


def per_pair(c, pairs_per_rcp=1):
    n = c['rcp'] * pairs_per_rcp
    return {'valu_cycles_per_64_pairs': c['cycles'] / n, 'trans_cycles_per_64_pairs': c['trans_cycles'] / n,
            'pairs_per_rcp': pairs_per_rcp,
            'loop_body': {k: c[k] for k in ('plain', 'pk', 'trans', 'lds', 'rcp', 'log', 'exp')}}


def analyse(name):
    src, frag, mode = KERNELS[name]
    body = function_body(assemble(src), frag)
    loops = [classify(body, lo, hi) for lo, hi in innermost_loops(body)]
    loops = [c for c in loops if c['rcp'] > 0]
    if mode == 'two_largest_rcp':
        # ApproxNDCG: the forward (ranks) and the backward sweep are the two loops with the most reciprocals (the x8
        # unrolled bodies); a pair is evaluated once in each
        # Round 4: the forward sweep of the default path shares ONE reciprocal between two columns (1/a = b rcp(a b)):
        # it is the loop with a v_pk_mul_f32 (the product a b), two pair evaluations per v_rcp_f32; the backward sweep
        # is the loop that reads two float4 (F and A) per four reciprocals.
        fwd2 = [c for c in loops if c['pk_mul'] > 0]
        rest = sorted((c for c in loops if c['pk_mul'] == 0), key=lambda c: -c['rcp'])
        bwd = next(c for c in rest if 2 * c['lds'] >= c['rcp'])
        if fwd2:
            parts = {'forward_sweep': per_pair(max(fwd2, key=lambda c: c['rcp']), 2), 'backward_sweep': per_pair(bwd)}
        else:
            fwd = next(c for c in rest if c is not bwd)
            parts = {'forward_sweep': per_pair(fwd), 'backward_sweep': per_pair(bwd)}
    else:
        # LambdaRank: per ACTIVE pair one "hi" evaluation (rcp + log) and one "lo" evaluation (rcp only)
        hi = max((c for c in loops if c['log'] > 0 and c['exp'] == 0), key=lambda c: c['rcp'])
        lo = max((c for c in loops if c['log'] == 0 and c['exp'] == 0), key=lambda c: c['rcp'])
        parts = {'hi_sweep': per_pair(hi), 'lo_sweep': per_pair(lo)}
    tot = sum(p['valu_cycles_per_64_pairs'] for p in parts.values())
    ttot = sum(p['trans_cycles_per_64_pairs'] for p in parts.values())
    return {'kernel': frag, 'source': src, 'parts': parts, 'valu_cycles_per_64_pairs': tot, 'trans_cycles_per_64_pairs': ttot}


def main():
    out = {'fingerprint': fingerprint(), 'issue_cost_cycles_at_2.4GHz': COST,
           'note': 'instruction-mix issue floor of the pair sweeps from the compiled gfx950 code (tools/isa_floor.py)'}
    for name in KERNELS:
        out[name] = analyse(name)
        print('%-12s %6.2f VALU cycles / 64 pairs (transcendental part %5.2f): %s' % (
            name, out[name]['valu_cycles_per_64_pairs'], out[name]['trans_cycles_per_64_pairs'],
            {k: round(v['valu_cycles_per_64_pairs'], 2) for k, v in out[name]['parts'].items()}))
    with open(OUT, 'w') as f:
        json.dump(out, f, indent=1)
    print('wrote', OUT)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'fingerprint':
        print(fingerprint())
    else:
        main()
