"""Prints the lines of a bench.py output file in brief: one row per workload (main + also)."""
import json
import sys


def row(name, d):
    if not isinstance(d, dict) or 'error' in d:
        print('  %-26s ERROR %s' % (name, (d or {}).get('error')))
        return
    if 'ms' in d and 'ms_per_step' not in d:               # an entry of the compact digest line (bench.digest_entry)
        print('  %-26s %10.4g lists/s  %8.4f ms/step  kernel_ms %s  frac %s  valu_frac %s  cpu %s  [digest]'
              % (name, d['value'], d['ms'], d.get('kernel_ms'), d.get('frac'), d.get('valu_frac'), d.get('cpu')))
        return
    r = d.get('roofline') or {}
    cb = d.get('cpu_baseline') or {}
    print('  %-26s %10.4g lists/s  %8.4f ms/step  kernel_ms %s  frac %s  valu_frac %s  cpu %s (%s cores)  x%s'
          % (name, d['value'], d['ms_per_step'], ('%.4f' % r['kernel_ms']) if r.get('kernel_ms') else None,
             ('%.4f' % r['frac']) if r.get('frac') is not None else None,
             ('%.3f' % r['valu_frac']) if r.get('valu_frac') else None,
             ('%.4g' % cb['value']) if cb.get('value') else None, cb.get('cores'),
             ('%.1f' % d['gpu_over_cpu']) if d.get('gpu_over_cpu') else None))


lines = [l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{')]
print('%d JSON line(s)' % len(lines))
for i, l in enumerate(lines):
    d = json.loads(l)
    print('line %d: n_gpus %s steps %s also=%s' % (i, d.get('n_gpus'), d.get('steps'), sorted(d.get('also', {}))))
    row(d['config']['workload'][:26], d)
    for k, v in d.get('also', {}).items():
        row(k, v)
