import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d.get('roofline', {})
print(sys.argv[1] if len(sys.argv) > 1 else '', d['metric'], 'ms/step %.5f' % d['ms_per_step'], 'value %.4g' % d['value'],
      'kernel_ms', r.get('kernel_ms'), 'frac %.4f' % r.get('frac', 0), 'valu_frac', r.get('valu_frac'))
