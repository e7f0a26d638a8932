"""Bit-identity check of a rank-step variant of the LambdaRank group kernel (TFR_LAMBDARANK_BUCKET=0 / 1): the ranks
are integers, so everything downstream of them must come out bit for bit the same.  Run once per setting with the
same output directory, then with `compare`:
    TFR_LAMBDARANK_BUCKET=0 python tools/lbucket_check.py gpurun_out/x a
    TFR_LAMBDARANK_BUCKET=1 python tools/lbucket_check.py gpurun_out/x b
    python tools/lbucket_check.py gpurun_out/x compare"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

out_dir, what = sys.argv[1], sys.argv[2]
if what == 'compare':
    a, b = torch.load(os.path.join(out_dir, 'lb_a.pt')), torch.load(os.path.join(out_dir, 'lb_b.pt'))
    ok = True
    for k in a:
        same = all(torch.equal(x, y) for x, y in zip(a[k], b[k]))
        fin = all(bool(torch.isfinite(x).all()) for x in b[k])
        print('%-28s %s%s' % (k, 'bit-identical' if same else 'DIFFERENT', '' if fin else '  (non-finite values!)'))
        ok = ok and same and fin
    print('lbucket_check:', 'OK' if ok else 'FAILED')
    sys.exit(0 if ok else 1)

import ranking_amd as ra
from ranking_amd.synthetic import make_batch
dev = torch.device('cuda:0')
loss = ra.keras.losses.PairwiseLogisticLoss(lambda_weight=ra.keras.losses.NDCGLambdaWeight())
res = {}
for name, B, L, seed in (('bench 4096x200', 4096, 200, 3), ('1024x256', 1024, 256, 11), ('ties 600x150', 600, 150, 12),
                         ('outlier 512x200', 512, 200, 13), ('short 512x100', 512, 100, 14)):
    labels, logits = make_batch(B, L, seed)
    if name.startswith('ties'):
        logits = torch.round(logits * 8) / 8
    if name.startswith('outlier'):
        logits = logits * 1e-3
        logits[:, 3] = 50.0
    v, g = loss.loss_and_grad(labels.to(dev), logits.to(dev))
    torch.cuda.synchronize()
    res[name] = (v.detach().cpu().reshape(-1), g.detach().cpu())
torch.save(res, os.path.join(out_dir, 'lb_%s.pt' % what))
print('wrote', what, {k: float(v[0][0]) for k, v in res.items()})
