import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ranking_amd import _ops
dev='cuda'
def timeit(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/n
for B,L in [(16384,200),(4096,200),(16384,100)]:
    labels, logits = bench.make_inputs(B, L, 0, dev)
    order = _ops.list_order(labels)
    print(B, L, 'approx index order   ', timeit(lambda: _ops.approx_ndcg(logits, labels, balance=False)))
    print(B, L, 'approx given order   ', timeit(lambda: _ops.approx_ndcg(logits, labels, balance=order)))
    print(B, L, 'approx order per call', timeit(lambda: _ops.approx_ndcg(logits, labels, balance=True)))
    print(B, L, 'list_order alone     ', timeit(lambda: _ops.list_order(labels)))
