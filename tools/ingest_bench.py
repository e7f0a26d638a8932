"""Host-fed scorer step: BASELINE config 2's tower (136-512-512-512-1, bf16, BatchNorm, Dropout 0.5)
forward + backward on B x L = 4096 x 100 rows whose FEATURES start in pinned host memory every step -- as fp32
(the reference's feed) and as bf16 (data.parse_from_example_list(example_dtype=torch.bfloat16)); the copy of step n + 1
runs on its own stream under the compute of step n (what data.Prefetcher does).  Prints the step time of both feeds,
the copy alone, and the compute alone.  Never bench.py's `value` (inputs resident in HBM there).
usage (through gpurun): python tools/ingest_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ranking_amd.tower import FusedTower

B, L, F, steps = 4096, 100, 136, 30
M = B * L
dev = torch.device('cuda:0')
torch.manual_seed(0)
tower = FusedTower(F, [512, 512, 512], 1, activation='relu', use_batch_norm=True, dropout=0.5).to(dev).train()
up = (torch.randn((M, 1)) / M ** 0.5).to(dev)
feats = torch.rand((M, F)) * 2 - 1


def timed(fn, n=steps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def compute(x):
    tower.zero_grad(set_to_none=True)
    tower(x).backward(up)


for tag, host in (('fp32', feats.pin_memory()), ('bf16', feats.to(torch.bfloat16).pin_memory())):
    bufs = [torch.empty_like(host, device=dev) for _ in range(2)]
    copied = [torch.cuda.Event() for _ in range(2)]
    consumed = [torch.cuda.Event() for _ in range(2)]
    side = torch.cuda.Stream()
    state = {'i': 0}
    for e in consumed:
        e.record()

    def fed():
        i = state['i'] & 1
        state['i'] += 1
        with torch.cuda.stream(side):
            side.wait_event(consumed[i])                  # the step that read this buffer two steps ago is done
            bufs[i].copy_(host, non_blocking=True)
            copied[i].record(side)
        torch.cuda.current_stream().wait_event(copied[i])
        compute(bufs[i])
        consumed[i].record()

    t_copy = timed(lambda: bufs[0].copy_(host, non_blocking=True))
    t_comp = timed(lambda: compute(bufs[0]))
    t_fed = timed(fed)
    nbytes = host.numel() * host.element_size()
    print('%s features: %6.1f MB per step; copy alone %.3f ms (%.1f GB/s), compute alone %.3f ms, fed step %.3f ms '
          '= %.2f M lists/s' % (tag, nbytes / 1e6, t_copy * 1e3, nbytes / t_copy / 1e9, t_comp * 1e3, t_fed * 1e3,
                                B / t_fed / 1e6))
