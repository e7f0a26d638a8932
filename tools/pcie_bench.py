"""PCIe-inclusive rate of the headline step: logits + labels start in PINNED HOST memory every step,
dlogits + per-list loss go back to pinned host memory.  Never bench.py's `value` (inputs resident in HBM there).
usage (through gpurun): python tools/pcie_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ranking_amd as ra
from ranking_amd.synthetic import make_batch

B, L, steps = 16384, 200, 50
labels, logits = make_batch(B, L, 4)
h_labels, h_logits = labels.pin_memory(), logits.pin_memory()
h_grad = torch.empty((B, L), dtype=torch.float32).pin_memory()
h_loss = torch.empty((), dtype=torch.float32).pin_memory()
dev = torch.device('cuda:0')
loss = ra.keras.losses.ApproxNDCGLoss()


def step():
    d_labels = h_labels.to(dev, non_blocking=True)
    d_logits = h_logits.to(dev, non_blocking=True)
    value, dlogits = loss.loss_and_grad(d_labels, d_logits)
    h_grad.copy_(dlogits, non_blocking=True)
    h_loss.copy_(value, non_blocking=True)


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
nbytes = 3 * B * L * 4
print('PCIe-inclusive: %.3f ms/step, %.1f M lists/s, %.1f GB/s over the link (%.1f MB per step)'
      % (dt * 1e3, B / dt / 1e6, nbytes / dt / 1e9, nbytes / 1e6))
