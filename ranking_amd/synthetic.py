"""Synthetic inputs of the BASELINE configurations: shared by bench.py, smoke() and the tests.

``features ~ U(-1, 1)``, valid length ``n_b ~ U{ceil(L/2)..L}``, valid labels ``randint{0..4}`` as fp32, padding
label ``-1``, ``logits ~ N(0, 1)`` with no tied values per list (the reference breaks ties randomly, so parity is
defined on tie-free inputs only).  Generated on the host from ``torch.Generator().manual_seed(seed)``.
"""
import math

import torch


def make_batch(B, L, seed, min_frac=0.5, max_label=4, device='cpu', full=False):
    """labels[b, :n_b] ~ randint{0..max_label} (fp32), labels[b, n_b:] = -1 with
    n_b ~ U{ceil(L*min_frac)..L}; logits ~ N(0,1) fp32 with no tied values per list."""
    g = torch.Generator().manual_seed(seed)
    lo = max(1, math.ceil(L * min_frac))
    n = torch.randint(lo, L + 1, (B,), generator=g) if not full else torch.full((B,), L)
    labels = torch.randint(0, max_label + 1, (B, L), generator=g).to(torch.float32)
    pos = torch.arange(L).unsqueeze(0)
    labels = torch.where(pos < n.unsqueeze(1), labels, torch.full_like(labels, -1.0))
    logits = torch.randn((B, L), generator=g, dtype=torch.float32)
    # tie-free per list (the reference breaks ties randomly: parity is undefined on ties)
    for _ in range(8):
        srt = torch.sort(logits, dim=1).values
        dup = (srt[:, 1:] == srt[:, :-1]).any(dim=1)
        if not dup.any():
            break
        logits[dup] = torch.randn((int(dup.sum()), L), generator=g, dtype=torch.float32)
    return labels.to(device), logits.to(device)


def make_weights(B, L, seed, device='cpu'):
    g = torch.Generator().manual_seed(seed + 7919)
    return (torch.rand((B, L), generator=g) * 2.0 + 0.25).to(device)
