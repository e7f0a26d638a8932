"""Torch-tensor level bindings of the groupwise-scoring entry points (include/tfr_hip.h, csrc/groupwise.hip):
group indices, the gather fused with the tower's bf16 input cast, and the scatter-average with its backward.
Device tensors only (no CPU fallback)."""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib
from . import _ops
from ._ops import _ptr, _stream, require_device


def _u8mask(t: torch.Tensor, name: str) -> torch.Tensor:
    require_device(t, name)
    if t.dtype == torch.bool:
        return t.contiguous().view(torch.uint8)
    if t.dtype != torch.uint8:
        t = (t != 0).view(torch.uint8)
    return t.contiguous()


def _i32(t: torch.Tensor, name: str) -> torch.Tensor:
    require_device(t, name)
    return t.to(torch.int32).contiguous()


def group_indices(is_valid: torch.Tensor, group_size: int, keys: Optional[torch.Tensor] = None
                  ) -> Tuple[torch.Tensor, torch.Tensor]:
    """model.py:205-244 for one shuffle: (idx int32 [B, L, group_size], mask bool [B, L]).  ``keys`` [B, L] fp32
    (U[0,1) draws) shuffle the valid items; None keeps them in index order."""
    m8 = _u8mask(is_valid, 'is_valid')
    B, L = m8.shape
    if keys is not None:
        keys = _ops._f32(keys, 'keys')
        if tuple(keys.shape) != (B, L):
            raise ValueError('keys must have the shape of is_valid')
    idx = torch.empty((B, L, int(group_size)), dtype=torch.int32, device=m8.device)
    gmask = torch.empty((B, L), dtype=torch.uint8, device=m8.device)
    _lib.check(_lib.load().tfr_group_indices_i32(_ptr(m8), _ptr(keys), B, L, int(group_size), _ptr(idx), _ptr(gmask),
                                                _stream()), 'tfr_group_indices_i32')
    return idx, gmask.view(torch.bool)


def group_gather_cast(x: torch.Tensor, idx: torch.Tensor, width: Optional[int] = None) -> torch.Tensor:
    """fp32 x [B, L, F], idx int32 [B, G, gs] -> bf16 [B * G, width] (default pad8(gs * F)): the MLP input of the
    groups, zero padded, gathered and converted in one pass."""
    require_device(x, 'x')
    x = x.to(torch.float32)
    if x.dim() != 3:
        raise ValueError('x must be [B, L, F]')
    if x.stride(2) != 1 or x.stride(0) != x.shape[1] * x.stride(1):
        x = x.contiguous()
    idx = _i32(idx, 'idx')
    B, L, F = x.shape
    if idx.dim() != 3 or idx.shape[0] != B:
        raise ValueError('idx must be [B, G, group_size]')
    G, gs = idx.shape[1], idx.shape[2]
    Kp = int(width) if width is not None else (gs * F + 7) // 8 * 8
    out = torch.empty((B * G, Kp), dtype=torch.bfloat16, device=x.device)
    _lib.check(_lib.load().tfr_group_gather_cast_f32_bf16(_ptr(x), x.stride(1), _ptr(idx), B, L, G, gs, F, Kp,
                                                         _ptr(out), _stream()), 'tfr_group_gather_cast_f32_bf16')
    return out


def group_scatter_avg(scores: torch.Tensor, idx: torch.Tensor, gmask: torch.Tensor, list_size: int,
                      want_counts: bool = True):
    """scores fp32 [B * G, gs] (or [B, G, gs]) -> (logits [B, L], counts [B, L] | None)."""
    scores = _ops._f32(scores, 'scores')
    idx = _i32(idx, 'idx')
    gm = _u8mask(gmask, 'gmask')
    B, G, gs = idx.shape
    if scores.numel() != B * G * gs or tuple(gm.shape) != (B, G):
        raise ValueError('scores / gmask do not match idx %s' % (tuple(idx.shape),))
    L = int(list_size)
    logits = torch.empty((B, L), dtype=torch.float32, device=scores.device)
    counts = torch.empty((B, L), dtype=torch.float32, device=scores.device) if want_counts else None
    _lib.check(_lib.load().tfr_group_scatter_avg_f32(_ptr(scores), _ptr(idx), _ptr(gm), B, L, G, gs, _ptr(logits),
                                                    _ptr(counts), _stream()), 'tfr_group_scatter_avg_f32')
    return logits, counts


def group_scatter_avg_bwd(dlogits: torch.Tensor, counts: torch.Tensor, idx: torch.Tensor, gmask: torch.Tensor
                          ) -> torch.Tensor:
    dlogits = _ops._f32(dlogits, 'dlogits'); counts = _ops._f32(counts, 'counts')
    idx = _i32(idx, 'idx')
    gm = _u8mask(gmask, 'gmask')
    B, G, gs = idx.shape
    L = dlogits.shape[1]
    out = torch.empty((B * G, gs), dtype=torch.float32, device=dlogits.device)
    _lib.check(_lib.load().tfr_group_scatter_avg_bwd_f32(_ptr(dlogits), _ptr(counts), _ptr(idx), _ptr(gm), B, L, G,
                                                        gs, _ptr(out), _stream()), 'tfr_group_scatter_avg_bwd_f32')
    return out


class GroupScatterAvgFn(torch.autograd.Function):
    """logits = scatter-average(scores) with the fused backward (one launch each way)."""

    @staticmethod
    def forward(ctx, scores, idx, gmask, list_size):
        logits, counts = group_scatter_avg(scores.detach(), idx, gmask, list_size, want_counts=True)
        ctx.save_for_backward(counts, idx, gmask)
        ctx.scores_shape = tuple(scores.shape)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        counts, idx, gmask = ctx.saved_tensors
        d = group_scatter_avg_bwd(dlogits.contiguous(), counts, idx, gmask)
        return d.reshape(ctx.scores_shape), None, None, None


_ops._guard_module(globals(), __name__)
