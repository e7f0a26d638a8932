"""Mirror of the groupwise multi-item scoring of ``tensorflow_ranking/python/model.py``
(`_rolling_window_indices` :164-202, `_form_group_indices_nd` :205-244,
`_GroupwiseRankingModel._compute_logits_impl` :341-421).

Groups are rolling windows of ``group_size`` consecutive items (mod n_valid) over
a valid-first ordering of the list; every group is scored by ``group_score_fn``
and an item's logit is the average of the scores that landed on it.  The
gather / scatter-average is index plumbing on device tensors; the GEMMs inside
``group_score_fn`` are where the time goes (config 5: 272-512-512-512-2).
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import torch

from . import _group_ops
from . import utils


def _rolling_window_indices(size: int, rw_size: int, num_valid_entries) -> Tuple[torch.Tensor, torch.Tensor]:
    """model.py:164-202."""
    n = torch.as_tensor(num_valid_entries).reshape(-1)
    dev = n.device
    rw = torch.arange(rw_size, device=dev).unsqueeze(0) + torch.arange(size, device=dev).unsqueeze(1)
    batch_rw = rw.unsqueeze(0).expand(n.shape[0], size, rw_size)
    mask = batch_rw.min(dim=2).values < n.reshape(-1, 1)
    n1 = torch.where(n < 1, torch.ones_like(n), n)
    return torch.remainder(batch_rw, n1.reshape(-1, 1, 1)), mask


def _form_group_indices_nd(is_valid, group_size: int, shuffle: bool = False, seed: Optional[int] = None):
    """model.py:205-244: ([B, G, group_size] item indices, [B, G] mask).  The batch coordinate of the reference's
    nd-indices is implicit.  Device tensors: ONE launch of ``tfr_group_indices_i32`` (int32 indices; the shuffle keys
    are a ``torch.rand`` draw from the op seed's persistent stream); host tensors: the same arithmetic in torch index
    ops (int64 indices) -- host-side plumbing like the rest of ``utils``."""
    is_valid = torch.as_tensor(is_valid).to(torch.bool)
    b, l = is_valid.shape
    if is_valid.is_cuda:
        keys = torch.rand((b, l), device=is_valid.device, generator=utils.random_stream(seed, is_valid.device)) \
            if shuffle else None
        return _group_ops.group_indices(is_valid, group_size, keys)
    n_valid = is_valid.sum(dim=1)
    rw, mask = _rolling_window_indices(l, group_size, n_valid)
    organized = utils.organize_valid_indices(is_valid, shuffle=shuffle, seed=seed)
    idx = torch.gather(organized.unsqueeze(1).expand(b, l, l), 2, rw)
    return idx, mask


class FusedGroupScoreFn(torch.nn.Module):
    """``group_score_fn`` for the standard groupwise DNN (examples/tf_ranking_libsvm.py-style score function with
    group_size > 1): the example features of the ``group_size`` members, each member's features concatenated in
    sorted-name order (keras/model.py:803-813), flattened member-major into one ``[rows, group_size * F]`` matrix and
    scored by a ``create_tower`` MLP with ``output_units == group_size``.

    Called with the reference protocol ``fn(context_features, group_features) -> [rows, group_size]`` it is an
    ordinary score function.  ``GroupwiseScorer`` recognises it and, on a HIP device with a fused bf16 tower and no
    context features, skips the ``[B * G, group_size, F]`` fp32 intermediate: the group gather happens inside the
    tower's input cast (``tfr_group_gather_cast_f32_bf16``)."""

    def __init__(self, tower: torch.nn.Module, feature_names=None):
        super().__init__()
        self.tower = tower
        self.feature_names = list(feature_names) if feature_names is not None else None

    def names(self, features):
        return self.feature_names if self.feature_names is not None else sorted(features)

    def forward(self, context_features, group_features):
        names = self.names(group_features)
        parts = [group_features[n] for n in names]
        rows, gs = parts[0].shape[0], parts[0].shape[1]
        x = parts[0].reshape(rows, gs, -1) if len(parts) == 1 else \
            torch.cat([p.reshape(rows, gs, -1) for p in parts], dim=2)
        x = x.reshape(rows, -1)
        ctx = [context_features[n].reshape(rows, -1) for n in sorted(context_features or {})]
        if ctx:
            x = torch.cat(ctx + [x], dim=1)
        return self.tower(x)


class GroupwiseScorer(torch.nn.Module):
    """model.py:276-421: ``group_score_fn(context, group_features) -> [B*G, group_size]``."""

    def __init__(self, group_score_fn: Callable, group_size: int, num_shuffles: Optional[int] = None):
        super().__init__()
        if group_size <= 0:
            raise ValueError('Invalid group_size %d' % group_size)                     # model.py:303-304
        self._score_fn = group_score_fn
        self._group_size = group_size
        self._num_shuffles = num_shuffles

    def _indices(self, is_valid, training: bool):
        """model.py:313-339."""
        if self._group_size == 1:
            shuffle, n = False, 1
        elif not training:
            shuffle, n = self._num_shuffles is not None, self._num_shuffles or 1
        else:
            shuffle, n = True, self._num_shuffles or 1
        # op seeds 77 + i like the reference (:330-334); each owns a stream that advances from step to step
        parts = [_form_group_indices_nd(is_valid, self._group_size, shuffle=shuffle, seed=i + 77)
                 for i in range(n)]
        if n == 1:
            return parts[0]
        return torch.cat([p[0] for p in parts], dim=1), torch.cat([p[1] for p in parts], dim=1)

    def group_indices(self, is_valid, shuffle: Optional[bool] = None):
        """model.py:313-339: the ([B, G, group_size] indices, [B, G] mask) of a batch.  They depend on
        the validity mask only, so a caller can build them once per batch (e.g. outside a captured
        hipGraph: the shuffle draws from a torch.Generator) and pass them to ``forward``."""
        is_valid = torch.as_tensor(is_valid).to(torch.bool)
        if shuffle is False:
            return _form_group_indices_nd(is_valid, self._group_size, shuffle=False)
        return self._indices(is_valid, self.training)

    def _fused_input(self, context_features, example_features):
        """The tower and the [B, L, F] feature tensor of the fused path, or None."""
        fn = self._score_fn
        if not isinstance(fn, FusedGroupScoreFn) or context_features:
            return None
        from .tower import FusedTower
        if not isinstance(fn.tower, FusedTower) or fn.tower.input_batch_norm:
            return None      # input BatchNorm needs the raw fp32 group features (batch statistics): op-by-op gather
        names = fn.names(example_features)
        parts = [example_features[n] for n in names]
        b, l = parts[0].shape[0], parts[0].shape[1]
        x = parts[0].reshape(b, l, -1) if len(parts) == 1 else torch.cat([p.reshape(b, l, -1) for p in parts], dim=2)
        if x.requires_grad or self._group_size * x.shape[2] != fn.tower.input_dim:
            return None
        return fn.tower, x

    def forward(self, context_features: Dict[str, torch.Tensor], example_features: Dict[str, torch.Tensor],
                is_valid, shuffle: Optional[bool] = None, group_indices=None) -> torch.Tensor:
        is_valid = torch.as_tensor(is_valid).to(torch.bool)
        b, l = is_valid.shape
        idx, mask = group_indices if group_indices is not None else self.group_indices(is_valid, shuffle)
        g, gs = idx.shape[1], self._group_size
        on_device = is_valid.is_cuda
        fused = self._fused_input(context_features, example_features) if on_device else None
        if fused is not None:
            # the gather of the group features IS the tower's input cast: bf16 [B * G, gs * F (+ k-step padding)]
            tower, x = fused
            from . import _tower_ops
            scores = tower(_group_ops.group_gather_cast(x, idx, width=_tower_ops.pad_k(gs * x.shape[2])))
        else:
            big_ctx = {k: v.unsqueeze(1).expand((b, g) + tuple(v.shape[1:])).reshape((b * g,) + tuple(v.shape[1:]))
                       for k, v in (context_features or {}).items()}
            big_ex = {}
            idx64 = idx.to(torch.int64)
            for k, v in example_features.items():
                f = v.reshape(b, l, -1)
                gathered = torch.gather(f.unsqueeze(1).expand(b, g, l, f.shape[2]), 2,
                                        idx64.unsqueeze(-1).expand(b, g, gs, f.shape[2]))
                big_ex[k] = gathered.reshape(b * g, gs, f.shape[2])
            scores = self._score_fn(big_ctx, big_ex)
        if on_device:     # the two scatter_nd + div_no_nan (and their backward) as one launch each way
            return _group_ops.GroupScatterAvgFn.apply(scores.reshape(b * g, gs).to(torch.float32), idx, mask, l)
        scores = scores.reshape(b, g, gs)
        scores_mask = mask.unsqueeze(2).expand(b, g, gs)
        flat_idx = idx.reshape(b, g * gs).to(torch.int64)
        counts = torch.zeros((b, l), dtype=scores.dtype, device=scores.device).scatter_add_(
            1, flat_idx, scores_mask.reshape(b, -1).to(scores.dtype))
        scores = torch.where(scores_mask, scores, torch.zeros_like(scores))
        logits = torch.zeros((b, l), dtype=scores.dtype, device=scores.device).scatter_add(
            1, flat_idx, scores.reshape(b, -1))
        return torch.where(counts != 0, logits / torch.where(counts != 0, counts, torch.ones_like(counts)),
                           torch.zeros_like(logits))
