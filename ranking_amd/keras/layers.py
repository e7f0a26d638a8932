"""Mirror of the scorer pieces of ``tensorflow_ranking/python/keras/layers.py``:
``create_tower`` (:26-77), ``FlattenList`` (:81-182), ``RestoreList`` (:186-272).

The tower is a stack of Dense -> BatchNorm -> activation -> Dropout blocks on the
flattened ``[B*L, F]`` matrix.  ``compute_dtype=torch.bfloat16`` (config 2 of
BASELINE.json) builds the fused MFMA tower (``ranking_amd/tower.py`` over
``csrc/tower.hip``) when the shapes allow it; ``torch.float32`` -- what the
reference computes in -- and the shapes the fused tower refuses build the layer
stack below with every Dense on the fp32 matrix-core kernels of
``csrc/gemm_f32.hip`` (``ranking_amd/scorer.py``).
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Tuple

import torch
from torch import nn

from .. import utils as _tfr_utils

_EPSILON = 1e-10


class _Activation(nn.Module):
    def __init__(self, fn):
        super().__init__()
        self._fn = fn

    def forward(self, x):
        return x if self._fn is None else self._fn(x)


class _LayerStack(nn.Sequential):
    """The unfused Dense -> BatchNorm -> activation -> Dropout stack.  Keras layers act on the LAST axis of an input
    of any rank (keras/layers_test.py:25-30 feeds [batch, list, 1]); torch's BatchNorm1d reads axis 1 of a 3-D input
    as the channels, so inputs are flattened to [rows, features] here and restored."""

    def forward(self, x):
        if x.dim() <= 2:
            return super().forward(x)
        lead = tuple(x.shape[:-1])
        y = super().forward(x.reshape(-1, x.shape[-1]))
        return y.reshape(lead + (y.shape[-1],))


def create_tower(hidden_layer_dims: List[int], output_units: int, activation: Optional[Callable] = None,
                 input_batch_norm: bool = False, use_batch_norm: bool = True,
                 batch_norm_moment: float = 0.999, dropout: float = 0.5, name: Optional[str] = None,
                 input_dim: Optional[int] = None, compute_dtype: torch.dtype = torch.float32,
                 **kwargs) -> nn.Module:
    """keras/layers.py:26-77.  ``input_dim`` is required up front (torch layers are
    not lazily shaped); Keras' BatchNormalization(momentum=m, epsilon=1e-3) maps to
    torch BatchNorm1d(momentum=1-m, eps=1e-3)."""
    from ..scorer import make_dense
    if input_dim is None:
        raise ValueError('input_dim is required')
    if compute_dtype == torch.bfloat16 and hidden_layer_dims:
        # MI355X fast path: one fused MFMA launch per layer, forward and backward (ranking_amd/tower.py).
        from ..tower import FusedTower, _act_code
        try:
            act = _act_code(activation)
            fusable = all(int(h) % 8 == 0 for h in hidden_layer_dims) and 1 <= int(output_units) <= 4
        except ValueError:
            fusable = False
        if fusable:
            return FusedTower(input_dim, list(hidden_layer_dims), output_units, activation=act,
                              use_batch_norm=use_batch_norm, batch_norm_moment=batch_norm_moment,
                              dropout=dropout or 0.0, input_batch_norm=input_batch_norm)
    layers: List[nn.Module] = []
    if input_batch_norm:
        layers.append(nn.BatchNorm1d(input_dim, momentum=1.0 - batch_norm_moment, eps=1e-3))
    width = input_dim
    for layer_width in hidden_layer_dims:
        layers.append(make_dense(width, layer_width, compute_dtype))
        if use_batch_norm:
            layers.append(nn.BatchNorm1d(layer_width, momentum=1.0 - batch_norm_moment, eps=1e-3))
        layers.append(_Activation(activation))
        if dropout:
            layers.append(nn.Dropout(p=dropout))
        width = layer_width
    layers.append(make_dense(width, output_units, compute_dtype))
    return _LayerStack(*layers)


class FlattenList(nn.Module):
    """keras/layers.py:81-182."""

    def __init__(self, circular_padding: bool = True, name: Optional[str] = None, **kwargs):
        super().__init__()
        self._circular_padding = circular_padding

    def forward(self, inputs: Tuple[Dict[str, torch.Tensor], Dict[str, torch.Tensor], torch.Tensor]):
        context_features, example_features, list_mask = inputs
        if not example_features:
            raise ValueError('Need a valid example feature.')
        list_mask = torch.as_tensor(list_mask).to(torch.bool)
        b, l = list_mask.shape
        flat_ctx = {}
        for name, t in (context_features or {}).items():
            flat_ctx[name] = t.unsqueeze(1).expand((b, l) + tuple(t.shape[1:])).reshape(
                (b * l,) + tuple(t.shape[1:]))
        idx = None
        if self._circular_padding:
            idx, _ = _tfr_utils.padded_nd_indices(is_valid=list_mask)
        flat_ex = {}
        for name, t in example_features.items():
            if idx is not None:
                t = _tfr_utils.gather_per_row(t, idx.to(t.device))
            flat_ex[name] = t.reshape((b * l,) + tuple(t.shape[2:]))
        return flat_ctx, flat_ex

    def get_config(self):
        return {'circular_padding': self._circular_padding}


class RestoreList(nn.Module):
    """keras/layers.py:186-272."""

    def __init__(self, name: Optional[str] = None, by_scatter: bool = False, **kwargs):
        super().__init__()
        self._by_scatter = by_scatter

    def forward(self, inputs: Tuple[torch.Tensor, torch.Tensor]):
        flattened_logits, list_mask = inputs
        list_mask = torch.as_tensor(list_mask).to(torch.bool)
        try:
            logits = flattened_logits.reshape(list_mask.shape)
        except RuntimeError:
            raise ValueError('`flattened_logits` needs to be either 1D of [batch_size * list_size] or '
                             '2D of [batch_size * list_size, 1].')
        fill = torch.full_like(logits, math.log(_EPSILON))
        if self._by_scatter:
            idx, _ = _tfr_utils.padded_nd_indices(is_valid=list_mask)
            counts = torch.zeros_like(logits).scatter_add_(1, idx, torch.ones_like(logits))
            summed = torch.zeros_like(logits).scatter_add_(1, idx, logits)
            return torch.where(counts > 0., summed / torch.clamp(counts, min=1.), fill)
        return torch.where(list_mask, logits, fill)

    def get_config(self):
        return {'by_scatter': self._by_scatter}
