"""Mirror of the scorer contract of ``tensorflow_ranking/python/keras/model.py``:
``Scorer`` / ``UnivariateScorer`` (:691-777) and ``DNNScorer`` (:780-817)."""
from __future__ import annotations

import abc
from typing import Callable, Dict, List, Optional

import torch
from torch import nn

from . import layers
from .. import utils as _tfr_utils
from .. import _tower_ops


class UnivariateScorer(nn.Module, metaclass=abc.ABCMeta):
    """keras/model.py:712-777: flatten -> score every item -> restore."""

    def __init__(self):
        super().__init__()
        self._flatten = layers.FlattenList()
        self._restore = layers.RestoreList()

    @abc.abstractmethod
    def _score_flattened(self, context_features, example_features) -> torch.Tensor:
        raise NotImplementedError('Calling an abstract method.')

    def forward(self, context_features, example_features, mask) -> torch.Tensor:
        flat_ctx, flat_ex = self._flatten((context_features, example_features, mask))
        flat_logits = self._score_flattened(flat_ctx, flat_ex)
        return self._restore((flat_logits, mask))


class DNNScorer(UnivariateScorer):
    """keras/model.py:780-817: features concatenated in sorted-name order (:803-813)
    then ``create_tower`` (keras/layers.py:26-77)."""

    def __init__(self, input_dim: int, **dnn_kwargs):
        super().__init__()
        self._dnn_kwargs = dict(dnn_kwargs)
        self._dnn_kwargs.setdefault('output_units', 1)
        self._tower = layers.create_tower(input_dim=input_dim, **self._dnn_kwargs)

    def forward(self, context_features, example_features, mask) -> torch.Tensor:
        """keras/model.py:712-777.  With the fused tower and one dense example feature the flatten step's
        circular-padding gather (a full copy of the [B, L, F] tensor) is folded into the tower's input cast."""
        from ..tower import FusedTower
        _feed = (torch.float32, torch.bfloat16)               # (bf16: features the parser already rounded, data.py)
        if not context_features and len(example_features) > 1 and all(
                torch.is_tensor(v) and v.dim() >= 2 and v.dtype in _feed for v in example_features.values()) and len(
                {v.dtype for v in example_features.values()}) == 1:
            # Many per-column features (the reference's data: "1".."136"): concatenate ONCE, in the order
            # _score_flattened concatenates the flattened columns (:803-813) -- the flatten gather uses one index for
            # every feature, so flatten(concat) == concat(flatten) -- instead of one gather per feature and a concat.
            example_features = {'concat': torch.cat(
                [example_features[k].reshape(example_features[k].shape[0], example_features[k].shape[1], -1)
                 for k in sorted(example_features)], dim=2)}
        if (isinstance(self._tower, FusedTower) and not context_features and len(example_features) == 1):
            (x,) = example_features.values()
            if torch.is_tensor(x) and x.dim() == 3 and x.dtype in _feed:
                mask = torch.as_tensor(mask, device=x.device).to(torch.bool)
                b, l = mask.shape
                if l <= 8192 and b * l < 2 ** 31:
                    rows = _tower_ops.flatten_row_index(mask)        # one launch (utils.py:308-356)
                else:
                    idx, _ = _tfr_utils.padded_nd_indices(is_valid=mask)
                    rows = (idx + torch.arange(b, device=x.device).unsqueeze(1) * l).reshape(-1)
                flat_logits = self._tower(x.reshape(b * l, x.shape[2]), row_index=rows)
                return self._restore((flat_logits, mask))
        return super().forward(context_features, example_features, mask)

    def _score_flattened(self, context_features, example_features):
        ctx = [context_features[k].reshape(context_features[k].shape[0], -1)
               for k in sorted(context_features)]
        ex = [example_features[k].reshape(example_features[k].shape[0], -1)
              for k in sorted(example_features)]
        cols = ctx + ex
        return self._tower(cols[0] if len(cols) == 1 else torch.cat(cols, dim=1))
