"""Mirror of the scorer contract of ``tensorflow_ranking/python/keras/model.py``:
``Scorer`` / ``UnivariateScorer`` (:691-777) and ``DNNScorer`` (:780-817)."""
from __future__ import annotations

import abc
from typing import Callable, Dict, List, Optional

import torch
from torch import nn

from . import layers


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

    def _score_flattened(self, context_features, example_features):
        ctx = [context_features[k].reshape(context_features[k].shape[0], -1)
               for k in sorted(context_features)]
        ex = [example_features[k].reshape(example_features[k].shape[0], -1)
              for k in sorted(example_features)]
        cols = ctx + ex
        return self._tower(cols[0] if len(cols) == 1 else torch.cat(cols, dim=1))
