"""Mirror of ``tensorflow_ranking/python/keras/utils.py``: serialisable gain,
rank-discount and positive functions (keras/utils.py:51-135)."""
from __future__ import annotations

import math
from typing import Any, Callable, Dict, Optional

import torch

from .. import losses_impl as _li
from .. import metrics_impl as _mi

_REGISTRY: Dict[str, Any] = {}


def register_keras_serializable(package='tensorflow_ranking'):
    """Stand-in for tf.keras.utils.register_keras_serializable."""
    def deco(obj):
        _REGISTRY['%s>%s' % (package, obj.__name__)] = obj
        _REGISTRY[obj.__name__] = obj
        return obj
    return deco


def _t(x):
    return x if torch.is_tensor(x) else torch.as_tensor(x, dtype=torch.float32)


@register_keras_serializable()
def identity(label):
    """keras/utils.py:51-62."""
    return label


@register_keras_serializable()
def inverse(rank):
    """keras/utils.py:65-76: divide_no_nan(1, rank)."""
    rank = _t(rank)
    ok = rank != 0
    return torch.where(ok, 1. / torch.where(ok, rank, torch.ones_like(rank)), torch.zeros_like(rank))


@register_keras_serializable()
def pow_minus_1(label):
    """keras/utils.py:79-92: 2**x - 1."""
    label = _t(label)
    return torch.pow(torch.tensor(2.0, dtype=label.dtype, device=label.device), label) - 1.


@register_keras_serializable()
def log2_inverse(rank):
    """keras/utils.py:95-108: divide_no_nan(log 2, log1p(rank))."""
    rank = _t(rank)
    den = torch.log1p(rank)
    ok = den != 0
    # true division (python `scalar / tensor` is reciprocal*scalar in torch: 2 roundings)
    return torch.where(ok, torch.full_like(den, math.log(2.)) / torch.where(ok, den, torch.ones_like(den)),
                       torch.zeros_like(den))


@register_keras_serializable()
def is_greater_equal_1(label):
    """keras/utils.py:111-121."""
    return _t(label) >= 1.0


@register_keras_serializable()
def symmetric_log1p(t):
    """keras/utils.py:124-135."""
    t = _t(t)
    return torch.log1p(t * torch.sign(t)) * torch.sign(t)


# The fused kernels evaluate these in-register.
_li.register_gain_kind(identity, 0)
_li.register_gain_kind(pow_minus_1, 1)
_mi.register_pow2_gain(pow_minus_1)


def serialize_keras_object(obj):
    """keras/utils.py:26-33."""
    if obj is None:
        return None
    if callable(obj) and hasattr(obj, '__name__') and not hasattr(obj, 'get_config'):
        return obj.__name__
    return {'class_name': type(obj).__name__,
            'config': {k: serialize_keras_object(v) if callable(v) else v
                       for k, v in obj.get_config().items()}}


def deserialize_keras_object(config, custom_objects=None):
    """keras/utils.py:36-44."""
    if config is None:
        return None
    reg = dict(_REGISTRY)
    reg.update(custom_objects or {})
    if isinstance(config, str):
        if config not in reg:
            raise ValueError('Unknown object: %s' % config)
        return reg[config]
    if isinstance(config, dict) and 'class_name' in config:
        cls = reg.get(config['class_name'])
        if cls is None:
            raise ValueError('Unknown object: %s' % config['class_name'])
        cfg = {k: (deserialize_keras_object(v, custom_objects) if isinstance(v, str) and v in reg else v)
               for k, v in config.get('config', {}).items()}
        return cls(**cfg)
    return config
