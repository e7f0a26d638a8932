"""Mirror of ``tensorflow_ranking/python/utils.py`` for the hot path.

Sorting goes through the gfx950 LDS bitonic kernel (``tfr_sort_ranks_f32``);
gathers of the sorted features are torch indexing (plumbing).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch

from . import _ops

_PADDING_LABEL = -1.          # utils.py:21
_PADDING_PREDICTION = -1e6    # utils.py:22
_PADDING_WEIGHT = 0.          # utils.py:23


def is_label_valid(labels):
    """utils.py:78-81."""
    return torch.as_tensor(labels) >= 0.


def gather_per_row(inputs, indices):
    """utils.py:38-75."""
    indices = indices.to(torch.int64)
    if inputs.dim() == 2:
        return torch.gather(inputs, 1, indices)
    idx = indices.reshape(indices.shape + (1,) * (inputs.dim() - 2)).expand(
        indices.shape + tuple(inputs.shape[2:]))
    return torch.gather(inputs, 1, idx)


# Op-level seeds behave like TensorFlow's (`tf.random.uniform(seed=s)`, utils.py:101): a seeded op owns a
# random STREAM that is fixed by (global seed, op seed) and ADVANCES on every call -- two calls with the same
# seed draw different numbers, a re-run of the program after `set_random_seed` draws the same sequence.  One
# persistent torch.Generator per (device, seed) gives exactly that; a fresh `manual_seed(seed)` per call would
# freeze "shuffled" orders to one permutation for the whole run.
_GLOBAL_SEED = 0
_STREAMS: Dict[tuple, torch.Generator] = {}


def set_random_seed(seed: int = 0) -> None:
    """`tf.random.set_seed`: restarts every seeded op stream (and re-keys them with ``seed``)."""
    global _GLOBAL_SEED
    _GLOBAL_SEED = int(seed)
    _STREAMS.clear()


def random_stream(seed: Optional[int], device) -> Optional[torch.Generator]:
    """The persistent generator of the op seed ``seed`` on ``device`` (None -> torch's global stream)."""
    if seed is None:
        return None
    device = torch.device(device)
    key = (device.type, device.index, int(seed))
    gen = _STREAMS.get(key)
    if gen is None:
        gen = torch.Generator(device=device)
        gen.manual_seed((_GLOBAL_SEED * 0x9E3779B1 + int(seed)) & 0x7FFFFFFFFFFFFFFF)
        _STREAMS[key] = gen
    return gen


def _tiebreak(shape, device, shuffle_ties, seed):
    if not shuffle_ties:
        return None
    return torch.randint(0, 32768, shape, dtype=torch.int32, device=device, generator=random_stream(seed, device))


def sort_by_scores(scores, features_list, topn=None, shuffle_ties=True, seed=None, mask=None):
    """utils.py:115-164.  Ties: random 15-bit secondary key when ``shuffle_ties``
    (the reference's shuffle is equally arbitrary), else lower index first."""
    scores = _ops.require_device(scores, 'scores').to(torch.float32)
    if scores.dim() != 2:
        raise ValueError('scores must have rank 2')
    list_size = scores.shape[1]
    topn = list_size if topn is None else min(int(topn), list_size)
    _, order = _ops.sort_ranks(scores, None, mask,
                               _tiebreak(scores.shape, scores.device, shuffle_ties, seed),
                               want_ranks=False, want_order=True)
    order = order[:, :topn]
    return [gather_per_row(f, order) for f in features_list]


def sorted_ranks(scores, shuffle_ties=True, seed=None):
    """utils.py:167-195: 1-based int32 ranks."""
    scores = _ops.require_device(scores, 'scores').to(torch.float32)
    ranks, _ = _ops.sort_ranks(scores, None, None,
                               _tiebreak(scores.shape, scores.device, shuffle_ties, seed),
                               want_ranks=True, want_order=False)
    return ranks


def is_ragged(x) -> bool:
    """A ragged batch is a python list/tuple of rows (lists or 1-D tensors)."""
    return isinstance(x, (list, tuple)) and len(x) > 0 and not torch.is_tensor(x) and all(
        isinstance(r, (list, tuple)) or (torch.is_tensor(r) and r.dim() == 1) for r in x)


def _pad_rows(rows, value, width, device):
    out = torch.full((len(rows), width), float(value), dtype=torch.float32, device=device)
    for i, r in enumerate(rows):
        r = torch.as_tensor(r, dtype=torch.float32, device=device)
        if r.numel():
            out[i, :r.numel()] = r
    return out


def ragged_to_dense(labels, predictions, weights, device=None):
    """utils.py:421-443.  Ragged tensors are lists of rows; returns dense
    (labels, predictions, weights, mask) with padding -1 / -1e6 / 0."""
    if device is None:
        device = next((r.device for r in list(labels) + list(predictions or [])
                       if torch.is_tensor(r)), None)
        if device is None:
            device = torch.device('cuda') if torch.cuda.is_available() else torch.device('cpu')
    width = max((len(r) for r in labels), default=0)
    mask = _pad_rows([[1.] * len(r) for r in labels], 0., width, device).to(torch.bool)
    dense_labels = _pad_rows(labels, _PADDING_LABEL, width, device)
    dense_pred = None if predictions is None else _pad_rows(predictions, _PADDING_PREDICTION, width, device)
    if is_ragged(weights) and all(len(w) == len(r) for w, r in zip(weights, labels)):
        weights = _pad_rows(weights, _PADDING_WEIGHT, width, device)
    elif weights is not None:
        weights = torch.as_tensor(weights, dtype=torch.float32, device=device)
    return dense_labels, dense_pred, weights, mask


def reshape_to_2d(tensor):
    """utils.py:273-284."""
    if tensor.dim() >= 3:
        return tensor.reshape(tensor.shape[0], tensor.shape[1])
    while tensor.dim() < 2:
        tensor = tensor.unsqueeze(-1)
    return tensor


def parse_keys_and_weights(key: str) -> Dict[str, float]:
    """utils.py:446-475: 'a:0.5,b:1' -> {'a': 0.5, 'b': 1.0}."""
    def _parse(pair):
        pair = pair.strip()
        if ':' not in pair:
            return pair, 1.0
        k, w = pair.split(':')
        return k.strip(), float(w.strip())
    return dict(_parse(p) for p in key.split(','))


def shuffle_valid_indices(is_valid, seed=None):
    """utils.py:198-200."""
    return organize_valid_indices(is_valid, shuffle=True, seed=seed)


def organize_valid_indices(is_valid, shuffle=True, seed=None):
    """utils.py:203-230: per-row order that puts valid entries first (in index
    order, or shuffled).  Returns [B, L] column indices (the nd batch index is
    implicit)."""
    is_valid = torch.as_tensor(is_valid, dtype=torch.bool)
    b, l = is_valid.shape
    if shuffle:
        values = torch.rand((b, l), device=is_valid.device, generator=random_stream(seed, is_valid.device))
    else:
        values = torch.arange(l - 1, -1, -1, dtype=torch.float32, device=is_valid.device).expand(b, l)
    rand = torch.where(is_valid, values, torch.full_like(values, -1e-6))
    return torch.sort(rand, dim=1, descending=True, stable=True).indices


def padded_nd_indices(is_valid, shuffle=False, seed=None):
    """utils.py:308-356: (indices [B, L], mask [B, L]); padding slots reuse the
    valid items circularly."""
    is_valid = torch.as_tensor(is_valid, dtype=torch.bool)
    b, l = is_valid.shape
    n_valid = is_valid.sum(dim=1, keepdim=True)
    pos = torch.arange(l, device=is_valid.device).unsqueeze(0).expand(b, l)
    mask = pos < n_valid
    circ = torch.remainder(pos, torch.clamp(n_valid, min=1))
    organized = organize_valid_indices(is_valid, shuffle=shuffle, seed=seed)
    return torch.gather(organized, 1, circ), mask
