"""Mirror of ``tensorflow_ranking/python/metrics_impl.py`` for the sort-based metrics: NDCG, MRR, DCG,
Hits, Recall, Precision, MAP and ARP.

``compute(labels, predictions, weights=None, mask=None)`` returns the same
``(per_list_metric [B, 1], per_list_weights [B, 1])`` pair as the reference; the
per-list work (validation, two masked sorts, DCG sums) is one gfx950 kernel
launch (``tfr_ndcg_metric_f32`` / ``tfr_mrr_metric_f32``).  Only the cross-list
"batch mean" fix-up of the list weights (metrics_impl.py:101-113) is done with
torch ops on ``[B]`` vectors.
"""
from __future__ import annotations

import abc
import math
from typing import Optional, Sequence

import torch

from . import _ops
from . import utils


def _pow2_minus_1(label):
    """metrics_impl.py:31 _DEFAULT_GAIN_FN."""
    return torch.pow(torch.tensor(2.0, dtype=label.dtype, device=label.device), label) - 1.


def _log2_inverse(rank):
    """metrics_impl.py:33 _DEFAULT_RANK_DISCOUNT_FN."""
    den = torch.log1p(rank)
    # true division (python `scalar / tensor` is reciprocal*scalar in torch: 2 roundings)
    return torch.full_like(den, math.log(2.)) / den


_DEFAULT_GAIN_FN = _pow2_minus_1
_DEFAULT_RANK_DISCOUNT_FN = _log2_inverse
_IN_KERNEL_GAINS = {_pow2_minus_1}


def register_pow2_gain(fn):
    """Declares that ``fn`` computes 2^label - 1 (evaluated in-kernel)."""
    _IN_KERNEL_GAINS.add(fn)


def _safe_div(num, den):
    ok = den != 0
    return torch.where(ok, num / torch.where(ok, den, torch.ones_like(den)), torch.zeros_like(num))


def per_list_weights_from_stats(stats):
    """metrics_impl.py:63-119 given per-list (sum w, sum rel, sum w*rel): one finishing launch."""
    return _ops.metric_list_weights(stats)


class _RankingMetric(object, metaclass=abc.ABCMeta):
    """metrics_impl.py:210-310.  Equal predictions: the reference sorts with ``shuffle_ties=True`` (utils.py:115-164, a
    random order of the tied items in every call); here ties keep index order unless ``shuffle_ties`` is set on the
    metric object -- then they are ordered by a counter-based hash of a tie seed, list and item (``seed``: None = a new
    seed per call from torch's host generator, an int = the same order in every call).  NDCG with ``shuffle_ties`` runs
    on the sort kernel instead of the counting forms."""

    shuffle_ties = False
    seed = None

    def __init__(self, ragged=False):
        self._ragged = ragged

    def _tie_seed(self):
        if not self.shuffle_ties:
            return 0
        if self.seed is None:
            from .losses_impl import _fresh_tie_seed      # a private generator: the caller's global random stream is not touched
            return _fresh_tie_seed()
        return (int(self.seed) & 0x7fffffff) or 1

    @property
    @abc.abstractmethod
    def name(self):
        raise NotImplementedError('Calling an abstract method.')

    def _prepare(self, labels, predictions, weights, mask):
        if any(utils.is_ragged(t) for t in (labels, predictions, weights)):
            if not self._ragged:
                raise ValueError('labels, predictions and/or weights are ragged tensors, '
                                 'use ragged=True to enable ragged support for metrics.')
            labels, predictions, weights, mask = utils.ragged_to_dense(labels, predictions, weights)
        predictions = _ops.require_device(torch.as_tensor(predictions), 'predictions').to(torch.float32)
        labels = torch.as_tensor(labels, dtype=torch.float32, device=predictions.device)
        if predictions.dim() != 2:
            raise ValueError('predictions must have rank 2')
        if labels.shape != predictions.shape:
            raise ValueError('labels %s and predictions %s are incompatible'
                             % (tuple(labels.shape), tuple(predictions.shape)))
        if weights is not None:
            weights = torch.as_tensor(weights, dtype=torch.float32, device=predictions.device)
        if mask is not None:
            mask = torch.as_tensor(mask, device=predictions.device).to(torch.bool)
        return labels, predictions, weights, mask

    @staticmethod
    def _no_items(predictions, n_cutoffs):
        """Lists without items (metrics_impl_test.py:1498-1506 feeds ``[[]]``): every sum over an empty list is 0, and
        so are the per-list weights; nothing to launch."""
        b = predictions.shape[0]
        return (torch.zeros((n_cutoffs, b), dtype=torch.float32, device=predictions.device),
                torch.zeros((b, 1), dtype=torch.float32, device=predictions.device))

    def compute(self, labels, predictions, weights=None, mask=None):
        labels, predictions, weights, mask = self._prepare(labels, predictions, weights, mask)
        if predictions.shape[1] == 0:
            out, w = self._no_items(predictions, 1)
        else:
            out, w = self._compute_multi(labels, predictions, weights, mask, [self._topn])
        return out[0].unsqueeze(1), w

    def compute_multi(self, labels, predictions, weights=None, mask=None,
                      topns: Sequence[Optional[int]] = (None,)):
        """Several cutoffs in ONE launch: returns ([K, B] metric, [B, 1] weights)."""
        labels, predictions, weights, mask = self._prepare(labels, predictions, weights, mask)
        if predictions.shape[1] == 0:
            return self._no_items(predictions, len(list(topns)))
        return self._compute_multi(labels, predictions, weights, mask, list(topns))


class MRRMetric(_RankingMetric):
    """metrics_impl.py:429-459."""

    def __init__(self, name, topn, ragged=False):
        super().__init__(ragged=ragged)
        self._name = name
        self._topn = topn

    @property
    def name(self):
        return self._name

    def _compute_multi(self, labels, predictions, weights, mask, topns):
        out, stats = _ops.mrr_metric(labels, predictions, weights, mask, topns, tie_seed=self._tie_seed())
        return out, per_list_weights_from_stats(stats)


class NDCGMetric(_RankingMetric):
    """metrics_impl.py:631-670."""

    def __init__(self, name, topn, gain_fn=_DEFAULT_GAIN_FN, rank_discount_fn=_DEFAULT_RANK_DISCOUNT_FN,
                 ragged=False):
        super().__init__(ragged=ragged)
        self._name = name
        self._topn = topn
        self._gain_fn = gain_fn
        self._rank_discount_fn = rank_discount_fn

    @property
    def name(self):
        return self._name

    def _compute_multi(self, labels, predictions, weights, mask, topns):
        gains = None
        if self._gain_fn not in _IN_KERNEL_GAINS:
            # metrics_impl.py:256-262: mask &= weights > 0; invalid labels -> 0.
            m = mask if mask is not None else labels >= 0
            if weights is not None:
                m = torch.logical_and(m, torch.broadcast_to(
                    weights if weights.dim() == 2 else weights.reshape(-1, 1), labels.shape) > 0)
            gains = self._gain_fn(torch.where(m, labels, torch.zeros_like(labels))).to(torch.float32)
        discount = _ops.rank_table(self._rank_discount_fn, labels.shape[1], labels.device)
        out, stats = _ops.ndcg_metric(labels, predictions, weights, mask, gains, discount, topns,
                                      tie_seed=self._tie_seed())
        return out, per_list_weights_from_stats(stats)


class _KindMetric(_RankingMetric):
    """Shared shape of the metrics served by ``tfr_rank_metric_f32`` (csrc/sort_metrics.hip)."""
    _KIND = None

    def __init__(self, name, topn, ragged=False):
        super().__init__(ragged=ragged)
        self._name = name
        self._topn = topn

    @property
    def name(self):
        return self._name

    def _compute_multi(self, labels, predictions, weights, mask, topns):
        out, stats = _ops.rank_metric(self._KIND, labels, predictions, weights, mask, topns, tie_seed=self._tie_seed())
        return out, per_list_weights_from_stats(stats)


class HitsMetric(_KindMetric):
    """metrics_impl.py:462-506."""
    _KIND = _ops.METRIC_HITS


class RecallMetric(_KindMetric):
    """metrics_impl.py:539-561."""
    _KIND = _ops.METRIC_RECALL


class PrecisionMetric(_KindMetric):
    """metrics_impl.py:564-586."""
    _KIND = _ops.METRIC_PRECISION


class MeanAveragePrecisionMetric(_KindMetric):
    """metrics_impl.py:589-628."""
    _KIND = _ops.METRIC_MAP


class ARPMetric(_KindMetric):
    """metrics_impl.py:509-536: the per-list weight is sum(label * weight) in sorted order."""
    _KIND = _ops.METRIC_ARP

    def __init__(self, name, ragged=False):
        super().__init__(name, None, ragged=ragged)

    def _compute_multi(self, labels, predictions, weights, mask, topns):
        out, stats = _ops.rank_metric(self._KIND, labels, predictions, weights, mask, [None], tie_seed=self._tie_seed())
        return out, stats[:, 2:3]


class OPAMetric(_KindMetric):
    """metrics_impl.py:708-743: ordered pair accuracy; the per-list weight is the sum of the pair weights."""
    _KIND = _ops.METRIC_OPA

    def __init__(self, name, ragged=False):
        super().__init__(name, None, ragged=ragged)

    def _compute_multi(self, labels, predictions, weights, mask, topns):
        out, stats = _ops.rank_metric(self._KIND, labels, predictions, weights, mask, [None])
        return out, stats[:, 2:3]


class BPrefMetric(_KindMetric):
    """metrics_impl.py:825-898."""

    def __init__(self, name, topn, use_trec_version=True, ragged=False):
        super().__init__(name, topn, ragged=ragged)
        self._use_trec_version = use_trec_version
        self._KIND = _ops.METRIC_BPREF if use_trec_version else _ops.METRIC_BPREF_NONTREC


class PWAMetric(_KindMetric):
    """metrics_impl.py:901-965: weights must be per list ([batch_size, 1])."""
    _KIND = _ops.METRIC_PWA

    def __init__(self, name, topn=5, ragged=False):
        super().__init__(name, topn, ragged=ragged)

    def compute(self, labels, predictions, weights=None, mask=None):
        if weights is not None and not utils.is_ragged(weights):
            w = torch.as_tensor(weights)
            if w.dim() != 2 or w.shape[1] != 1:
                raise ValueError('Weights should be a `Tensor` of the shape[batch_size, 1]')
        return super().compute(labels, predictions, weights, mask)

    def _compute_multi(self, labels, predictions, weights, mask, topns):
        out, _ = _ops.rank_metric(self._KIND, labels, predictions, weights, mask, topns, tie_seed=self._tie_seed())
        b = labels.shape[0]
        if weights is None:
            w = torch.ones((b, 1), dtype=torch.float32, device=labels.device)
        else:
            w = torch.broadcast_to(weights, labels.shape).mean(dim=1, keepdim=True)
        return out, w


class _DivRankingMetric(_RankingMetric):
    """metrics_impl.py:313-426: diversity metrics on subtopic labels [batch_size, list_size, subtopic_size],
    served by ``tfr_div_metric_f32`` (csrc/sort_metrics.hip)."""
    _KIND = None

    def __init__(self, name, topn=None, ragged=False):
        super().__init__(ragged=ragged)
        self._name = name
        self._topn = topn

    @property
    def name(self):
        return self._name

    def _prepare(self, labels, predictions, weights, mask):
        if self._ragged and utils.is_ragged(predictions):
            # ragged [B, (L), S]: pad the list dimension with -1 labels / -1e6 scores (utils.py:421-443)
            n_sub = max((len(r[0]) for r in labels if len(r)), default=1)
            dev = next((r.device for r in predictions if torch.is_tensor(r)), None)
            _, predictions, weights, mask = utils.ragged_to_dense([[0.] * len(r) for r in predictions], predictions,
                                                                  weights, device=dev)
            dense = torch.full((len(labels), predictions.shape[1], n_sub), -1.0, dtype=torch.float32,
                               device=predictions.device)
            for i, r in enumerate(labels):
                if len(r):
                    dense[i, :len(r)] = torch.as_tensor(r, dtype=torch.float32, device=predictions.device)
            labels = dense
        predictions = _ops.require_device(torch.as_tensor(predictions), 'predictions').to(torch.float32)
        labels = torch.as_tensor(labels, dtype=torch.float32, device=predictions.device)
        if predictions.dim() != 2 or labels.dim() != 3 or tuple(labels.shape[:2]) != tuple(predictions.shape):
            raise ValueError('labels must be [batch_size, list_size, subtopic_size], predictions [batch_size, '
                             'list_size]; got %s and %s' % (tuple(labels.shape), tuple(predictions.shape)))
        if weights is not None:
            weights = torch.as_tensor(weights, dtype=torch.float32, device=predictions.device)
        if mask is not None:
            mask = torch.as_tensor(mask, device=predictions.device).to(torch.bool)
            if mask.dim() == 3:
                mask = mask.any(dim=2)                      # :356-357
        return labels, predictions, weights, mask


class PrecisionIAMetric(_DivRankingMetric):
    """metrics_impl.py:746-782."""

    def _compute_multi(self, labels, predictions, weights, mask, topns):
        out, stats = _ops.div_metric(_ops.DIV_PRECISION_IA, labels, predictions, weights, mask, topns,
                                     tie_seed=self._tie_seed())
        return out, per_list_weights_from_stats(stats)


class AlphaDCGMetric(_DivRankingMetric):
    """metrics_impl.py:785-822 (``seed``: the reference's op seed of the tie shuffle; used here when ``shuffle_ties`` is
    set on the object, see ``_RankingMetric``)."""

    def __init__(self, name, topn, alpha=0.5, rank_discount_fn=_DEFAULT_RANK_DISCOUNT_FN, seed=None, ragged=False):
        super().__init__(name, topn, ragged=ragged)
        self._alpha = alpha
        self._rank_discount_fn = rank_discount_fn
        self._seed = seed
        self.seed = seed

    def _compute_multi(self, labels, predictions, weights, mask, topns):
        discount = _ops.rank_table(self._rank_discount_fn, predictions.shape[1], predictions.device)
        out, stats = _ops.div_metric(_ops.DIV_ALPHA_DCG, labels, predictions, weights, mask, topns, discount,
                                     self._alpha, tie_seed=self._tie_seed())
        plw = per_list_weights_from_stats(stats)
        return _safe_div(out, plw.reshape(1, -1)), plw


class DCGMetric(NDCGMetric):
    """metrics_impl.py:673-705."""

    def _compute_multi(self, labels, predictions, weights, mask, topns):
        gains = None
        if self._gain_fn not in _IN_KERNEL_GAINS:
            m = mask if mask is not None else labels >= 0
            if weights is not None:
                m = torch.logical_and(m, torch.broadcast_to(
                    weights if weights.dim() == 2 else weights.reshape(-1, 1), labels.shape) > 0)
            gains = self._gain_fn(torch.where(m, labels, torch.zeros_like(labels))).to(torch.float32)
        discount = _ops.rank_table(self._rank_discount_fn, labels.shape[1], labels.device)
        out, stats = _ops.rank_metric(_ops.METRIC_DCG, labels, predictions, weights, mask, topns, gains, discount,
                                      tie_seed=self._tie_seed())
        plw = per_list_weights_from_stats(stats)
        return _safe_div(out, plw.reshape(1, -1)), plw
