"""Mirror of ``tensorflow_ranking/python/keras/metrics.py`` for NDCG and MRR.

``tf.keras.metrics.Mean`` semantics (keras/metrics.py:156-193): each
``update_state`` adds sum(value * weight) and sum(weight) of the per-list
metric; ``result()`` is their ratio.  Accumulators are device scalars (no host
sync per step).  Under data parallelism ``result()`` all-reduces the two sums
 when ``torch.distributed`` is initialised.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional

import torch

from .. import metrics_impl
from . import utils


class RankingMetricKey(object):
    """keras/metrics.py:31-66."""
    MRR = 'mrr'
    ARP = 'arp'
    NDCG = 'ndcg'
    DCG = 'dcg'
    PRECISION = 'precision'
    MAP = 'map'
    ORDERED_PAIR_ACCURACY = 'ordered_pair_accuracy'
    ALPHA_DCG = 'alpha_dcg'
    PRECISION_IA = 'precision_ia'
    HITS = 'hits'
    RECALL = 'recall'


def get(key: str, name: Optional[str] = None, dtype=None, topn: Optional[int] = None, **kwargs):
    """keras/metrics.py:69-128."""
    if not isinstance(key, str):
        raise ValueError('Input `key` needs to be string.')
    key_to_cls = {RankingMetricKey.MRR: MRRMetric, RankingMetricKey.NDCG: NDCGMetric,
                  RankingMetricKey.DCG: DCGMetric, RankingMetricKey.ARP: ARPMetric,
                  RankingMetricKey.PRECISION: PrecisionMetric, RankingMetricKey.MAP: MeanAveragePrecisionMetric,
                  RankingMetricKey.HITS: HitsMetric, RankingMetricKey.RECALL: RecallMetric,
                  RankingMetricKey.ORDERED_PAIR_ACCURACY: OPAMetric, RankingMetricKey.ALPHA_DCG: AlphaDCGMetric,
                  RankingMetricKey.PRECISION_IA: PrecisionIAMetric}
    metric_kwargs = {'name': name, 'dtype': dtype}
    if topn:
        metric_kwargs.update({'topn': topn})
    if kwargs:
        metric_kwargs.update(kwargs)
    if key in key_to_cls:
        return key_to_cls[key](**metric_kwargs)
    raise ValueError('Unsupported metric: {}'.format(key))


def default_keras_metrics(**kwargs) -> List['_RankingMetric']:
    """keras/metrics.py:131-153, restricted to the metrics on the hot path
    (NDCG@{1,3,5,10,all}, MRR); the remaining sort-based metrics are SURVEY 8f."""
    list_kwargs = [dict(key='ndcg', topn=topn, name='metric/ndcg_{}'.format(topn), **kwargs)
                   for topn in [1, 3, 5, 10]]
    list_kwargs += [dict(key='mrr', name='metric/mrr', **kwargs),
                    dict(key='ndcg', name='metric/ndcg', **kwargs)]
    return [get(**kw) for kw in list_kwargs]


def update_metrics(metrics, y_true, y_pred, sample_weight=None):
    """``m.update_state(y_true, y_pred, sample_weight)`` for every metric object, with the objects that differ ONLY in their
    cut-off (same class, gain / discount functions, tie handling) served by ONE kernel launch and ONE per-list-weights launch
    (``compute_multi``: up to 8 cut-offs per pass over the lists) -- round 6, VERDICT r5 next #7: ``default_keras_metrics()``
    is NDCG@{1, 3, 5, 10, all} + MRR; updated one object at a time that is five passes over the batch and five launches of
    the batch-mean weight for the same answer.  The accumulated (total, count) of every object are the ones `update_state`
    produces: the same per-list values and weights, summed by the same torch reductions."""
    groups: Dict[Any, list] = {}
    singles = []
    for m in metrics:
        impl = getattr(m, '_metric', None)
        key = None
        if impl is not None and hasattr(impl, '_compute_multi') and hasattr(impl, '_topn') and type(m).update_state is _RankingMetric.update_state:
            key = (type(impl), getattr(impl, '_gain_fn', None), getattr(impl, '_rank_discount_fn', None),
                   getattr(impl, '_ragged', False), impl.shuffle_ties, impl.seed)
            if impl.shuffle_ties and impl.seed is None:
                key = None                                  # a fresh random tie order per call and object: not shareable
        (groups.setdefault(key, []) if key is not None else singles).append(m)
    for key, ms in groups.items():
        if len(ms) == 1:
            singles.append(ms[0])
            continue
        for i in range(0, len(ms), 8):                       # TFR_MAX_TOPN cut-offs per launch
            chunk = ms[i:i + 8]
            out, w = chunk[0]._metric.compute_multi(y_true, y_pred, sample_weight, topns=[m._metric._topn for m in chunk])
            totals = (out * w.reshape(1, -1)).sum(dim=1)     # [K]
            count = w.sum()
            for j, m in enumerate(chunk):
                m._accumulate(totals[j], count)
    for m in singles:
        m.update_state(y_true, y_pred, sample_weight)
    return metrics


class _RankingMetric(object):
    """keras/metrics.py:156-201."""

    def __init__(self, name=None, dtype=None, ragged=False, **kwargs):
        self.name = name
        self._dtype = dtype or torch.float32
        self._metric = None
        self._ragged = ragged
        self.total = None
        self.count = None

    def _accumulate(self, t, c):
        if self.total is None:
            self.total, self.count = t, c
        else:
            self.total = self.total + t
            self.count = self.count + c

    def update_state(self, y_true, y_pred, sample_weight=None):
        val, w = self._metric.compute(y_true, y_pred, sample_weight)
        self._accumulate((val * w).sum(), w.sum())
        return self

    def __call__(self, y_true, y_pred, sample_weight=None):
        return self.update_state(y_true, y_pred, sample_weight).result()

    def result(self, sync: bool = True):
        if self.total is None:
            return torch.zeros((), dtype=torch.float32)
        total, count = self.total, self.count
        if sync and torch.distributed.is_available() and torch.distributed.is_initialized() \
                and torch.distributed.get_world_size() > 1:
            buf = torch.stack([total, count])
            torch.distributed.all_reduce(buf)
            total, count = buf[0], buf[1]
        return torch.where(count != 0, total / torch.where(count != 0, count, torch.ones_like(count)),
                           torch.zeros_like(total))

    def reset_state(self):
        self.total = None
        self.count = None

    reset_states = reset_state

    def get_config(self) -> Dict[str, Any]:
        return {'name': self.name, 'dtype': self._dtype, 'ragged': self._ragged}

    @classmethod
    def from_config(cls, config):
        return cls(**config)


@utils.register_keras_serializable()
class MRRMetric(_RankingMetric):
    """keras/metrics.py:204-268."""

    def __init__(self, name=None, topn=None, dtype=None, ragged=False, **kwargs):
        super().__init__(name=name, dtype=dtype, ragged=ragged, **kwargs)
        self._topn = topn
        self._metric = metrics_impl.MRRMetric(name=name, topn=topn, ragged=ragged)

    def get_config(self):
        config = super().get_config()
        config.update({'topn': self._topn})
        return config


@utils.register_keras_serializable()
class NDCGMetric(_RankingMetric):
    """keras/metrics.py:710-800."""

    def __init__(self, name=None, topn=None, gain_fn=None, rank_discount_fn=None, dtype=None,
                 ragged=False, **kwargs):
        super().__init__(name=name, dtype=dtype, ragged=ragged, **kwargs)
        self._topn = topn
        self._gain_fn = gain_fn or utils.pow_minus_1
        self._rank_discount_fn = rank_discount_fn or utils.log2_inverse
        self._metric = metrics_impl.NDCGMetric(name=name, topn=topn, gain_fn=self._gain_fn,
                                               rank_discount_fn=self._rank_discount_fn, ragged=ragged)

    def get_config(self):
        config = super().get_config()
        config.update({'topn': self._topn, 'gain_fn': self._gain_fn,
                       'rank_discount_fn': self._rank_discount_fn})
        return config


def _topn_metric(impl_cls, doc):
    class _M(_RankingMetric):
        __doc__ = doc

        def __init__(self, name=None, topn=None, dtype=None, ragged=False, **kwargs):
            super().__init__(name=name, dtype=dtype, ragged=ragged, **kwargs)
            self._topn = topn
            self._metric = impl_cls(name=name, topn=topn, ragged=ragged)

        def get_config(self):
            config = super().get_config()
            config.update({'topn': self._topn})
            return config
    return _M


HitsMetric = utils.register_keras_serializable()(type('HitsMetric', (_topn_metric(
    metrics_impl.HitsMetric, 'keras/metrics.py:269-332.'),), {}))
PrecisionMetric = utils.register_keras_serializable()(type('PrecisionMetric', (_topn_metric(
    metrics_impl.PrecisionMetric, 'keras/metrics.py:383-456.'),), {}))
RecallMetric = utils.register_keras_serializable()(type('RecallMetric', (_topn_metric(
    metrics_impl.RecallMetric, 'keras/metrics.py:459-531.'),), {}))
MeanAveragePrecisionMetric = utils.register_keras_serializable()(type('MeanAveragePrecisionMetric', (_topn_metric(
    metrics_impl.MeanAveragePrecisionMetric, 'keras/metrics.py:629-707.'),), {}))


@utils.register_keras_serializable()
class ARPMetric(_RankingMetric):
    """keras/metrics.py:335-380."""

    def __init__(self, name=None, dtype=None, ragged=False, **kwargs):
        super().__init__(name=name, dtype=dtype, ragged=ragged, **kwargs)
        self._metric = metrics_impl.ARPMetric(name=name, ragged=ragged)


PrecisionIAMetric = utils.register_keras_serializable()(type('PrecisionIAMetric', (_topn_metric(
    metrics_impl.PrecisionIAMetric, 'keras/metrics.py:534-626 (y_true: [batch, list, subtopic]).'),), {}))


@utils.register_keras_serializable()
class AlphaDCGMetric(_RankingMetric):
    """keras/metrics.py:886-1010 (y_true: [batch, list, subtopic])."""

    def __init__(self, name='alpha_dcg_metric', topn=None, alpha=0.5, rank_discount_fn=None, seed=None, dtype=None,
                 ragged=False, **kwargs):
        super().__init__(name=name, dtype=dtype, ragged=ragged, **kwargs)
        self._topn = topn
        self._alpha = alpha
        self._rank_discount_fn = rank_discount_fn or utils.log2_inverse
        self._seed = seed
        self._metric = metrics_impl.AlphaDCGMetric(name=name, topn=topn, alpha=alpha,
                                                   rank_discount_fn=self._rank_discount_fn, seed=seed, ragged=ragged)

    def get_config(self):
        config = super().get_config()
        config.update({'topn': self._topn, 'alpha': self._alpha, 'rank_discount_fn': self._rank_discount_fn,
                       'seed': self._seed})
        return config


@utils.register_keras_serializable()
class OPAMetric(_RankingMetric):
    """keras/metrics.py:1013-1060."""

    def __init__(self, name=None, dtype=None, ragged=False, **kwargs):
        super().__init__(name=name, dtype=dtype, ragged=ragged, **kwargs)
        self._metric = metrics_impl.OPAMetric(name=name, ragged=ragged)


@utils.register_keras_serializable()
class DCGMetric(NDCGMetric):
    """keras/metrics.py:800-883."""

    def __init__(self, name=None, topn=None, gain_fn=None, rank_discount_fn=None, dtype=None,
                 ragged=False, **kwargs):
        super().__init__(name=name, topn=topn, gain_fn=gain_fn, rank_discount_fn=rank_discount_fn, dtype=dtype,
                         ragged=ragged, **kwargs)
        self._metric = metrics_impl.DCGMetric(name=name, topn=topn, gain_fn=self._gain_fn,
                                              rank_discount_fn=self._rank_discount_fn, ragged=ragged)
