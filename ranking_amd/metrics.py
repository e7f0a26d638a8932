"""Mirror of ``tensorflow_ranking/python/metrics.py`` (estimator-era factory)
for NDCG and MRR: ``make_ranking_metric_fn`` (metrics.py:124-300), ``compute_mean``
(:79-121)."""
from __future__ import annotations

from typing import Any, Callable, Dict, Optional

import torch

from . import metrics_impl
from . import utils

_DEFAULT_GAIN_FN = metrics_impl._DEFAULT_GAIN_FN
_DEFAULT_RANK_DISCOUNT_FN = metrics_impl._DEFAULT_RANK_DISCOUNT_FN


class RankingMetricKey(object):
    """metrics.py:37-76."""
    ARP = 'arp'
    MRR = 'mrr'
    NDCG = 'ndcg'
    DCG = 'dcg'
    PRECISION = 'precision'
    RECALL = 'recall'
    MAP = 'map'
    PRECISION_IA = 'precision_ia'
    ORDERED_PAIR_ACCURACY = 'ordered_pair_accuracy'
    ALPHA_DCG = 'alpha_dcg'
    BPREF = 'bpref'
    HITS = 'hits'
    PWA = 'pwa'


def _weighted_mean(values, weights):
    den = weights.sum()
    return torch.where(den != 0, (values * weights).sum() / torch.where(den != 0, den, torch.ones_like(den)),
                       torch.zeros_like(den))


def normalized_discounted_cumulative_gain(labels, predictions, weights=None, topn=None, name=None,
                                          gain_fn=_DEFAULT_GAIN_FN,
                                          rank_discount_fn=_DEFAULT_RANK_DISCOUNT_FN):
    """metrics.py:467-505: weighted mean NDCG over the batch."""
    metric = metrics_impl.NDCGMetric(name, topn, gain_fn, rank_discount_fn)
    v, w = metric.compute(labels, predictions, weights)
    return _weighted_mean(v, w)


def mean_reciprocal_rank(labels, predictions, weights=None, topn=None, name=None):
    """metrics.py:303-330."""
    metric = metrics_impl.MRRMetric(name, topn)
    v, w = metric.compute(labels, predictions, weights)
    return _weighted_mean(v, w)


def compute_mean(metric_key, labels, predictions, weights=None, topn=None, name=None):
    """metrics.py:79-121."""
    fns = {RankingMetricKey.MRR: metrics_impl.MRRMetric(name, topn),
           RankingMetricKey.NDCG: metrics_impl.NDCGMetric(name, topn),
           RankingMetricKey.DCG: metrics_impl.DCGMetric(name, topn),
           RankingMetricKey.ARP: metrics_impl.ARPMetric(name),
           RankingMetricKey.PRECISION: metrics_impl.PrecisionMetric(name, topn),
           RankingMetricKey.RECALL: metrics_impl.RecallMetric(name, topn),
           RankingMetricKey.MAP: metrics_impl.MeanAveragePrecisionMetric(name, topn),
           RankingMetricKey.HITS: metrics_impl.HitsMetric(name, topn),
           RankingMetricKey.ORDERED_PAIR_ACCURACY: metrics_impl.OPAMetric(name),
           RankingMetricKey.BPREF: metrics_impl.BPrefMetric(name, topn),
           RankingMetricKey.PWA: metrics_impl.PWAMetric(metric_key, topn),
           RankingMetricKey.PRECISION_IA: metrics_impl.PrecisionIAMetric(name, topn),
           RankingMetricKey.ALPHA_DCG: metrics_impl.AlphaDCGMetric(name, topn)}
    if metric_key not in fns:
        raise ValueError('Invalid metric_key: {}'.format(metric_key))
    v, w = fns[metric_key].compute(labels, predictions, weights)
    return _weighted_mean(v, w)


def make_ranking_metric_fn(metric_key, weights_feature_name=None, topn=None, name=None,
                           gain_fn=_DEFAULT_GAIN_FN, rank_discount_fn=_DEFAULT_RANK_DISCOUNT_FN,
                           **kwargs) -> Callable:
    """metrics.py:124-300: returns fn(labels, predictions, features) -> mean metric."""
    def _get_weights(features):
        if weights_feature_name is None:
            return None
        return utils.reshape_to_2d(torch.as_tensor(features[weights_feature_name]))

    def _mrr(labels, predictions, features):
        return mean_reciprocal_rank(labels, predictions, _get_weights(features), topn, name)

    def _ndcg(labels, predictions, features):
        return normalized_discounted_cumulative_gain(labels, predictions, _get_weights(features), topn,
                                                     name, gain_fn, rank_discount_fn)

    def _generic(metric):
        def fn(labels, predictions, features):
            v, w = metric.compute(labels, predictions, _get_weights(features))
            return _weighted_mean(v, w)
        return fn

    fns = {RankingMetricKey.MRR: _mrr, RankingMetricKey.NDCG: _ndcg,
           RankingMetricKey.DCG: _generic(metrics_impl.DCGMetric(name, topn, gain_fn, rank_discount_fn)),
           RankingMetricKey.ARP: _generic(metrics_impl.ARPMetric(name)),
           RankingMetricKey.PRECISION: _generic(metrics_impl.PrecisionMetric(name, topn)),
           RankingMetricKey.RECALL: _generic(metrics_impl.RecallMetric(name, topn)),
           RankingMetricKey.MAP: _generic(metrics_impl.MeanAveragePrecisionMetric(name, topn)),
           RankingMetricKey.HITS: _generic(metrics_impl.HitsMetric(name, topn)),
           RankingMetricKey.ORDERED_PAIR_ACCURACY: _generic(metrics_impl.OPAMetric(name)),
           RankingMetricKey.BPREF: _generic(metrics_impl.BPrefMetric(name, topn,
                                                                     kwargs.get('use_trec_version', True))),
           RankingMetricKey.PWA: _generic(metrics_impl.PWAMetric(name, topn)),
           RankingMetricKey.PRECISION_IA: _generic(metrics_impl.PrecisionIAMetric(name, topn)),
           RankingMetricKey.ALPHA_DCG: _generic(metrics_impl.AlphaDCGMetric(
               name, topn, alpha=kwargs.get('alpha', 0.5), rank_discount_fn=rank_discount_fn,
               seed=kwargs.get('seed')))}
    if metric_key not in fns:
        raise ValueError('Invalid metric_key: {}'.format(metric_key))
    return fns[metric_key]
