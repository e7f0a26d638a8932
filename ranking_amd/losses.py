"""Mirror of ``tensorflow_ranking/python/losses.py`` (estimator-era factory).

``make_loss_fn(...)`` returns ``fn(labels, logits, features) -> scalar`` exactly
like the reference (losses.py:265-311).  One difference, on purpose: the
reference draws Gumbel samples for EVERY loss key, used or not (losses.py:216-217);
here the sampler only runs for the Gumbel keys.
"""
from __future__ import annotations

from typing import Any, Callable, Dict, List, Mapping, Optional, Sequence, Union

import torch

from . import losses_impl
from . import utils
from .losses_impl import Reduction


class RankingLossKey(object):
    """losses.py:29-56."""
    PAIRWISE_HINGE_LOSS = 'pairwise_hinge_loss'
    PAIRWISE_LOGISTIC_LOSS = 'pairwise_logistic_loss'
    PAIRWISE_SOFT_ZERO_ONE_LOSS = 'pairwise_soft_zero_one_loss'
    PAIRWISE_MSE_LOSS = 'pairwise_mse_loss'
    YETI_LOGISTIC_LOSS = 'yeti_logistic_loss'
    CIRCLE_LOSS = 'circle_loss'
    SOFTMAX_LOSS = 'softmax_loss'
    POLY_ONE_SOFTMAX_LOSS = 'poly_one_softmax_loss'
    UNIQUE_SOFTMAX_LOSS = 'unique_softmax_loss'
    SIGMOID_CROSS_ENTROPY_LOSS = 'sigmoid_cross_entropy_loss'
    MEAN_SQUARED_LOSS = 'mean_squared_loss'
    LIST_MLE_LOSS = 'list_mle_loss'
    APPROX_NDCG_LOSS = 'approx_ndcg_loss'
    APPROX_MRR_LOSS = 'approx_mrr_loss'
    GUMBEL_APPROX_NDCG_LOSS = 'gumbel_approx_ndcg_loss'
    NEURAL_SORT_CROSS_ENTROPY_LOSS = 'neural_sort_cross_entropy_loss'
    GUMBEL_NEURAL_SORT_CROSS_ENTROPY_LOSS = 'gumbel_neural_sort_cross_entropy_loss'
    NEURAL_SORT_NDCG_LOSS = 'neural_sort_ndcg_loss'
    GUMBEL_NEURAL_SORT_NDCG_LOSS = 'gumbel_neural_sort_ndcg_loss'

    @classmethod
    def all_keys(cls) -> List[str]:
        return [v for k, v in vars(cls).items() if k.isupper()]


_SUPPORTED = {
    RankingLossKey.PAIRWISE_LOGISTIC_LOSS: (losses_impl.PairwiseLogisticLoss, True, False),
    RankingLossKey.PAIRWISE_HINGE_LOSS: (losses_impl.PairwiseHingeLoss, True, False),
    RankingLossKey.PAIRWISE_SOFT_ZERO_ONE_LOSS: (losses_impl.PairwiseSoftZeroOneLoss, True, False),
    RankingLossKey.SOFTMAX_LOSS: (losses_impl.SoftmaxLoss, True, False),
    RankingLossKey.POLY_ONE_SOFTMAX_LOSS: (losses_impl.PolyOneSoftmaxLoss, True, False),
    RankingLossKey.SIGMOID_CROSS_ENTROPY_LOSS: (losses_impl.SigmoidCrossEntropyLoss, False, False),
    RankingLossKey.MEAN_SQUARED_LOSS: (losses_impl.MeanSquaredLoss, False, False),
    RankingLossKey.APPROX_NDCG_LOSS: (losses_impl.ApproxNDCGLoss, False, False),
    RankingLossKey.APPROX_MRR_LOSS: (losses_impl.ApproxMRRLoss, False, False),
    RankingLossKey.LIST_MLE_LOSS: (losses_impl.ListMLELoss, True, False),
    RankingLossKey.UNIQUE_SOFTMAX_LOSS: (losses_impl.UniqueSoftmaxLoss, True, False),
    RankingLossKey.PAIRWISE_MSE_LOSS: (losses_impl.PairwiseMSELoss, True, False),
    RankingLossKey.YETI_LOGISTIC_LOSS: (losses_impl.PairwiseLogisticLoss, False, True),
    RankingLossKey.CIRCLE_LOSS: (losses_impl.CircleLoss, True, False),
    RankingLossKey.GUMBEL_APPROX_NDCG_LOSS: (losses_impl.ApproxNDCGLoss, False, True),
    RankingLossKey.NEURAL_SORT_CROSS_ENTROPY_LOSS: (losses_impl.NeuralSortCrossEntropyLoss, False, False),
    RankingLossKey.GUMBEL_NEURAL_SORT_CROSS_ENTROPY_LOSS: (losses_impl.NeuralSortCrossEntropyLoss, False, True),
    RankingLossKey.NEURAL_SORT_NDCG_LOSS: (losses_impl.NeuralSortNDCGLoss, False, False),
    RankingLossKey.GUMBEL_NEURAL_SORT_NDCG_LOSS: (losses_impl.NeuralSortNDCGLoss, False, True),
}


def make_loss_fn(loss_keys: Union[str, Sequence[str]],
                 loss_weights: Optional[Sequence[Union[float, int]]] = None,
                 weights_feature_name: Optional[str] = None,
                 lambda_weight=None,
                 reduction: str = Reduction.SUM_BY_NONZERO_WEIGHTS,
                 name: Optional[str] = None,
                 params: Optional[Mapping[str, Any]] = None,
                 gumbel_params: Optional[Mapping[str, Any]] = None) -> Callable:
    """losses.py:265-311 / _LossFunctionMaker.make :163-260."""
    if isinstance(loss_keys, str) and ':' in loss_keys or (isinstance(loss_keys, str) and ',' in loss_keys):
        if loss_weights is not None:
            raise ValueError('`loss_weights` has to be None when weights are encoded in `loss_keys`.')
        kw = utils.parse_keys_and_weights(loss_keys)
        loss_keys, loss_weights = list(kw.keys()), list(kw.values())
    if reduction not in Reduction.all() or reduction == Reduction.NONE:
        raise ValueError('Invalid reduction: {}'.format(reduction))
    if not loss_keys:
        raise ValueError('loss_keys cannot be None or empty.')
    if not isinstance(loss_keys, list):
        loss_keys = [loss_keys] if isinstance(loss_keys, str) else list(loss_keys)
    if loss_weights and len(loss_keys) != len(loss_weights):
        raise ValueError('loss_keys and loss_weights must have the same size.')
    params = dict(params or {})
    gumbel_sampler = losses_impl.GumbelSampler(**dict(gumbel_params or {}))

    def _loss_fn(labels, logits, features: Dict[str, Any]):
        weights = None
        if weights_feature_name:
            weights = utils.reshape_to_2d(torch.as_tensor(features[weights_feature_name]))
        loss_ops = []
        for key in loss_keys:
            if key not in _SUPPORTED:
                raise ValueError('Invalid loss_key: {}.'.format(key))
            cls, takes_lambda, gumbel = _SUPPORTED[key]
            kwargs = dict(params)
            if takes_lambda or (gumbel and lambda_weight is not None):
                kwargs['lambda_weight'] = lambda_weight
            loss = cls(name, **kwargs)
            l_, s_, w_ = labels, logits, weights
            if gumbel:
                l_, s_, w_ = gumbel_sampler.sample(labels, logits, weights=weights)
            loss_ops.append(loss.compute(l_, s_, w_, reduction))
        if loss_weights:
            loss_ops = [op * w for op, w in zip(loss_ops, loss_weights)]
        out = loss_ops[0]
        for op in loss_ops[1:]:
            out = out + op
        return out

    return _loss_fn


def create_ndcg_lambda_weight(topn=None, smooth_fraction=0.):
    """losses.py:450-457."""
    return losses_impl.DCGLambdaWeight(topn, gain_fn=losses_impl._pow2_minus_1,
                                       rank_discount_fn=losses_impl._inverse_log1p, normalized=True,
                                       smooth_fraction=smooth_fraction)
