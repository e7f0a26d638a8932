"""Torch-CPU restatement of the TF-Ranking math core (TEST INFRASTRUCTURE ONLY).

Every function cites the reference lines it follows; paths are relative to
/root/reference/tensorflow_ranking/python/.  All arithmetic is fp32 unless the
caller passes fp64 tensors (used by the gradient tests as a higher-precision
arbiter).  The structure deliberately mirrors the reference's op graph -- the
``[B, L, L]`` broadcast tensors are materialised -- because this file is also
the timed CPU baseline ("torch-CPU restatement of the TF-Ranking op graph").

Deterministic tie rule (the reference shuffles ties at random): descending by
score, equal scores keep index order, masked-out entries last in index order.
"""
from __future__ import annotations

import math
from typing import Callable, List, Optional, Sequence

import torch

_EPSILON = 1e-10          # losses_impl.py:28
_PADDING_LABEL = -1.0     # utils.py:21
_PADDING_PREDICTION = -1e6  # utils.py:22
_PADDING_WEIGHT = 0.0     # utils.py:23


def _t(x, dtype=torch.float32):
    if torch.is_tensor(x):
        return x
    return torch.as_tensor(x, dtype=dtype)


# ----------------------------------------------------------------------------
# utils.py
# ----------------------------------------------------------------------------
def is_label_valid(labels):
    """utils.py:78-81."""
    return _t(labels) >= 0.0


def _get_shuffle_indices(shape, mask=None):
    """utils.py:84-112 with shuffle_ties=False (zeros, +2.0 where masked out)."""
    shuffle_values = torch.zeros(shape, dtype=torch.float32)
    if mask is not None:
        shuffle_values = torch.where(mask, shuffle_values, shuffle_values + 2.0)
    return torch.sort(shuffle_values, dim=-1, stable=True).indices


def sort_by_scores(scores, features_list, topn=None, mask=None):
    """utils.py:115-164, deterministic tie rule.  Returns sorted features."""
    scores = _t(scores).to(torch.float32)
    assert scores.dim() == 2
    list_size = scores.shape[1]
    if topn is None:
        topn = list_size
    topn = min(topn, list_size)
    shuffle_ind = None
    if mask is not None:
        mask = _t(mask, torch.bool)
        # :150 (global min; TF reduces an empty tensor to +inf, torch refuses it)
        scores = torch.where(mask, scores, scores.min() if scores.numel() else scores.new_tensor(float('inf')))
        shuffle_ind = _get_shuffle_indices(scores.shape, mask)
        scores = torch.gather(scores, 1, shuffle_ind)
    # tf.math.top_k(sorted=True): descending, ties -> lower index first.
    indices = torch.sort(scores, dim=1, descending=True, stable=True).indices[:, :topn]
    if shuffle_ind is not None:
        indices = torch.gather(shuffle_ind, 1, indices)
    out = []
    for f in features_list:
        f = _t(f)
        if f.dim() == 2:
            out.append(torch.gather(f, 1, indices))
        else:
            idx = indices.unsqueeze(-1).expand(-1, -1, f.shape[2])
            out.append(torch.gather(f, 1, idx))
    return out


def sorted_ranks(scores):
    """utils.py:167-195: 1-based int ranks."""
    scores = _t(scores)
    b, l = scores.shape
    positions = torch.arange(l).unsqueeze(0).expand(b, l)
    sorted_positions = sort_by_scores(scores, [positions])[0]
    return (torch.sort(sorted_positions, dim=1, stable=True).indices + 1).to(torch.int32)


def ragged_to_dense(labels, predictions, weights):
    """utils.py:421-443.  Ragged inputs are python lists of lists."""
    n = len(labels)
    width = max((len(r) for r in labels), default=0)

    def pad(rows, value):
        out = torch.full((n, width), value, dtype=torch.float32)
        for i, r in enumerate(rows):
            if len(r):
                out[i, :len(r)] = torch.as_tensor(r, dtype=torch.float32)
        return out

    mask = pad([[1.0] * len(r) for r in labels], 0.0).to(torch.bool)
    dense_labels = pad(labels, _PADDING_LABEL)
    dense_pred = pad(predictions, _PADDING_PREDICTION) if predictions is not None else None
    if isinstance(weights, (list, tuple)) and len(weights) == n and all(
            isinstance(w, (list, tuple)) and len(w) == len(r) for w, r in zip(weights, labels)):
        weights = pad(weights, _PADDING_WEIGHT)   # ragged per-item weights
    elif weights is not None:
        weights = _t(weights)
    return dense_labels, dense_pred, weights, mask


# ----------------------------------------------------------------------------
# Reductions.
# ----------------------------------------------------------------------------
class Reduction:
    """tf.compat.v1.losses.Reduction / tf.keras.losses.Reduction strings."""
    NONE = 'none'
    SUM = 'weighted_sum'
    MEAN = 'weighted_mean'
    SUM_OVER_BATCH_SIZE = 'weighted_sum_over_batch_size'
    SUM_BY_NONZERO_WEIGHTS = 'weighted_sum_by_nonzero_weights'
    AUTO = 'auto'
    KERAS_SUM = 'sum'
    KERAS_SUM_OVER_BATCH_SIZE = 'sum_over_batch_size'


def _safe_div(num, den):
    return torch.where(den != 0, num / torch.where(den != 0, den, torch.ones_like(den)),
                       torch.zeros_like(num))


def compute_weighted_loss_v1(losses, weights, reduction):
    """tf.compat.v1.losses.compute_weighted_loss semantics (third-party TF op,
    call sites losses_impl.py:813,1167)."""
    losses = _t(losses)
    weights = _t(weights).to(losses.dtype)
    weighted = losses * weights
    if reduction == Reduction.NONE:
        return weighted
    total = weighted.sum()
    bw = torch.broadcast_to(weights, weighted.shape) if weights.dim() <= weighted.dim() \
        else weights
    if reduction == Reduction.SUM:
        return total
    if reduction == Reduction.MEAN:
        return _safe_div(total, bw.sum())
    if reduction == Reduction.SUM_BY_NONZERO_WEIGHTS:
        return _safe_div(total, (bw != 0).sum().to(losses.dtype))
    if reduction == Reduction.SUM_OVER_BATCH_SIZE:
        return total / weighted.numel()
    raise ValueError('Invalid reduction: %s' % reduction)


def keras_compute_weighted_loss(losses, sample_weight, reduction):
    """tf.keras losses_utils.compute_weighted_loss (call sites
    keras/losses.py:272,831-832).  sample_weight of rank losses.rank+1 with a
    trailing 1 is squeezed; rank losses.rank-1 is expanded."""
    losses = _t(losses)
    if sample_weight is None:
        sample_weight = 1.0
    w = _t(sample_weight).to(losses.dtype)
    if w.dim() == losses.dim() + 1 and w.shape[-1] == 1:
        w = w.squeeze(-1)
    elif w.dim() == losses.dim() - 1 and w.dim() > 0:
        w = w.unsqueeze(-1)
    weighted = losses * w
    if reduction in (Reduction.NONE,):
        return weighted
    if reduction in (Reduction.KERAS_SUM, Reduction.SUM):
        return weighted.sum()
    if reduction in (Reduction.AUTO, Reduction.KERAS_SUM_OVER_BATCH_SIZE,
                     Reduction.SUM_OVER_BATCH_SIZE):
        return weighted.sum() / weighted.numel()
    raise ValueError('Invalid reduction: %s' % reduction)


# ----------------------------------------------------------------------------
# Gain / discount functions (keras/utils.py:51-121, metrics_impl.py:31-33).
# ----------------------------------------------------------------------------
def identity(label):
    return label


def inverse(rank):
    rank = _t(rank)
    return _safe_div(torch.ones_like(rank), rank)


def pow_minus_1(label):
    label = _t(label)
    return torch.pow(torch.tensor(2.0, dtype=label.dtype), label) - 1.0


def log2_inverse(rank):
    rank = _t(rank)
    return _safe_div(torch.full_like(rank, math.log(2.0)), torch.log1p(rank))


def log1p_inverse(rank):
    """1/log1p(rank): losses_impl.py:111 default, losses.py:455."""
    return 1.0 / torch.log1p(_t(rank))


# ----------------------------------------------------------------------------
# losses_impl.py helpers.
# ----------------------------------------------------------------------------
def _safe_default_gain_fn(labels):
    """losses_impl.py:33-49."""
    max_labels = labels.max(dim=-1, keepdim=True).values
    two = torch.tensor(2.0, dtype=labels.dtype)
    return torch.pow(two, labels - max_labels) - torch.pow(two, -max_labels)


def _apply_pairwise_op(op, tensor):
    """losses_impl.py:61-64."""
    return op(tensor.unsqueeze(2), tensor.unsqueeze(1))


def _get_valid_pairs_and_clean_labels(labels):
    """losses_impl.py:67-74."""
    is_valid = is_label_valid(labels)
    valid_pairs = _apply_pairwise_op(torch.logical_and, is_valid)
    labels = torch.where(is_valid, labels, torch.zeros_like(labels))
    return valid_pairs, labels


def approx_ranks(logits):
    """losses_impl.py:77-106 (tile, tile, sub, sigmoid, sum)."""
    logits = _t(logits)
    list_size = logits.shape[1]
    x = logits.unsqueeze(2).repeat(1, 1, list_size)
    y = logits.unsqueeze(1).repeat(1, list_size, 1)
    pairs = torch.sigmoid(y - x)
    return pairs.sum(dim=-1) + 0.5


def inverse_max_dcg(labels, gain_fn=pow_minus_1, rank_discount_fn=log1p_inverse, topn=None):
    """losses_impl.py:109-134."""
    labels = _t(labels)
    ideal_sorted_labels, = sort_by_scores(labels, [labels], topn=topn)
    rank = torch.arange(ideal_sorted_labels.shape[1]) + 1
    discounted_gain = gain_fn(ideal_sorted_labels) * rank_discount_fn(rank.to(labels.dtype))
    discounted_gain = discounted_gain.sum(dim=1, keepdim=True)
    return torch.where(discounted_gain > 0.0, 1.0 / discounted_gain,
                       torch.zeros_like(discounted_gain))


def ndcg(labels, ranks=None, perm_mat=None):
    """losses_impl.py:137-167."""
    labels = _t(labels)
    if ranks is not None and perm_mat is not None:
        raise ValueError('Cannot use both ranks and perm_mat simultaneously.')
    if ranks is None:
        ranks = torch.arange(labels.shape[1]) + 1
    discounts = 1.0 / torch.log1p(_t(ranks).to(labels.dtype))
    gains = _safe_default_gain_fn(labels)
    if perm_mat is not None:
        gains = (perm_mat * gains.unsqueeze(1)).sum(dim=-1)
    dcg = (gains * discounts).sum(dim=-1, keepdim=True)
    return dcg * inverse_max_dcg(labels, gain_fn=_safe_default_gain_fn)


def _compute_ranks(logits, is_valid):
    """losses_impl.py:483-500."""
    scores = torch.where(is_valid, logits,
                         -1e-6 * torch.ones_like(logits) + logits.min(dim=1, keepdim=True).values)
    return sorted_ranks(scores)


def _pairwise_comparison(labels, logits, mask):
    """losses_impl.py:503-537."""
    pairwise_label_diff = _apply_pairwise_op(torch.sub, labels)
    pairwise_logits = _apply_pairwise_op(torch.sub, logits)
    pairwise_labels = (pairwise_label_diff > 0).to(logits.dtype)
    valid_pair = _apply_pairwise_op(torch.logical_and, mask)
    pairwise_labels = pairwise_labels * valid_pair.to(logits.dtype)
    return pairwise_labels, pairwise_logits


# ----------------------------------------------------------------------------
# Lambda weights (losses_impl.py:170-369).
# ----------------------------------------------------------------------------
class LabelDiffLambdaWeight:
    """losses_impl.py:210-217."""

    def pair_weights(self, labels, ranks):
        return torch.abs(_apply_pairwise_op(torch.sub, _t(labels)))

    def individual_weights(self, labels, ranks):
        return labels


class DCGLambdaWeight:
    """losses_impl.py:219-369 (AbstractDCGLambdaWeight + DCGLambdaWeight)."""

    def __init__(self, topn=None, gain_fn=identity, rank_discount_fn=inverse,
                 normalized=False, smooth_fraction=0.0):
        if not 0.0 <= smooth_fraction <= 1.0:
            raise ValueError('smooth_fraction %s should be in range [0, 1].' % smooth_fraction)
        self._topn = topn
        self._gain_fn = gain_fn
        self._rank_discount_fn = rank_discount_fn
        self._normalized = normalized
        self._smooth_fraction = smooth_fraction

    def _pair_rank_discount(self, ranks, topn):
        """losses_impl.py:334-369."""
        f32 = torch.float32
        pair_valid_rank = _apply_pairwise_op(torch.logical_or, ranks <= topn)
        rank_diff = torch.abs(_apply_pairwise_op(torch.sub, ranks)).to(f32)
        u = torch.where(
            torch.logical_and(rank_diff > 0, pair_valid_rank),
            torch.abs(self._rank_discount_fn(torch.clamp(rank_diff, min=1.0))
                      - self._rank_discount_fn(rank_diff + 1)),
            torch.zeros_like(rank_diff))
        rank_discount = torch.where(ranks > topn, torch.zeros_like(ranks.to(f32)),
                                    self._rank_discount_fn(ranks.to(f32)))
        v = torch.abs(_apply_pairwise_op(torch.sub, rank_discount))
        pair_discount = (1.0 - self._smooth_fraction) * u + self._smooth_fraction * v
        pair_mask = _apply_pairwise_op(torch.logical_or, ranks <= topn)
        return pair_discount * pair_mask.to(f32)

    def pair_weights(self, labels, ranks):
        """losses_impl.py:255-279."""
        labels = _t(labels)
        ranks = _t(ranks, torch.int32)
        valid_pair, labels = _get_valid_pairs_and_clean_labels(labels)
        gain = self._gain_fn(labels)
        if self._normalized:
            gain = gain * inverse_max_dcg(labels, gain_fn=self._gain_fn,
                                          rank_discount_fn=self._rank_discount_fn,
                                          topn=self._topn)
        pair_gain = _apply_pairwise_op(torch.sub, gain)
        pair_gain = pair_gain * valid_pair.to(torch.float32)
        list_size = labels.shape[1]
        topn = self._topn or list_size
        pair_weight = torch.abs(pair_gain) * self._pair_rank_discount(ranks, topn)
        pair_weight = pair_weight * float(list_size)
        return pair_weight

    def individual_weights(self, labels, ranks):
        """losses_impl.py:281-296."""
        labels = _t(labels)
        labels = torch.where(is_label_valid(labels), labels, torch.zeros_like(labels))
        gain = self._gain_fn(labels)
        if self._normalized:
            gain = gain * inverse_max_dcg(labels, gain_fn=self._gain_fn,
                                          rank_discount_fn=self._rank_discount_fn,
                                          topn=self._topn)
        rank_discount = self._rank_discount_fn(_t(ranks).to(torch.float32))
        return gain * rank_discount


class DCGLambdaWeightV2(DCGLambdaWeight):
    """losses_impl.py:372-394 (shares AbstractDCGLambdaWeight.pair_weights with the class above)."""

    def __init__(self, topn=None, gain_fn=identity, rank_discount_fn=inverse, normalized=False):
        super().__init__(topn, gain_fn, rank_discount_fn, normalized, 0.0)

    def _pair_rank_discount(self, ranks, topn):
        f32 = torch.float32
        rank_diff = torch.abs(_apply_pairwise_op(torch.sub, ranks)).to(f32)
        max_rank = _apply_pairwise_op(torch.maximum, ranks).to(f32)
        multiplier = torch.where(max_rank > float(topn), 1.0 / (1.0 - self._rank_discount_fn(max_rank)),
                                 torch.ones_like(max_rank))
        return torch.where(
            rank_diff > 0.0,
            torch.abs(self._rank_discount_fn(torch.clamp(rank_diff, min=1.0))
                      - self._rank_discount_fn(rank_diff + 1)) * multiplier,
            torch.zeros_like(rank_diff))


class YetiDCGLambdaWeight(DCGLambdaWeightV2):
    """losses_impl.py:397-407."""

    def pair_weights(self, labels, ranks):
        pair_weight = super().pair_weights(labels, ranks)
        ranks = _t(ranks, torch.int32)
        neighbor_pair = torch.abs(_apply_pairwise_op(torch.sub, ranks)) == 1
        return pair_weight * neighbor_pair.to(torch.float32)


class PrecisionLambdaWeight:
    """losses_impl.py:410-454."""

    def __init__(self, topn, positive_fn=lambda label: label >= 1.0):
        self._topn = topn
        self._positive_fn = positive_fn

    def pair_weights(self, labels, ranks):
        labels = _t(labels)
        ranks = _t(ranks, torch.int32)
        valid_pair, labels = _get_valid_pairs_and_clean_labels(labels)
        binary_labels = self._positive_fn(labels).to(torch.float32)
        label_diff = torch.abs(_apply_pairwise_op(torch.sub, binary_labels))
        label_diff = label_diff * valid_pair.to(torch.float32)
        rank_mask = _apply_pairwise_op(torch.logical_xor, ranks <= self._topn)
        return label_diff * rank_mask.to(torch.float32)

    def individual_weights(self, labels, ranks):
        return labels


def NDCGLambdaWeightV2(topn=None, gain_fn=None, rank_discount_fn=None):
    """keras/losses.py:151-162."""
    return DCGLambdaWeightV2(topn, gain_fn or pow_minus_1, rank_discount_fn or log2_inverse, normalized=True)


def KerasYetiDCGLambdaWeight(topn=None, gain_fn=None, rank_discount_fn=None, normalized=False):
    """keras/losses.py:172-186."""
    return YetiDCGLambdaWeight(topn, gain_fn or pow_minus_1, rank_discount_fn or log2_inverse, normalized=normalized)


def NDCGLambdaWeight(topn=None, gain_fn=None, rank_discount_fn=None, smooth_fraction=0.0):
    """keras/losses.py:197-212."""
    return DCGLambdaWeight(topn, gain_fn or pow_minus_1, rank_discount_fn or log2_inverse,
                           normalized=True, smooth_fraction=smooth_fraction)


def create_ndcg_lambda_weight(topn=None, smooth_fraction=0.0):
    """losses.py:450-457."""
    return DCGLambdaWeight(topn, gain_fn=pow_minus_1, rank_discount_fn=log1p_inverse,
                           normalized=True, smooth_fraction=smooth_fraction)


def create_reciprocal_rank_lambda_weight(topn=None, smooth_fraction=0.0):
    """losses.py:460-467."""
    return DCGLambdaWeight(topn, gain_fn=identity, rank_discount_fn=inverse, normalized=True,
                           smooth_fraction=smooth_fraction)


def create_p_list_mle_lambda_weight(list_size):
    """losses.py:470-484: rank discount 2^(list_size - rank) - 1."""
    return ListMLELambdaWeight(rank_discount_fn=lambda rank: torch.pow(torch.tensor(2.), list_size - rank) - 1.)


# ----------------------------------------------------------------------------
# Gumbel sampler (losses_impl.py:540-649).
# ----------------------------------------------------------------------------
def gumbel_noise_from_uniform(u, eps=1e-20):
    """losses_impl.py:647-649."""
    return -torch.log(-torch.log(u + eps) + eps)


class GumbelSampler:
    """losses_impl.py:540-644.  The uniform noise is injected (``uniform``
    [B, S, L]) or drawn from a torch generator; TF's Philox stream cannot be
    reproduced ("parity unpinned" for the noise values)."""

    def __init__(self, sample_size=8, temperature=1.0, seed=None):
        self._sample_size = sample_size
        self._temperature = temperature
        self._seed = seed

    def sample(self, labels, logits, weights=None, uniform=None):
        labels = _t(labels)
        logits = _t(logits)
        b, l = labels.shape
        s = self._sample_size
        expanded_labels = labels.unsqueeze(1).repeat(1, s, 1).reshape(b * s, l)
        if uniform is None:
            gen = torch.Generator().manual_seed(0 if self._seed is None else self._seed)
            uniform = torch.rand((b, s, l), generator=gen, dtype=torch.float32)
        sampled = logits.unsqueeze(1).repeat(1, s, 1) + gumbel_noise_from_uniform(uniform)
        sampled = sampled.reshape(b * s, l)
        valid = is_label_valid(expanded_labels)
        sampled = torch.where(valid, sampled / self._temperature,
                              math.log(1e-20) * torch.ones_like(sampled))
        sampled = torch.log(torch.softmax(sampled, dim=-1) + 1e-20)
        expanded_weights = weights
        if expanded_weights is not None:
            w = _t(expanded_weights)
            if w.dim() == 1:
                w = w.unsqueeze(1).unsqueeze(1)
            else:
                w = w.unsqueeze(1)
            w = w.repeat(1, s, 1)
            expanded_weights = w.reshape(b * s, -1)
        return expanded_labels, sampled, expanded_weights


# ----------------------------------------------------------------------------
# Losses (losses_impl.py:652-1603).
# ----------------------------------------------------------------------------
class _RankingLoss:
    """losses_impl.py:652-860."""

    def __init__(self, name=None, lambda_weight=None, temperature=1.0, ragged=False):
        self._name = name
        self._lambda_weight = lambda_weight
        self._temperature = temperature
        self._ragged = ragged

    def _prepare_and_validate_params(self, labels, logits, weights, mask):
        if self._ragged:
            labels, logits, weights, mask = ragged_to_dense(labels, logits, weights)
        labels = _t(labels)
        if mask is None:
            mask = is_label_valid(labels)
        if weights is None:
            weights = 1.0
        return labels, _t(logits), _t(weights), _t(mask, torch.bool)

    def compute_unreduced_loss(self, labels, logits, mask=None):
        labels, logits, _, mask = self._prepare_and_validate_params(labels, logits, None, mask)
        return self._compute_unreduced_loss_impl(labels, logits, mask)

    def normalize_weights(self, labels, weights):
        if self._ragged:
            labels, _, weights, _ = ragged_to_dense(labels, None, weights)
        return self._normalize_weights_impl(_t(labels), None if weights is None else _t(weights))

    def _normalize_weights_impl(self, labels, weights):
        return 1.0 if weights is None else weights

    def get_logits(self, logits):
        return _t(logits) / self._temperature

    def compute(self, labels, logits, weights, reduction, mask=None):
        """losses_impl.py:787-814."""
        labels = _t(labels)
        logits = self.get_logits(logits)
        if mask is not None:
            mask = _t(mask, torch.bool)
        losses, loss_weights = self._compute_unreduced_loss_impl(labels, logits, mask)
        weights = _t(self._normalize_weights_impl(labels, None if weights is None else _t(weights))) \
            * loss_weights
        return compute_weighted_loss_v1(losses, weights, reduction)


class _PairwiseLoss(_RankingLoss):
    """losses_impl.py:863-930."""

    def _compute_unreduced_loss_impl(self, labels, logits, mask=None):
        if mask is None:
            mask = is_label_valid(labels)
        ranks = _compute_ranks(logits, mask)
        pairwise_labels, pairwise_logits = _pairwise_comparison(labels, logits, mask)
        pairwise_weights = pairwise_labels
        if self._lambda_weight is not None:
            pairwise_weights = pairwise_weights * self._lambda_weight.pair_weights(labels, ranks)
        pairwise_weights = pairwise_weights.detach()
        return self._pairwise_loss(pairwise_logits), pairwise_weights

    def compute_per_list(self, labels, logits, weights, mask=None):
        labels, logits, weights, mask = self._prepare_and_validate_params(
            labels, logits, weights, mask)
        losses, loss_weights = self._compute_unreduced_loss_impl(labels, logits, mask)
        weights = self._normalize_weights_impl(labels, weights) * loss_weights
        weighted = losses * weights
        per_list_weights = weights.sum(dim=(1, 2))
        per_list_losses = weighted.sum(dim=(1, 2))
        return _safe_div(per_list_losses, per_list_weights), per_list_weights

    def _normalize_weights_impl(self, labels, weights):
        if weights is None:
            weights = 1.0
        weights = torch.where(is_label_valid(labels), torch.ones_like(labels) * weights,
                              torch.zeros_like(labels))
        return weights.unsqueeze(2)


class PairwiseLogisticLoss(_PairwiseLoss):
    """losses_impl.py:933-940."""

    def _pairwise_loss(self, pairwise_logits):
        return torch.relu(-pairwise_logits) + torch.log1p(torch.exp(-torch.abs(pairwise_logits)))


class PairwiseHingeLoss(_PairwiseLoss):
    """losses_impl.py:943-948."""

    def _pairwise_loss(self, pairwise_logits):
        return torch.relu(1 - pairwise_logits)


class PairwiseSoftZeroOneLoss(_PairwiseLoss):
    """losses_impl.py:951-958."""

    def _pairwise_loss(self, pairwise_logits):
        return torch.where(pairwise_logits > 0, 1. - torch.sigmoid(pairwise_logits),
                           torch.sigmoid(-pairwise_logits))


class PairwiseMSELoss(_PairwiseLoss):
    """losses_impl.py:961-998."""

    def _compute_unreduced_loss_impl(self, labels, logits, mask=None):
        if mask is None:
            mask = is_label_valid(labels)
        pairwise_label_diff = _apply_pairwise_op(torch.sub, labels)
        pairwise_logit_diff = _apply_pairwise_op(torch.sub, logits)
        pairwise_mse_loss = torch.square(pairwise_logit_diff - pairwise_label_diff)
        valid_pair = _apply_pairwise_op(torch.logical_and, mask)
        pairwise_weights = torch.ones_like(pairwise_mse_loss)
        pairwise_weights = pairwise_weights - torch.eye(labels.shape[1], dtype=pairwise_weights.dtype).unsqueeze(0)
        pairwise_weights = pairwise_weights * valid_pair.to(torch.float32)
        if self._lambda_weight is not None:
            ranks = _compute_ranks(logits, mask)
            pairwise_weights = pairwise_weights * self._lambda_weight.pair_weights(labels, ranks)
        return pairwise_mse_loss, pairwise_weights.detach()


class _ListwiseLoss(_RankingLoss):
    """losses_impl.py:1001-1033."""

    def _normalize_weights_impl(self, labels, weights):
        if weights is None:
            return 1.0
        weights = _t(weights)
        is_valid = is_label_valid(labels)
        labels = torch.where(is_valid, labels, torch.zeros_like(labels))
        return _safe_div((weights * labels).sum(dim=1, keepdim=True),
                         labels.sum(dim=1, keepdim=True))

    def compute_per_list(self, labels, logits, weights, mask=None):
        # NB: no temperature here (losses_impl.py:1017-1033); weights=None has
        # already become 1.0 in _prepare_and_validate_params (:689-690).
        labels, logits, weights, mask = self._prepare_and_validate_params(
            labels, logits, weights, mask)
        losses, loss_weights = self._compute_unreduced_loss_impl(labels, logits, mask)
        weights = _t(self._normalize_weights_impl(labels, weights)) * loss_weights
        return losses.squeeze(1), weights.squeeze(1)


class SoftmaxLoss(_ListwiseLoss):
    """losses_impl.py:1119-1197."""

    def precompute(self, labels, logits, weights, mask=None):
        if mask is None:
            mask = is_label_valid(labels)
        ranks = _compute_ranks(logits, mask)
        labels = torch.where(mask, labels, torch.zeros_like(labels))
        logits = torch.where(mask, logits, math.log(_EPSILON) * torch.ones_like(logits))
        if self._lambda_weight is not None and isinstance(self._lambda_weight, DCGLambdaWeight):
            labels = self._lambda_weight.individual_weights(labels, ranks)
        if weights is not None:
            labels = labels * weights
        return labels, logits

    def _compute_unreduced_loss_impl(self, labels, logits, mask=None):
        if mask is None:
            mask = is_label_valid(labels)
        label_sum = labels.sum(dim=1, keepdim=True)
        nonzero_mask = label_sum.reshape(-1) > 0.0
        padded_labels = torch.where(nonzero_mask.unsqueeze(1), labels,
                                    _EPSILON * torch.ones_like(labels))
        padded_labels = torch.where(mask, padded_labels, torch.zeros_like(padded_labels))
        padded_label_sum = padded_labels.sum(dim=1, keepdim=True)
        labels_for_softmax = _safe_div(padded_labels, padded_label_sum)
        weights_for_softmax = label_sum.reshape(-1)
        losses = -(labels_for_softmax * torch.log_softmax(logits, dim=1)).sum(dim=1)
        return losses, weights_for_softmax

    def compute(self, labels, logits, weights, reduction, mask=None):
        labels, logits, weights, mask = self._prepare_and_validate_params(
            labels, logits, weights, mask)
        logits = self.get_logits(logits)
        labels, logits = self.precompute(labels, logits, weights, mask)
        losses, weights = self._compute_unreduced_loss_impl(labels, logits, mask)
        return compute_weighted_loss_v1(losses, weights, reduction)

    def compute_per_list(self, labels, logits, weights, mask=None):
        labels, logits, weights, mask = self._prepare_and_validate_params(
            labels, logits, weights, mask)
        logits = self.get_logits(logits)
        labels, logits = self.precompute(labels, logits, weights, mask)
        return self._compute_unreduced_loss_impl(labels, logits, mask)

    def compute_unreduced_loss(self, labels, logits, mask=None):
        labels, logits, _, mask = self._prepare_and_validate_params(labels, logits, None, mask)
        logits = self.get_logits(logits)
        labels, logits = self.precompute(labels, logits, weights=None, mask=mask)
        return self._compute_unreduced_loss_impl(labels, logits, mask)


class ApproxNDCGLoss(_ListwiseLoss):
    """losses_impl.py:1579-1603."""

    def __init__(self, name=None, lambda_weight=None, temperature=0.1, ragged=False):
        super().__init__(name, lambda_weight, temperature, ragged)

    def _compute_unreduced_loss_impl(self, labels, logits, mask=None):
        if mask is None:
            mask = is_label_valid(labels)
        labels = torch.where(mask, labels, torch.zeros_like(labels))
        logits = torch.where(mask, logits,
                             -1e3 * torch.ones_like(logits)
                             + logits.min(dim=-1, keepdim=True).values)
        label_sum = labels.sum(dim=1, keepdim=True)
        nonzero_mask = label_sum.reshape(-1) > 0.0
        labels = torch.where(nonzero_mask.unsqueeze(1), labels, _EPSILON * torch.ones_like(labels))
        ranks = approx_ranks(logits)
        return -ndcg(labels, ranks), nonzero_mask.to(logits.dtype).reshape(-1, 1)


class CircleLoss(_ListwiseLoss):
    """losses_impl.py:1036-1116."""

    def __init__(self, lambda_weight=None, gamma=64, margin=0.25, ragged=False):
        super().__init__(lambda_weight=lambda_weight, temperature=1.0, ragged=ragged)
        self._margin = margin
        self._gamma = gamma

    def get_logits(self, logits):
        return torch.clamp(_t(logits), 0., 1.)

    def _compute_unreduced_loss_impl(self, labels, logits, mask=None):
        if mask is None:
            mask = is_label_valid(labels)
        score_i = logits.unsqueeze(2)
        score_j = logits.unsqueeze(1)
        alpha_i = torch.relu(1 - score_i + self._margin).detach()
        alpha_j = torch.relu(score_j + self._margin).detach()
        pairwise_logits = alpha_i * (1 - score_i - self._margin) + alpha_j * (score_j - self._margin)
        pairwise_labels, _ = _pairwise_comparison(labels, logits, mask)
        pairwise_weights = pairwise_labels.detach()
        losses = torch.exp(self._gamma * pairwise_logits)
        per_list_losses = torch.log1p((losses * pairwise_weights).sum(dim=(1, 2)))
        per_list_weights = pairwise_weights.sum(dim=(1, 2)) / (pairwise_weights > 0).to(logits.dtype).sum(dim=(1, 2))
        return per_list_losses.unsqueeze(1), per_list_weights.unsqueeze(1)


def neural_sort(logits, mask=None):
    """losses_impl.py:1716-1801, op for op."""
    logits = _t(logits)
    if mask is None:
        mask = torch.ones_like(logits, dtype=torch.bool)
    mask = torch.as_tensor(mask, dtype=torch.bool)
    logits = torch.where(mask, logits, torch.zeros_like(logits))
    num_valid_entries = mask.to(torch.int32).sum(dim=1, keepdim=True)
    logit_diff = torch.abs(logits.unsqueeze(2) - logits.unsqueeze(1))
    valid_pair_mask = _apply_pairwise_op(torch.logical_and, mask)
    logit_diff = torch.where(valid_pair_mask, logit_diff, torch.zeros_like(logit_diff))
    logit_diff_sum = logit_diff.sum(dim=1, keepdim=True)
    masked_range = torch.cumsum(mask.to(torch.int32), dim=1)
    scaling = (num_valid_entries + 1 - 2 * masked_range).to(logits.dtype)
    scaling = scaling.unsqueeze(2)
    scaled_logits = scaling * logits.unsqueeze(1)
    p_logits = scaled_logits - logit_diff_sum
    p_logits = torch.where(valid_pair_mask, p_logits, torch.full_like(p_logits, -math.inf))
    p_logits = torch.where(_apply_pairwise_op(torch.logical_or, mask), p_logits, torch.zeros_like(p_logits))
    sorted_mask_indices = torch.argsort(mask.to(torch.int32), dim=1, descending=True, stable=True)
    p_logits = torch.gather(p_logits, 1, sorted_mask_indices.unsqueeze(2).expand_as(p_logits))
    return torch.softmax(p_logits, dim=-1)


class NeuralSortCrossEntropyLoss(_ListwiseLoss):
    """losses_impl.py:1635-1673."""

    def _compute_unreduced_loss_impl(self, labels, logits, mask=None):
        if mask is None:
            mask = is_label_valid(labels)
        labels = torch.where(mask, labels, torch.zeros_like(labels))
        logits = torch.where(mask, logits, torch.zeros_like(logits))
        label_sum = labels.sum(dim=1, keepdim=True)
        nonzero_mask = label_sum.reshape(-1) > 0.0
        true_perm = neural_sort(labels, mask=mask)
        smooth_perm = neural_sort(logits, mask=mask)
        # tf.nn.softmax_cross_entropy_with_logits_v2(labels, logits, axis=2) = -sum labels * log_softmax(logits)
        losses = -(true_perm * torch.log_softmax(torch.log(1e-20 + smooth_perm), dim=2)).sum(dim=2)
        sorted_mask = torch.sort(mask.to(logits.dtype), dim=1, descending=True).values.to(torch.bool)
        losses = torch.where(sorted_mask, losses, torch.zeros_like(losses))
        losses = _safe_div(losses.sum(dim=-1, keepdim=True), mask.to(logits.dtype).sum(dim=-1, keepdim=True))
        return losses, nonzero_mask.to(logits.dtype).reshape(-1, 1)


class NeuralSortNDCGLoss(_ListwiseLoss):
    """losses_impl.py:1676-1713."""

    def _compute_unreduced_loss_impl(self, labels, logits, mask=None):
        if mask is None:
            mask = is_label_valid(labels)
        labels = torch.where(mask, labels, torch.zeros_like(labels))
        logits = torch.where(mask, logits, torch.zeros_like(logits))
        label_sum = labels.sum(dim=1, keepdim=True)
        nonzero_mask = label_sum.reshape(-1) > 0.0
        labels = torch.where(nonzero_mask.unsqueeze(1), labels, _EPSILON * torch.ones_like(labels))
        smooth_perm = neural_sort(logits, mask=mask)
        return -ndcg(labels, perm_mat=smooth_perm), nonzero_mask.to(logits.dtype).reshape(-1, 1)


class UniqueSoftmaxLoss(_ListwiseLoss):
    """losses_impl.py:1250-1281 (the [B, L, L+1] denominator tensor, op for op)."""

    def _compute_unreduced_loss_impl(self, labels, logits, mask=None):
        if mask is None:
            mask = is_label_valid(labels)
        labels = torch.where(mask, labels, torch.zeros_like(labels))
        logits = torch.where(mask, logits, math.log(_EPSILON) * torch.ones_like(logits))
        pairwise_labels, _ = _pairwise_comparison(labels, logits, mask)
        denominator_logits = logits.unsqueeze(1) * pairwise_labels
        denominator_logits = torch.cat([denominator_logits, logits.unsqueeze(2)], dim=2)
        denominator_mask = torch.cat([pairwise_labels, torch.ones_like(logits).unsqueeze(2)], dim=2)
        denominator_logits = torch.where(denominator_mask > 0.0, denominator_logits,
                                         -1e-3 + denominator_logits.min() * torch.ones_like(denominator_logits))
        logits_max = denominator_logits.max(dim=-1, keepdim=True).values
        denominator_logits = denominator_logits - logits_max
        logits = logits - logits_max.squeeze(-1)
        gains = torch.pow(torch.tensor(2.0, dtype=labels.dtype), labels) - 1
        per_doc_softmax = -logits + torch.log((torch.exp(denominator_logits) * denominator_mask).sum(dim=-1))
        losses = (per_doc_softmax * gains).sum(dim=1, keepdim=True)
        return losses, torch.ones_like(losses)


class ListMLELambdaWeight:
    """losses_impl.py:457-480."""

    def __init__(self, rank_discount_fn):
        self._rank_discount_fn = rank_discount_fn

    def pair_weights(self, labels, ranks):
        pass

    def individual_weights(self, labels, ranks):
        return torch.ones_like(_t(labels)) * self._rank_discount_fn(_t(ranks).to(torch.float32))


class ListMLELoss(_ListwiseLoss):
    """losses_impl.py:1541-1576 with the deterministic tie rule (the reference shuffles ties,
    shuffle_ties=True, seed=37: parity unpinned on tied labels)."""

    def _compute_unreduced_loss_impl(self, labels, logits, mask=None):
        if mask is None:
            mask = is_label_valid(labels)
        labels = torch.where(mask, labels, torch.zeros_like(labels))
        logits = torch.where(mask, logits, math.log(_EPSILON) * torch.ones_like(logits))
        scores = torch.where(mask, labels, labels.min(dim=1, keepdim=True).values - 1e-6 * torch.ones_like(labels))
        sorted_labels, sorted_logits = sort_by_scores(scores, [labels, logits])
        raw_max = sorted_logits.max(dim=1, keepdim=True).values
        sorted_logits = sorted_logits - raw_max
        sums = torch.flip(torch.cumsum(torch.flip(torch.exp(sorted_logits), dims=[1]), dim=1), dims=[1])
        sums = torch.log(sums) - sorted_logits
        if self._lambda_weight is not None and isinstance(self._lambda_weight, ListMLELambdaWeight):
            b, l = sorted_labels.shape
            sums = sums * self._lambda_weight.individual_weights(
                sorted_labels, (torch.arange(l) + 1).unsqueeze(0).expand(b, l))
        nll = sums.sum(dim=1, keepdim=True)
        return nll, torch.ones_like(nll)


class ApproxMRRLoss(_ListwiseLoss):
    """losses_impl.py:1606-1632."""

    def __init__(self, name=None, lambda_weight=None, temperature=0.1, ragged=False):
        super().__init__(name, lambda_weight, temperature, ragged)

    def _compute_unreduced_loss_impl(self, labels, logits, mask=None):
        if mask is None:
            mask = is_label_valid(labels)
        labels = torch.where(mask, labels, torch.zeros_like(labels))
        logits = torch.where(mask, logits,
                             -1e3 * torch.ones_like(logits)
                             + logits.min(dim=-1, keepdim=True).values)
        label_sum = labels.sum(dim=1, keepdim=True)
        nonzero_mask = label_sum.reshape(-1) > 0.0
        labels = torch.where(nonzero_mask.unsqueeze(1), labels, _EPSILON * torch.ones_like(labels))
        rr = 1. / approx_ranks(logits)
        rr = (rr * labels).sum(dim=-1, keepdim=True)
        mrr = rr / labels.sum(dim=-1, keepdim=True)
        return -mrr, nonzero_mask.to(logits.dtype).reshape(-1, 1)


class PolyOneSoftmaxLoss(SoftmaxLoss):
    """losses_impl.py:1200-1247."""

    def __init__(self, name=None, lambda_weight=None, epsilon=1.0, temperature=1.0, ragged=False):
        super().__init__(name, lambda_weight=lambda_weight, temperature=temperature, ragged=ragged)
        self._epsilon = epsilon

    def _compute_unreduced_loss_impl(self, labels, logits, mask=None):
        if mask is None:
            mask = is_label_valid(labels)
        label_sum = labels.sum(dim=1, keepdim=True)
        nonzero_mask = label_sum.reshape(-1) > 0.0
        padded_labels = torch.where(nonzero_mask.unsqueeze(1), labels, _EPSILON * torch.ones_like(labels))
        padded_labels = torch.where(mask, padded_labels, torch.zeros_like(padded_labels))
        padded_label_sum = padded_labels.sum(dim=1, keepdim=True)
        labels_for_softmax = _safe_div(padded_labels, padded_label_sum)
        weights_for_softmax = label_sum.reshape(-1)
        pt = (labels_for_softmax * torch.softmax(logits, dim=-1)).sum(dim=-1)
        ce = -(labels_for_softmax * torch.log_softmax(logits, dim=1)).sum(dim=1)
        return ce + self._epsilon * (1 - pt), weights_for_softmax


class _PointwiseLoss(_RankingLoss):
    """losses_impl.py:1284-1321."""

    def _normalize_weights_impl(self, labels, weights):
        if weights is None:
            weights = 1.0
        return torch.where(is_label_valid(labels), torch.ones_like(labels) * weights,
                           torch.zeros_like(labels))

    def compute_per_list(self, labels, logits, weights, mask=None):
        labels, logits, weights, mask = self._prepare_and_validate_params(
            labels, logits, weights, mask)
        losses, loss_weights = self._compute_unreduced_loss_impl(labels, logits, mask)
        weights = self._normalize_weights_impl(labels, weights) * loss_weights
        per_list_weights = weights.sum(dim=1)
        per_list_losses = (losses * weights).sum(dim=1)
        return _safe_div(per_list_losses, per_list_weights), per_list_weights


class SigmoidCrossEntropyLoss(_PointwiseLoss):
    """losses_impl.py:1425-1446 (config 1, CPU reference path only)."""

    def __init__(self, name=None, temperature=1.0, ragged=False):
        super().__init__(name, None, temperature, ragged)

    def _compute_unreduced_loss_impl(self, labels, logits, mask=None):
        if mask is None:
            mask = is_label_valid(labels)
        labels = torch.where(mask, labels, torch.zeros_like(labels))
        logits = torch.where(mask, logits, torch.zeros_like(logits))
        losses = torch.relu(logits) - logits * labels + torch.log1p(torch.exp(-torch.abs(logits)))
        return losses, mask.to(logits.dtype)


class MeanSquaredLoss(_PointwiseLoss):
    """losses_impl.py:1449-1469."""

    def __init__(self, name=None, ragged=False):
        super().__init__(name, None, 1.0, ragged)

    def _compute_unreduced_loss_impl(self, labels, logits, mask=None):
        if mask is None:
            mask = is_label_valid(labels)
        labels = torch.where(mask, labels, torch.zeros_like(labels))
        logits = torch.where(mask, logits, torch.zeros_like(logits))
        return torch.square(labels - logits), mask.to(logits.dtype)


# ----------------------------------------------------------------------------
# Keras-level wrappers (keras/losses.py:247-335, 824-832, 1332-1341).
# ----------------------------------------------------------------------------
# ----------------------------------------------------------------------------
# losses.make_loss_fn (losses.py:57-311): the estimator-era factory over the classes above.
# ----------------------------------------------------------------------------
def make_loss_fn(loss_keys, loss_weights=None, weights_feature_name=None, lambda_weight=None,
                 reduction=Reduction.SUM_BY_NONZERO_WEIGHTS, name=None, params=None, gumbel_params=None, uniform=None):
    """losses.py:258-311 / _LossFunctionMaker (:57-255).  ``'a:0.1,b:0.9'`` keys carry their weights (:100-108); the
    per-example weights come from ``features[weights_feature_name]`` reshaped to 2-D (:204-209); the Gumbel keys see the
    sampled (labels, logits, weights) and the caller's lambda_weight, pointwise / approx / neural-sort keys never see a
    lambda_weight (:125-160, :225-236); the result is the weighted sum of the per-key reduced losses (:241-252).
    ``uniform`` injects the sampler's noise (TF's stream is unpinned)."""
    if isinstance(loss_keys, str) and (':' in loss_keys or ',' in loss_keys):
        if loss_weights is not None:
            raise ValueError('`loss_weights` has to be None when weights are encoded in `loss_keys`.')
        parsed = {}
        for part in loss_keys.split(','):                       # utils.py:382-418 parse_keys_and_weights
            if ':' in part:
                k, w = part.split(':')
                parsed[k.strip()] = float(w.strip())
            else:
                parsed[part.strip()] = 1.0
        loss_keys, loss_weights = list(parsed.keys()), list(parsed.values())
    if reduction not in (Reduction.SUM, Reduction.MEAN, Reduction.SUM_BY_NONZERO_WEIGHTS, Reduction.SUM_OVER_BATCH_SIZE):
        raise ValueError('Invalid reduction: %s' % reduction)
    if not loss_keys:
        raise ValueError('loss_keys cannot be None or empty.')
    if not isinstance(loss_keys, list):
        loss_keys = [loss_keys]
    if loss_weights and len(loss_keys) != len(loss_weights):
        raise ValueError('loss_keys and loss_weights must have the same size.')
    params = dict(params or {})
    sampler = GumbelSampler(**(gumbel_params or {}))
    with_lambda = {'pairwise_hinge_loss': PairwiseHingeLoss, 'pairwise_logistic_loss': PairwiseLogisticLoss,
                   'pairwise_soft_zero_one_loss': PairwiseSoftZeroOneLoss, 'pairwise_mse_loss': PairwiseMSELoss,
                   'circle_loss': CircleLoss, 'softmax_loss': SoftmaxLoss, 'poly_one_softmax_loss': PolyOneSoftmaxLoss,
                   'unique_softmax_loss': UniqueSoftmaxLoss, 'list_mle_loss': ListMLELoss}
    plain = {'sigmoid_cross_entropy_loss': SigmoidCrossEntropyLoss, 'mean_squared_loss': MeanSquaredLoss,
             'approx_ndcg_loss': ApproxNDCGLoss, 'approx_mrr_loss': ApproxMRRLoss,
             'neural_sort_cross_entropy_loss': NeuralSortCrossEntropyLoss, 'neural_sort_ndcg_loss': NeuralSortNDCGLoss}
    gumbel = {'yeti_logistic_loss': PairwiseLogisticLoss, 'gumbel_approx_ndcg_loss': ApproxNDCGLoss,
              'gumbel_neural_sort_cross_entropy_loss': NeuralSortCrossEntropyLoss,
              'gumbel_neural_sort_ndcg_loss': NeuralSortNDCGLoss}

    def _loss_fn(labels, logits, features):
        weights = None
        if weights_feature_name:
            weights = _t(features[weights_feature_name])
            weights = weights.reshape(weights.shape[0], -1) if weights.dim() != 1 else weights.reshape(-1, 1)
        terms = []
        for key in loss_keys:
            if key in with_lambda:
                loss = with_lambda[key](lambda_weight=lambda_weight, **params)
                terms.append(loss.compute(labels, logits, weights, reduction))
            elif key in plain:
                terms.append(plain[key](**params).compute(labels, logits, weights, reduction))
            elif key in gumbel:
                g_labels, g_logits, g_weights = sampler.sample(labels, logits, weights=weights, uniform=uniform)
                kw = dict(params)
                if lambda_weight is not None and key == 'yeti_logistic_loss':
                    kw['lambda_weight'] = lambda_weight
                terms.append(gumbel[key](**kw).compute(g_labels, g_logits, g_weights, reduction))
            else:
                raise ValueError('Invalid loss_key: %s.' % key)
        if loss_weights:
            terms = [t * w for t, w in zip(terms, loss_weights)]
        total = terms[0]
        for t in terms[1:]:
            total = total + t
        return total
    return _loss_fn


def keras_loss_call(loss, y_true, y_pred, sample_weight=None, reduction=Reduction.AUTO,
                    gumbel_sampler: Optional[GumbelSampler] = None, uniform=None):
    """Restates ``tfr.keras.losses.<Loss>.__call__`` for an L1 ``loss`` object."""
    if gumbel_sampler is not None:       # keras/losses.py:1332-1341
        y_true, y_pred, sample_weight = gumbel_sampler.sample(
            y_true, y_pred, weights=sample_weight, uniform=uniform)
    if isinstance(loss, SoftmaxLoss):    # keras/losses.py:824-832
        losses, sw = loss.compute_per_list(y_true, y_pred, sample_weight)
        return keras_compute_weighted_loss(losses, sw, reduction)
    sw = loss.normalize_weights(y_true, sample_weight)          # :270
    logits = y_pred
    if not loss._ragged:
        logits = loss.get_logits(y_pred)                         # :277
        losses, weights = loss.compute_unreduced_loss(labels=y_true, logits=logits)
    else:
        # ragged: get_logits on the dense view (temperature is a scalar divide).
        dl, dp, _, _ = ragged_to_dense(y_true, y_pred, None)
        saved, loss._ragged = loss._ragged, False
        try:
            losses, weights = loss.compute_unreduced_loss(labels=dl, logits=loss.get_logits(dp))
        finally:
            loss._ragged = saved
    out = losses * weights
    if isinstance(loss, _PairwiseLoss):  # keras/losses.py:324-335
        out = out.sum(dim=2)
    return keras_compute_weighted_loss(out, sw, reduction)


def keras_calibrated_softmax_call(y_true, y_pred, sample_weight=None, reduction=Reduction.AUTO, lambda_weight=None,
                                  temperature=1.0, virtual_label=0.0):
    """Restates ``tfr.keras.losses.CalibratedSoftmaxLoss.__call__`` (keras/losses.py:899-936): a virtual item with
    label ``virtual_label`` and score 0 (weight 1 under per-item weights) is appended, then SoftmaxLoss.__call__."""
    y_true, y_pred = _t(y_true), _t(y_pred)
    b = y_true.shape[0]
    y_true = torch.cat([y_true, torch.ones((b, 1), dtype=y_true.dtype) * virtual_label], dim=1)       # :915-918
    y_pred = torch.cat([y_pred, torch.zeros((b, 1), dtype=y_pred.dtype)], dim=1)                      # :921-922
    if sample_weight is not None:
        sample_weight = _t(sample_weight)
        if sample_weight.dim() == 2 and sample_weight.shape[1] > 1:                                   # :924-928
            sample_weight = torch.cat([sample_weight, torch.ones((b, 1), dtype=sample_weight.dtype)], dim=1)
    loss = SoftmaxLoss(lambda_weight=lambda_weight, temperature=temperature)
    return keras_loss_call(loss, y_true, y_pred, sample_weight, reduction)


# ----------------------------------------------------------------------------
# metrics_impl.py
# ----------------------------------------------------------------------------
def tree_sum(x):
    """Fixed-order fp32 row sum shared with the HIP metric kernel so that
    NDCG@k is bit-reproducible: zero-pad the row to P = next power of two, then
    repeatedly fold the upper half onto the lower half (t[i] += t[i + h]).
    (The reference's tf.reduce_sum order is TF/Eigen-internal and unknowable
    here; any order is within 1e-6 of the reference literals.)"""
    x = _t(x)
    n = x.shape[1]
    p = 1
    while p < n:
        p *= 2
    if p != n:
        x = torch.cat([x, torch.zeros(x.shape[0], p - n, dtype=x.dtype)], dim=1)
    h = p // 2
    while h >= 1:
        x = x[:, :h] + x[:, h:2 * h]
        h //= 2
    return x  # [B, 1]


def _per_example_weights_to_per_list_weights(weights, relevance, row_sum=None):
    """metrics_impl.py:63-119."""
    rs = row_sum or (lambda t: t.sum(dim=1, keepdim=True))
    nonzero_weights = rs(weights) > 0.0
    per_list_relevance = rs(relevance)
    nonzero_relevance = torch.where(nonzero_weights, (per_list_relevance > 0.0).to(torch.float32),
                                    torch.zeros_like(per_list_relevance))
    nonzero_relevance_count = nonzero_relevance.sum(dim=0, keepdim=True)
    per_list_weights = _safe_div(rs(weights * relevance), per_list_relevance)
    sum_weights = per_list_weights.sum(dim=0, keepdim=True)
    avg_weight = torch.where(nonzero_relevance_count > 0.0,
                             _safe_div(sum_weights, nonzero_relevance_count),
                             torch.ones_like(nonzero_relevance_count))
    return torch.where(nonzero_weights,
                       torch.where(per_list_relevance > 0.0, per_list_weights,
                                   torch.ones_like(per_list_weights) * avg_weight),
                       torch.zeros_like(per_list_weights))


def _discounted_cumulative_gain(labels, weights, gain_fn=pow_minus_1,
                                rank_discount_fn=log2_inverse, row_sum=None, full_list_size=None):
    """metrics_impl.py:122-151.  The discount is evaluated for the positions of the
    FULL list and sliced to the (top-n truncated) length: mathematically identical to
    the reference, and independent of torch's vectorised-vs-scalar log1p paths so
    that NDCG@k stays bit-reproducible against the kernel's shared table."""
    rs = row_sum or (lambda t: t.sum(dim=1, keepdim=True))
    list_size = labels.shape[1]
    position = torch.arange(1, (full_list_size or list_size) + 1, dtype=torch.float32)
    gain = gain_fn(labels.to(torch.float32))
    discount = rank_discount_fn(position)[:list_size]
    return rs(weights * gain * discount)


def _row_min(x):
    """tf.reduce_min(x, axis=1, keepdims=True); an empty axis reduces to +inf in TF (metrics_impl_test.py:1498-1506
    feeds lists without items), torch refuses it."""
    if x.shape[1] == 0:
        return torch.full((x.shape[0], 1), float('inf'), dtype=x.dtype)
    return x.min(dim=1, keepdim=True).values


class _RankingMetric:
    """metrics_impl.py:210-310."""

    def __init__(self, ragged=False):
        self._ragged = ragged

    def _prepare_and_validate_params(self, labels, predictions, weights, mask):
        labels = _t(labels)
        predictions = _t(predictions)
        weights = 1.0 if weights is None else _t(weights)
        example_weights = torch.ones_like(labels) * weights
        if mask is None:
            mask = is_label_valid(labels)
        mask = torch.logical_and(_t(mask, torch.bool), example_weights > 0.0)
        labels = torch.where(mask, labels, torch.zeros_like(labels))
        predictions = torch.where(
            mask, predictions,
            -1e-6 * torch.ones_like(predictions) + _row_min(predictions))
        return labels, predictions, example_weights, mask

    def compute(self, labels, predictions, weights=None, mask=None):
        if self._ragged:
            labels, predictions, weights, mask = ragged_to_dense(labels, predictions, weights)
        labels, predictions, weights, mask = self._prepare_and_validate_params(
            labels, predictions, weights, mask)
        return self._compute_impl(labels, predictions, weights, mask)


class MRRMetric(_RankingMetric):
    """metrics_impl.py:429-459."""

    def __init__(self, name=None, topn=None, ragged=False):
        super().__init__(ragged)
        self._topn = topn

    def _compute_impl(self, labels, predictions, weights, mask):
        topn = predictions.shape[1] if self._topn is None else self._topn
        sorted_labels, = sort_by_scores(predictions, [labels], topn=topn, mask=mask)
        n = sorted_labels.shape[1]
        relevance = (sorted_labels >= 1.0).to(torch.float32)
        reciprocal_rank = 1.0 / torch.arange(1, n + 1, dtype=torch.float32)
        mrr = (relevance * reciprocal_rank).max(dim=1, keepdim=True).values
        per_list_weights = _per_example_weights_to_per_list_weights(
            weights=weights, relevance=(labels >= 1.0).to(torch.float32), row_sum=tree_sum)
        return mrr, per_list_weights


class NDCGMetric(_RankingMetric):
    """metrics_impl.py:631-670.  Row sums use ``tree_sum`` (see its docstring)."""

    def __init__(self, name=None, topn=None, gain_fn=pow_minus_1, rank_discount_fn=log2_inverse,
                 ragged=False):
        super().__init__(ragged)
        self._topn = topn
        self._gain_fn = gain_fn
        self._rank_discount_fn = rank_discount_fn

    def _compute_impl(self, labels, predictions, weights, mask):
        topn = predictions.shape[1] if self._topn is None else self._topn
        sorted_labels, sorted_weights = sort_by_scores(
            predictions, [labels, weights], topn=topn, mask=mask)
        dcg = _discounted_cumulative_gain(sorted_labels, sorted_weights, self._gain_fn,
                                          self._rank_discount_fn, row_sum=tree_sum,
                                          full_list_size=predictions.shape[1])
        weighted_gains = weights * self._gain_fn(labels.to(torch.float32))
        ideal_sorted_labels, ideal_sorted_weights = sort_by_scores(
            weighted_gains, [labels, weights], topn=topn, mask=mask)
        ideal_dcg = _discounted_cumulative_gain(ideal_sorted_labels, ideal_sorted_weights,
                                                self._gain_fn, self._rank_discount_fn,
                                                row_sum=tree_sum, full_list_size=predictions.shape[1])
        per_list_ndcg = _safe_div(dcg, ideal_dcg)
        per_list_weights = _per_example_weights_to_per_list_weights(
            weights=weights, relevance=self._gain_fn(labels.to(torch.float32)), row_sum=tree_sum)
        return per_list_ndcg, per_list_weights


class HitsMetric(_RankingMetric):
    """metrics_impl.py:462-506."""

    def __init__(self, name=None, topn=None, ragged=False):
        super().__init__(ragged)
        self._topn = topn

    def _compute_impl(self, labels, predictions, weights, mask):
        topn = predictions.shape[1] if self._topn is None else self._topn
        sorted_labels, = sort_by_scores(predictions, [labels], topn=topn, mask=mask)
        relevance = (sorted_labels >= 1.0).to(torch.float32)
        hits = relevance.max(dim=1, keepdim=True).values
        per_list_weights = _per_example_weights_to_per_list_weights(
            weights=weights, relevance=(labels >= 1.0).to(torch.float32), row_sum=tree_sum)
        return hits, per_list_weights


class ARPMetric(_RankingMetric):
    """metrics_impl.py:509-536 (row sums in the sorted order, tree_sum)."""

    def __init__(self, name=None, ragged=False):
        super().__init__(ragged)

    def _compute_impl(self, labels, predictions, weights, mask):
        topn = predictions.shape[1]
        sorted_labels, sorted_weights = sort_by_scores(predictions, [labels, weights], topn=topn, mask=mask)
        weighted_labels = sorted_labels * sorted_weights
        position = torch.arange(1, topn + 1, dtype=torch.float32) * torch.ones_like(weighted_labels)
        per_list_weights = tree_sum(weighted_labels)
        per_list_arp = _safe_div(tree_sum(position * weighted_labels), per_list_weights)
        return per_list_arp, per_list_weights


def _per_list_recall(labels, predictions, topn, mask):
    """metrics_impl.py:154-177."""
    sorted_labels = sort_by_scores(predictions, [labels], topn=topn, mask=mask)[0]
    topn_positives = (sorted_labels >= 1.0).to(torch.float32)
    rel = (labels >= 1.0).to(torch.float32)
    return _safe_div(topn_positives.sum(dim=1, keepdim=True), rel.sum(dim=1, keepdim=True))


def _per_list_precision(labels, predictions, topn, mask):
    """metrics_impl.py:180-207."""
    sorted_labels = sort_by_scores(predictions, [labels], topn=topn, mask=mask)[0]
    relevance = (sorted_labels >= 1.0).to(torch.float32)
    if topn is None:
        topn = relevance.shape[1]
    valid_topn = torch.clamp(mask.to(torch.int32).sum(dim=1, keepdim=True), max=topn)
    return _safe_div(relevance.sum(dim=1, keepdim=True), valid_topn.to(torch.float32))


class RecallMetric(_RankingMetric):
    """metrics_impl.py:539-561."""

    def __init__(self, name=None, topn=None, ragged=False):
        super().__init__(ragged)
        self._topn = topn

    def _compute_impl(self, labels, predictions, weights, mask):
        topn = predictions.shape[1] if self._topn is None else self._topn
        out = _per_list_recall(labels, predictions, topn, mask)
        w = _per_example_weights_to_per_list_weights(weights, (labels >= 1.0).to(torch.float32), row_sum=tree_sum)
        return out, w


class PrecisionMetric(_RankingMetric):
    """metrics_impl.py:564-586."""

    def __init__(self, name=None, topn=None, ragged=False):
        super().__init__(ragged)
        self._topn = topn

    def _compute_impl(self, labels, predictions, weights, mask):
        topn = predictions.shape[1] if self._topn is None else self._topn
        out = _per_list_precision(labels, predictions, topn, mask)
        w = _per_example_weights_to_per_list_weights(weights, (labels >= 1.0).to(torch.float32), row_sum=tree_sum)
        return out, w


class MeanAveragePrecisionMetric(_RankingMetric):
    """metrics_impl.py:589-628 (float row sums: tree_sum)."""

    def __init__(self, name=None, topn=None, ragged=False):
        super().__init__(ragged)
        self._topn = topn

    def _compute_impl(self, labels, predictions, weights, mask):
        topn = predictions.shape[1] if self._topn is None else self._topn
        relevance = (labels >= 1.0).to(torch.float32)
        sorted_relevance, sorted_weights = sort_by_scores(predictions, [relevance, weights], topn=topn, mask=mask)
        counts = torch.cumsum(sorted_relevance, dim=1)
        cutoffs = torch.cumsum(torch.ones_like(sorted_relevance), dim=1)
        precisions = _safe_div(counts, cutoffs)
        total_precision = tree_sum(precisions * sorted_weights * sorted_relevance)
        total_relevance = tree_sum(weights * relevance)
        per_list_map = _safe_div(total_precision, total_relevance)
        w = _per_example_weights_to_per_list_weights(weights, relevance, row_sum=tree_sum)
        return per_list_map, w


class OPAMetric(_RankingMetric):
    """metrics_impl.py:708-743."""

    def __init__(self, name=None, ragged=False):
        super().__init__(ragged)

    def _compute_impl(self, labels, predictions, weights, mask):
        valid_pair = torch.logical_and(mask.unsqueeze(2), mask.unsqueeze(1))
        pair_label_diff = labels.unsqueeze(2) - labels.unsqueeze(1)
        pair_pred_diff = predictions.unsqueeze(2) - predictions.unsqueeze(1)
        correct_pairs = (pair_label_diff > 0).to(torch.float32) * (pair_pred_diff > 0).to(torch.float32)
        pair_weights = (pair_label_diff > 0).to(torch.float32) * weights.unsqueeze(2) * valid_pair.to(torch.float32)
        per_list_weights = pair_weights.sum(dim=(1, 2)).unsqueeze(1)
        per_list_opa = _safe_div((correct_pairs * pair_weights).sum(dim=(1, 2)).unsqueeze(1), per_list_weights)
        return per_list_opa, per_list_weights


class BPrefMetric(_RankingMetric):
    """metrics_impl.py:825-898."""

    def __init__(self, name=None, topn=None, use_trec_version=True, ragged=False):
        super().__init__(ragged)
        self._topn = topn
        self._use_trec_version = use_trec_version

    def _compute_impl(self, labels, predictions, weights, mask):
        topn = predictions.shape[1] if self._topn is None else self._topn
        relevance = (labels >= 1.0).to(torch.float32)
        irrelevance = mask.to(torch.float32) - relevance
        total_relevance = relevance.sum(dim=1, keepdim=True)
        total_irrelevance = irrelevance.sum(dim=1, keepdim=True)
        sorted_relevance, sorted_irrelevance = sort_by_scores(predictions, [relevance, irrelevance], mask=mask, topn=topn)
        numerator = torch.minimum(torch.cumsum(sorted_irrelevance, dim=1), total_relevance)
        denominator = torch.minimum(total_irrelevance, total_relevance) if self._use_trec_version else total_relevance
        bpref = _safe_div(((1. - _safe_div(numerator, denominator)) * sorted_relevance).sum(dim=1, keepdim=True),
                          total_relevance)
        per_list_weights = _per_example_weights_to_per_list_weights(
            weights=weights, relevance=(relevance >= 1.0).to(torch.float32))
        return bpref, per_list_weights


class PWAMetric(_RankingMetric):
    """metrics_impl.py:901-965."""

    def __init__(self, name=None, topn=5, ragged=False):
        super().__init__(ragged)
        self._topn = topn

    def compute(self, labels, predictions, weights=None, mask=None):
        if weights is not None and not self._ragged:
            w = torch.as_tensor(weights)
            if w.dim() != 2 or w.shape[1] != 1:
                raise ValueError('Weights should be a `Tensor` of the shape[batch_size, 1]')
        return super().compute(labels, predictions, weights, mask)

    def _compute_impl(self, labels, predictions, weights, mask):
        topn = predictions.shape[1] if self._topn is None else self._topn
        sorted_labels, sorted_mask = sort_by_scores(predictions, [labels, mask], topn=topn, mask=mask)
        sorted_list_size = sorted_labels.shape[1]
        position_weights = 1.0 / torch.arange(1, sorted_list_size + 1).to(torch.float32)
        masked_position_weights = sorted_mask.to(torch.float32) * position_weights
        pwa = _safe_div((sorted_labels * masked_position_weights).sum(dim=1, keepdim=True),
                        masked_position_weights.sum(dim=1, keepdim=True))
        per_list_weights = weights.mean(dim=1, keepdim=True)
        return pwa, per_list_weights


class DCGMetric(_RankingMetric):
    """metrics_impl.py:673-705."""

    def __init__(self, name=None, topn=None, gain_fn=pow_minus_1, rank_discount_fn=log2_inverse, ragged=False):
        super().__init__(ragged)
        self._topn = topn
        self._gain_fn = gain_fn
        self._rank_discount_fn = rank_discount_fn

    def _compute_impl(self, labels, predictions, weights, mask):
        topn = predictions.shape[1] if self._topn is None else self._topn
        sorted_labels, sorted_weights = sort_by_scores(predictions, [labels, weights], topn=topn, mask=mask)
        dcg = _discounted_cumulative_gain(sorted_labels, sorted_weights, self._gain_fn, self._rank_discount_fn,
                                          row_sum=tree_sum, full_list_size=predictions.shape[1])
        per_list_weights = _per_example_weights_to_per_list_weights(
            weights=weights, relevance=self._gain_fn(labels.to(torch.float32)), row_sum=tree_sum)
        return _safe_div(dcg, per_list_weights), per_list_weights


def _alpha_dcg_gain_fn(labels, alpha):
    """metrics_impl.py:36-60."""
    cum_subtopics = torch.cumsum(labels, dim=1) - labels          # tf.cumsum(exclusive=True)
    return (labels * torch.pow(torch.tensor(1 - alpha, dtype=labels.dtype), cum_subtopics)).sum(dim=-1)


class _DivRankingMetric(_RankingMetric):
    """metrics_impl.py:313-426."""

    def __init__(self, name=None, topn=None, ragged=False):
        super().__init__(ragged)
        self._topn = topn

    def compute(self, labels, predictions, weights=None, mask=None):
        if self._ragged:
            n_sub = max((len(r[0]) for r in labels if len(r)), default=1)
            _, predictions, weights, mask = ragged_to_dense([[0.] * len(r) for r in predictions], predictions, weights)
            dense = torch.full((len(labels), predictions.shape[1], n_sub), -1.0)
            for i, r in enumerate(labels):
                if len(r):
                    dense[i, :len(r)] = torch.as_tensor(r, dtype=torch.float32)
            labels = dense
        labels, predictions, weights, mask = self._prepare_and_validate_params(labels, predictions, weights, mask)
        return self._compute_impl(labels, predictions, weights, mask)

    def _prepare_and_validate_params(self, labels, predictions, weights, mask):
        labels = _t(labels)
        predictions = _t(predictions)
        assert labels.dim() == 3
        if mask is None:
            mask = is_label_valid(labels)
        mask = _t(mask, torch.bool)
        if mask.dim() == 3:
            mask = mask.any(dim=2)
        predictions = torch.where(mask, predictions,
                                  -1e-6 * torch.ones_like(predictions) + _row_min(predictions))
        labels = torch.where(mask.unsqueeze(2), labels, torch.zeros_like(labels))
        weights = torch.tensor(1.0) if weights is None else _t(weights)
        example_weights = torch.ones_like(predictions) * weights
        return labels, predictions, example_weights, mask

    def _compute_per_list_weights(self, weights, labels):
        return _per_example_weights_to_per_list_weights(weights, (labels >= 1.0).any(dim=-1).to(torch.float32),
                                                        row_sum=tree_sum)

    def _compute_impl(self, labels, predictions, weights, mask):
        topn = predictions.shape[1] if self._topn is None else self._topn
        per_list_metric = self._compute_per_list_metric(labels, predictions, weights, topn, mask)
        return per_list_metric, self._compute_per_list_weights(weights, labels)


class PrecisionIAMetric(_DivRankingMetric):
    """metrics_impl.py:746-782."""

    def _compute_per_list_metric(self, labels, predictions, weights, topn, mask):
        sorted_labels = sort_by_scores(predictions, [labels], topn=topn, mask=mask)[0]
        relevance = (sorted_labels >= 1.0).to(torch.float32).sum(dim=-1)
        num_subtopics = (labels >= 1.0).any(dim=1, keepdim=True).to(torch.float32).sum(dim=-1)
        valid_topn = torch.clamp(mask.to(torch.int32).sum(dim=1, keepdim=True), max=topn)
        return _safe_div(relevance.sum(dim=1, keepdim=True),
                         (valid_topn.to(torch.float32) * num_subtopics).sum(dim=1, keepdim=True))


class AlphaDCGMetric(_DivRankingMetric):
    """metrics_impl.py:785-822."""

    def __init__(self, name=None, topn=None, alpha=0.5, rank_discount_fn=log2_inverse, seed=None, ragged=False):
        super().__init__(name, topn, ragged)
        self._alpha = alpha
        self._rank_discount_fn = rank_discount_fn

    def _compute_per_list_metric(self, labels, predictions, weights, topn, mask):
        sorted_labels, sorted_weights = sort_by_scores(predictions, [labels, weights], topn=topn, mask=mask)
        alpha_dcg = _discounted_cumulative_gain(sorted_labels, sorted_weights,
                                                lambda l: _alpha_dcg_gain_fn(l, self._alpha), self._rank_discount_fn,
                                                row_sum=tree_sum, full_list_size=predictions.shape[1])
        return _safe_div(alpha_dcg, self._compute_per_list_weights(weights, labels))


def keras_metric_mean(metric, batches):
    """keras/metrics.py:156-193: running weighted mean over update_state calls.
    ``batches`` = iterable of (y_true, y_pred, sample_weight)."""
    total = torch.zeros((), dtype=torch.float32)
    count = torch.zeros((), dtype=torch.float32)
    for y_true, y_pred, sw in batches:
        val, w = metric.compute(y_true, y_pred, sw)
        total = total + (val * w).sum()
        count = count + w.sum()
    return _safe_div(total, count)


# ----------------------------------------------------------------------------
# Scorer pieces (keras/layers.py:26-77,126-175,231-265; model.py:164-244,341-421).
# ----------------------------------------------------------------------------
def padded_nd_indices(is_valid):
    """utils.py:308-356 with shuffle=False: per row, valid indices first (in
    order), then padding slots filled circularly with the valid indices.
    Returns the [B, L] index of the source item for each slot."""
    is_valid = _t(is_valid, torch.bool)
    b, l = is_valid.shape
    # organize_valid_indices: valid first, stable.
    keys = torch.where(is_valid, torch.zeros(b, l), torch.ones(b, l))
    order = torch.sort(keys, dim=1, stable=True).indices
    n_valid = is_valid.sum(dim=1, keepdim=True)
    pos = torch.arange(l).unsqueeze(0).expand(b, l)
    circ = torch.where(n_valid > 0, pos % torch.clamp(n_valid, min=1), torch.zeros_like(pos))
    return torch.gather(order, 1, circ)


def flatten_list(context, example, mask):
    """keras/layers.py:126-175 (circular padding, tile context, flatten)."""
    mask = _t(mask, torch.bool)
    b, l = mask.shape
    idx = padded_nd_indices(mask)
    flat_ctx = None
    if context is not None:
        flat_ctx = context.unsqueeze(1).repeat(1, l, 1).reshape(b * l, -1)
    g = torch.gather(example, 1, idx.unsqueeze(-1).expand(-1, -1, example.shape[2]))
    return flat_ctx, g.reshape(b * l, -1)


def restore_list(flattened_logits, mask):
    """keras/layers.py:231-265: reshape to [B, L]; invalid := log(1e-10)."""
    mask = _t(mask, torch.bool)
    logits = flattened_logits.reshape(mask.shape)
    return torch.where(mask, logits, math.log(_EPSILON) * torch.ones_like(logits))


def dnn_tower(x, weights: Sequence[torch.Tensor], biases: Sequence[torch.Tensor],
              activation=torch.relu):
    """keras/layers.py:26-77 with use_batch_norm=False, dropout=0 (inference /
    deterministic path): Dense->act for hidden layers, final Dense(output_units)."""
    for i, (w, bias) in enumerate(zip(weights, biases)):
        x = x @ w + bias
        if i < len(weights) - 1:
            x = activation(x)
    return x


def create_tower_train(x, weights: Sequence[torch.Tensor], biases: Sequence[torch.Tensor],
                       gammas: Sequence[torch.Tensor], betas: Sequence[torch.Tensor], activation=torch.relu,
                       epsilon=1e-3, keep_masks: Optional[Sequence[torch.Tensor]] = None, rate=0.0):
    """keras/layers.py:26-77 in TRAINING mode, as separate ops like the reference graph: per hidden layer
    Dense -> BatchNormalization (batch mean, biased batch variance, tf.keras epsilon 1e-3) -> activation
    (-> Dropout with the given keep masks, scaled by 1/(1-rate)), then the output Dense.  ``weights[i]`` is
    ``[in, out]``; the last weight / bias pair is the output layer."""
    n_h = len(weights) - 1
    for i in range(n_h):
        z = x @ weights[i] + biases[i]
        mean = z.mean(dim=0, keepdim=True)
        var = ((z - mean) ** 2).mean(dim=0, keepdim=True)
        z = (z - mean) * torch.rsqrt(var + epsilon) * gammas[i] + betas[i]
        x = activation(z) if activation is not None else z
        if keep_masks is not None and rate > 0.0:
            x = x * keep_masks[i] / (1.0 - rate)
    return x @ weights[-1] + biases[-1]


def rolling_window_indices(size, rw_size, num_valid_entries):
    """model.py:164-202."""
    num_valid_entries = torch.as_tensor(num_valid_entries).reshape(-1, 1, 1)
    rw = (torch.arange(rw_size).unsqueeze(0) + torch.arange(size).unsqueeze(1)).unsqueeze(0)
    batch = num_valid_entries.shape[0]
    rw = rw.repeat(batch, 1, 1)
    return torch.remainder(rw, torch.clamp(num_valid_entries, min=1)) * (num_valid_entries > 0)


def form_group_indices(is_valid, group_size):
    """model.py:205-244 with shuffle=False: ([B, G=L, group_size] item indices, [B, G] mask)."""
    is_valid = _t(is_valid, torch.bool)
    b, l = is_valid.shape
    n_valid = is_valid.sum(dim=1)
    rw = rolling_window_indices(l, group_size, n_valid)                       # [B, L, gs]
    rw_raw = torch.arange(group_size).unsqueeze(0) + torch.arange(l).unsqueeze(1)
    mask = rw_raw.min(dim=1).values.unsqueeze(0) < n_valid.reshape(-1, 1)     # model.py:190-192
    keys = torch.where(is_valid, torch.zeros(b, l), torch.ones(b, l))
    organized = torch.sort(keys, dim=1, stable=True).indices                  # utils.py:203-230
    idx = torch.gather(organized.unsqueeze(1).expand(b, l, l), 2, rw)
    return idx, mask


def groupwise_logits(score_fn, example_features, is_valid, group_size, indices=None):
    """model.py:341-421 (single example feature tensor [B, L, F]): gather groups -> score [B*G, gs] -> masked
    scatter-add -> divide by counts.  ``indices`` = (idx [B, G, gs], mask [B, G]) of a shuffled / multi-shuffle run
    (model.py:313-339); default: the no-shuffle indices."""
    b, l, f = example_features.shape
    idx, mask = form_group_indices(is_valid, group_size) if indices is None else indices
    idx = idx.to(torch.int64)
    g = idx.shape[1]
    gathered = torch.gather(example_features.unsqueeze(1).expand(b, g, l, f), 2,
                            idx.unsqueeze(-1).expand(b, g, group_size, f))
    scores = score_fn(gathered.reshape(b * g, group_size, f)).reshape(b, g, group_size)
    scores_mask = mask.unsqueeze(2).expand(b, g, group_size)
    counts = torch.zeros(b, l).scatter_add_(1, idx.reshape(b, -1), scores_mask.reshape(b, -1).float())
    scores = torch.where(scores_mask, scores, torch.zeros_like(scores))
    logits = torch.zeros(b, l).scatter_add(1, idx.reshape(b, -1), scores.reshape(b, -1))
    return _safe_div(logits, counts)


def form_group_indices_with_keys(is_valid, group_size, keys):
    """model.py:205-244 with shuffle=True for GIVEN uniform draws ``keys`` [B, L] (utils.py:219-227: valid entries
    keep their draw, invalid entries get -1e-6, stable descending argsort), then the rolling windows."""
    is_valid = _t(is_valid, torch.bool)
    b, l = is_valid.shape
    n_valid = is_valid.sum(dim=1)
    rw = rolling_window_indices(l, group_size, n_valid)
    rw_raw = torch.arange(group_size).unsqueeze(0) + torch.arange(l).unsqueeze(1)
    mask = rw_raw.min(dim=1).values.unsqueeze(0) < n_valid.reshape(-1, 1)
    rand = torch.where(is_valid, _t(keys), torch.full((b, l), -1e-6))
    organized = torch.sort(rand, dim=1, descending=True, stable=True).indices
    idx = torch.gather(organized.unsqueeze(1).expand(b, l, l), 2, rw)
    return idx, mask
