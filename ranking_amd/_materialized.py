"""API-parity implementations of the reference functions whose *contract* is a
materialised tensor (``[B, L, L]`` pair weights / pair losses, approx ranks).

These are NOT on the hot path: every reduced loss / metric entry point runs a
fused gfx950 kernel instead.  They exist so that code written against the L1
protocol of the reference (losses_impl.py) keeps working, and they double as an
independent on-device cross-check in the GPU tests.  Plain torch ops on whatever
device the inputs live on; deterministic tie rule (lower index first).
"""
from __future__ import annotations

import math

import torch

_EPSILON = 1e-10


def _is_valid(labels):
    return labels >= 0.


def _pairwise(op, t):
    return op(t.unsqueeze(2), t.unsqueeze(1))


def _safe_div(num, den):
    ok = den != 0
    return torch.where(ok, num / torch.where(ok, den, torch.ones_like(den)), torch.zeros_like(num))


def _sort_desc_indices(scores, mask=None):
    """Descending, ties by index, masked-out entries last (utils.py:115-164)."""
    if mask is not None:
        key_valid = (~mask).to(torch.int8)
        first = torch.sort(key_valid, dim=1, stable=True).indices
        s = torch.gather(torch.where(mask, scores, scores.min()), 1, first)
        idx = torch.sort(s, dim=1, descending=True, stable=True).indices
        return torch.gather(first, 1, idx)
    return torch.sort(scores, dim=1, descending=True, stable=True).indices


def sorted_ranks(scores, mask=None):
    order = _sort_desc_indices(scores, mask)
    return (torch.sort(order, dim=1, stable=True).indices + 1).to(torch.int32)


def compute_ranks(logits, is_valid):
    """losses_impl.py:483-500 with 'invalid strictly last'."""
    return sorted_ranks(logits, is_valid)


def approx_ranks(logits):
    """losses_impl.py:77-106."""
    pairs = torch.sigmoid(logits.unsqueeze(1) - logits.unsqueeze(2))
    return pairs.sum(dim=-1) + .5


def inverse_max_dcg(labels, gain_fn, rank_discount_fn, topn=None):
    """losses_impl.py:109-134."""
    l = labels.shape[1]
    topn = l if topn is None else min(topn, l)
    ideal = torch.sort(labels, dim=1, descending=True, stable=True).values[:, :topn]
    rank = torch.arange(1, topn + 1, dtype=labels.dtype, device=labels.device)
    dg = (gain_fn(ideal) * rank_discount_fn(rank)).sum(dim=1, keepdim=True)
    return torch.where(dg > 0., 1. / dg, torch.zeros_like(dg))


def _safe_default_gain_fn(labels):
    """losses_impl.py:33-49."""
    max_labels = labels.max(dim=-1, keepdim=True).values
    two = torch.tensor(2.0, dtype=labels.dtype, device=labels.device)
    return torch.pow(two, labels - max_labels) - torch.pow(two, -max_labels)


def ndcg(labels, ranks=None, perm_mat=None):
    """losses_impl.py:137-167."""
    if ranks is None:
        ranks = torch.arange(1, labels.shape[1] + 1, device=labels.device)
    discounts = 1. / torch.log1p(ranks.to(torch.float32))
    gains = _safe_default_gain_fn(labels.to(torch.float32))
    if perm_mat is not None:
        gains = (perm_mat * gains.unsqueeze(1)).sum(dim=-1)
    dcg = (gains * discounts).sum(dim=-1, keepdim=True)
    return dcg * inverse_max_dcg(labels, _safe_default_gain_fn, lambda r: 1. / torch.log1p(r))


def label_diff_pair_weights(labels):
    """losses_impl.py:213-217."""
    return torch.abs(_pairwise(torch.sub, labels))


def dcg_pair_rank_discount(lw, ranks, topn):
    """losses_impl.py:334-369."""
    f32 = torch.float32
    in_top = _pairwise(torch.logical_or, ranks <= topn)
    rank_diff = torch.abs(_pairwise(torch.sub, ranks)).to(f32)
    fn = lw._rank_discount_fn
    u = torch.where(torch.logical_and(rank_diff > 0, in_top),
                    torch.abs(fn(torch.clamp(rank_diff, min=1.)) - fn(rank_diff + 1)),
                    torch.zeros_like(rank_diff))
    rd = torch.where(ranks > topn, torch.zeros_like(ranks, dtype=f32), fn(ranks.to(f32)))
    v = torch.abs(_pairwise(torch.sub, rd))
    pd = (1. - lw._smooth_fraction) * u + lw._smooth_fraction * v
    return pd * in_top.to(f32)


def dcg_v2_pair_rank_discount(lw, ranks, topn):
    """losses_impl.py:380-394."""
    f32 = torch.float32
    fn = lw._rank_discount_fn
    rank_diff = torch.abs(_pairwise(torch.sub, ranks)).to(f32)
    max_rank = _pairwise(torch.maximum, ranks).to(f32)
    mult = torch.where(max_rank > float(topn), 1. / (1. - fn(max_rank)), torch.ones_like(max_rank))
    return torch.where(rank_diff > 0., torch.abs(fn(torch.clamp(rank_diff, min=1.)) - fn(rank_diff + 1)) * mult,
                       torch.zeros_like(rank_diff))


def precision_pair_weights(lw, labels, ranks):
    """losses_impl.py:426-454."""
    is_valid = _is_valid(labels)
    valid_pair = _pairwise(torch.logical_and, is_valid)
    labels = torch.where(is_valid, labels, torch.zeros_like(labels))
    binary = lw._positive_fn(labels).to(torch.float32)
    diff = torch.abs(_pairwise(torch.sub, binary)) * valid_pair.to(torch.float32)
    rank_mask = _pairwise(torch.logical_xor, ranks <= lw._topn)
    return diff * rank_mask.to(torch.float32)


def dcg_pair_weights(lw, labels, ranks):
    """losses_impl.py:255-279."""
    is_valid = _is_valid(labels)
    valid_pair = _pairwise(torch.logical_and, is_valid)
    labels = torch.where(is_valid, labels, torch.zeros_like(labels))
    gain = lw._gain_fn(labels)
    if lw._normalized:
        gain = gain * inverse_max_dcg(labels, lw._gain_fn, lw._rank_discount_fn, lw._topn)
    pair_gain = _pairwise(torch.sub, gain) * valid_pair.to(torch.float32)
    list_size = labels.shape[1]
    topn = lw._topn or list_size
    pw = torch.abs(pair_gain) * lw._pair_rank_discount(ranks, topn)
    return pw * float(list_size)


def dcg_individual_weights(lw, labels, ranks):
    """losses_impl.py:281-296."""
    labels = torch.where(_is_valid(labels), labels, torch.zeros_like(labels))
    gain = lw._gain_fn(labels)
    if lw._normalized:
        gain = gain * inverse_max_dcg(labels, lw._gain_fn, lw._rank_discount_fn, lw._topn)
    return gain * lw._rank_discount_fn(ranks.to(torch.float32))


def pairwise_unreduced(loss, labels, logits, mask=None):
    """losses_impl.py:871-884 + :503-537: ([B,L,L] losses, [B,L,L] weights)."""
    if mask is None:
        mask = _is_valid(labels)
    ranks = compute_ranks(logits, mask)
    pairwise_labels = (_pairwise(torch.sub, labels) > 0).to(torch.float32)
    pairwise_labels = pairwise_labels * _pairwise(torch.logical_and, mask).to(torch.float32)
    pairwise_logits = _pairwise(torch.sub, logits)
    w = pairwise_labels
    if loss._lambda_weight is not None:
        w = w * loss._lambda_weight.pair_weights(labels, ranks)
    return loss._pairwise_loss(pairwise_logits), w.detach()


def pairwise_mse_unreduced(loss, labels, logits, mask=None):
    """losses_impl.py:972-998."""
    if mask is None:
        mask = _is_valid(labels)
    mse = torch.square(_pairwise(torch.sub, logits) - _pairwise(torch.sub, labels))
    l = labels.shape[1]
    w = (1.0 - torch.eye(l, dtype=torch.float32, device=labels.device)).unsqueeze(0)
    w = w * _pairwise(torch.logical_and, mask).to(torch.float32)
    if loss._lambda_weight is not None:
        w = w * loss._lambda_weight.pair_weights(labels, compute_ranks(logits, mask))
    return mse, w.detach()


def softmax_precompute(loss, labels, logits, weights, mask=None):
    """losses_impl.py:1122-1137."""
    from . import losses_impl
    if mask is None:
        mask = _is_valid(labels)
    ranks = compute_ranks(logits, mask)
    labels = torch.where(mask, labels, torch.zeros_like(labels))
    logits = torch.where(mask, logits, math.log(_EPSILON) * torch.ones_like(logits))
    if isinstance(loss._lambda_weight, losses_impl.DCGLambdaWeight):
        labels = loss._lambda_weight.individual_weights(labels, ranks)
    if weights is not None:
        labels = labels * weights
    return labels, logits


def softmax_unreduced(labels, logits, mask=None):
    """losses_impl.py:1139-1158."""
    if mask is None:
        mask = _is_valid(labels)
    label_sum = labels.sum(dim=1, keepdim=True)
    nonzero = label_sum.reshape(-1) > 0.
    padded = torch.where(nonzero.unsqueeze(1), labels, _EPSILON * torch.ones_like(labels))
    padded = torch.where(mask, padded, torch.zeros_like(padded))
    p = _safe_div(padded, padded.sum(dim=1, keepdim=True))
    losses = -(p * torch.log_softmax(logits, dim=1)).sum(dim=1)
    return losses, label_sum.reshape(-1)


def neural_sort(logits, mask=None):
    """losses_impl.py:1716-1801: the [B, L, L] relaxed permutation matrix (API parity; the fused losses
    never build it)."""
    logits = torch.as_tensor(logits, dtype=torch.float32)
    if mask is None:
        mask = torch.ones_like(logits, dtype=torch.bool)
    mask = torch.as_tensor(mask, device=logits.device).to(torch.bool)
    logits = torch.where(mask, logits, torch.zeros_like(logits))
    n_valid = mask.to(torch.int32).sum(dim=1, keepdim=True)
    diff = torch.abs(logits.unsqueeze(2) - logits.unsqueeze(1))
    valid_pair = _pairwise(torch.logical_and, mask)
    diff = torch.where(valid_pair, diff, torch.zeros_like(diff))
    diff_sum = diff.sum(dim=1, keepdim=True)
    masked_range = torch.cumsum(mask.to(torch.int32), dim=1)
    scaling = (n_valid + 1 - 2 * masked_range).to(torch.float32).unsqueeze(2)
    p_logits = scaling * logits.unsqueeze(1) - diff_sum
    p_logits = torch.where(valid_pair, p_logits, torch.full_like(p_logits, -math.inf))
    p_logits = torch.where(_pairwise(torch.logical_or, mask), p_logits, torch.zeros_like(p_logits))
    order = torch.argsort(mask.to(torch.int32), dim=1, descending=True, stable=True)
    p_logits = torch.gather(p_logits, 1, order.unsqueeze(2).expand_as(p_logits))
    return torch.softmax(p_logits, dim=-1)


def gumbel_neural_sort(logits, sample_size=8, temperature=1.0, seed=None):
    """losses_impl.py:1804-1847: [B, sample_size, L, L]."""
    logits = torch.as_tensor(logits, dtype=torch.float32)
    b, l = logits.shape
    gen = torch.Generator(device=logits.device)
    if seed is not None:
        gen.manual_seed(int(seed))
    else:
        gen.seed()
    u = torch.rand((b, sample_size, l), generator=gen, device=logits.device)
    gumbel = -torch.log(-torch.log(u + 1e-20) + 1e-20)
    sampled = (logits.unsqueeze(1) + gumbel).reshape(b * sample_size, l)
    return neural_sort(sampled / temperature).reshape(b, sample_size, l, l)
