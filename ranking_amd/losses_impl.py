"""Mirror of ``tensorflow_ranking/python/losses_impl.py`` for the hot-path losses.

Same class names, constructor arguments and method protocol as the reference
(`compute`, `compute_per_list`, `compute_unreduced_loss`, `normalize_weights`,
`get_logits`, `pair_weights`, `individual_weights`), but every reduced entry
point runs ONE fused gfx950 kernel (forward + backward) instead of a chain of
`[B, L, L]` TensorFlow ops.  Tensors are torch tensors on a HIP device.

The two methods whose *contract* is a materialised `[B, L, L]` tensor
(`_PairwiseLoss.compute_unreduced_loss`, `_LambdaWeight.pair_weights`) are
API-parity utilities implemented with torch device ops in `_materialized.py`;
no reduced path uses them.
"""
from __future__ import annotations

import abc
import os
import math
from typing import Callable, Optional

import torch

from . import _materialized as _mat
from . import _ops
from . import utils

_EPSILON = 1e-10


class Reduction:
    """tf.compat.v1.losses.Reduction values (strings identical to TF's)."""
    NONE = 'none'
    SUM = 'weighted_sum'
    MEAN = 'weighted_mean'
    SUM_OVER_BATCH_SIZE = 'weighted_sum_over_batch_size'
    SUM_OVER_NONZERO_WEIGHTS = 'weighted_sum_by_nonzero_weights'
    SUM_BY_NONZERO_WEIGHTS = 'weighted_sum_by_nonzero_weights'

    @classmethod
    def all(cls):
        return (cls.NONE, cls.SUM, cls.MEAN, cls.SUM_OVER_BATCH_SIZE, cls.SUM_BY_NONZERO_WEIGHTS)


def _safe_div(num, den):
    den_t = torch.as_tensor(den, dtype=num.dtype, device=num.device)
    ok = den_t != 0
    return torch.where(ok, num / torch.where(ok, den_t, torch.ones_like(den_t)), torch.zeros_like(num))


def compute_weighted_loss(losses, weights, reduction):
    """tf.compat.v1.losses.compute_weighted_loss semantics (call sites
    losses_impl.py:813,1167)."""
    weights = torch.as_tensor(weights, dtype=losses.dtype, device=losses.device)
    weighted = losses * weights
    if reduction == Reduction.NONE:
        return weighted
    total = weighted.sum()
    bw = torch.broadcast_to(weights, weighted.shape)
    if reduction == Reduction.SUM:
        return total
    if reduction == Reduction.MEAN:
        return _safe_div(total, bw.sum())
    if reduction == Reduction.SUM_BY_NONZERO_WEIGHTS:
        return _safe_div(total, (bw != 0).sum().to(losses.dtype))
    if reduction == Reduction.SUM_OVER_BATCH_SIZE:
        return total / weighted.numel()
    raise ValueError('Invalid reduction: {}'.format(reduction))


# --- default gain / discount functions (named so that the kernels can recognise them).
def _identity_gain(label):
    return label


def _inverse_rank(rank):
    return 1. / rank


def _pow2_minus_1(label):
    return torch.pow(torch.tensor(2.0, dtype=label.dtype, device=label.device), label) - 1.


def _inverse_log1p(rank):
    return 1. / torch.log1p(rank)


_KNOWN_GAINS = {}


def register_gain_kind(fn: Callable, kind: int):
    """Lets the kernels evaluate a known gain function in-register."""
    _KNOWN_GAINS[fn] = kind


register_gain_kind(_identity_gain, _ops.GAIN_IDENTITY)
register_gain_kind(_pow2_minus_1, _ops.GAIN_POW2M1)


def _gain_args(gain_fn, clean_labels_fn):
    """(gain_kind, gains tensor or None) for a gain callable."""
    kind = _KNOWN_GAINS.get(gain_fn)
    if kind is not None:
        return kind, None
    return _ops.GAIN_CUSTOM, gain_fn(clean_labels_fn()).to(torch.float32).contiguous()


def _check_tensor_shapes(tensors):
    """losses_impl.py:52-58."""
    first = tensors[0]
    for t in tensors:
        if t.dim() != 2:
            raise ValueError('Shape %s must have rank 2' % (tuple(t.shape),))
        if t.shape != first.shape:
            raise ValueError('Shapes %s and %s are incompatible' % (tuple(t.shape), tuple(first.shape)))


def approx_ranks(logits):
    """losses_impl.py:77-106 (API parity; torch device ops)."""
    return _mat.approx_ranks(logits)


def inverse_max_dcg(labels, gain_fn=_pow2_minus_1, rank_discount_fn=_inverse_log1p, topn=None):
    """losses_impl.py:109-134 (API parity)."""
    return _mat.inverse_max_dcg(labels, gain_fn, rank_discount_fn, topn)


def ndcg(labels, ranks=None, perm_mat=None):
    """losses_impl.py:137-167 (API parity)."""
    if ranks is not None and perm_mat is not None:
        raise ValueError('Cannot use both ranks and perm_mat simultaneously.')
    return _mat.ndcg(labels, ranks, perm_mat)


# ------------------------------------------------------------- lambda weights
class _LambdaWeight(object, metaclass=abc.ABCMeta):
    """losses_impl.py:170-207."""

    @abc.abstractmethod
    def pair_weights(self, labels, ranks):
        raise NotImplementedError('Calling an abstract method.')

    def individual_weights(self, labels, ranks):
        del ranks
        return labels


class LabelDiffLambdaWeight(_LambdaWeight):
    """losses_impl.py:210-217."""

    def pair_weights(self, labels, ranks):
        del ranks
        return _mat.label_diff_pair_weights(labels)


class AbstractDCGLambdaWeight(_LambdaWeight):
    """losses_impl.py:219-296."""

    def __init__(self, topn=None, gain_fn=_identity_gain, rank_discount_fn=_inverse_rank,
                 normalized=False):
        self._topn = topn
        self._gain_fn = gain_fn
        self._rank_discount_fn = rank_discount_fn
        self._normalized = normalized

    @abc.abstractmethod
    def _pair_rank_discount(self, ranks, topn):
        raise NotImplementedError('Calling an abstract method.')

    def pair_weights(self, labels, ranks):
        _check_tensor_shapes([labels, ranks])
        return _mat.dcg_pair_weights(self, labels, ranks)

    def individual_weights(self, labels, ranks):
        _check_tensor_shapes([labels, ranks])
        return _mat.dcg_individual_weights(self, labels, ranks)


class DCGLambdaWeight(AbstractDCGLambdaWeight):
    """losses_impl.py:299-369."""

    def __init__(self, topn=None, gain_fn=_identity_gain, rank_discount_fn=_inverse_rank,
                 normalized=False, smooth_fraction=0.):
        super().__init__(topn, gain_fn, rank_discount_fn, normalized)
        if not 0. <= smooth_fraction <= 1.:
            raise ValueError('smooth_fraction %s should be in range [0, 1].' % smooth_fraction)
        self._smooth_fraction = smooth_fraction

    def _pair_rank_discount(self, ranks, topn):
        return _mat.dcg_pair_rank_discount(self, ranks, topn)

    # -- what the fused kernels need
    def _kernel_args(self, labels, list_size, device):
        kind, gains = _gain_args(
            self._gain_fn,
            lambda: torch.where(labels >= 0, labels, torch.zeros_like(labels)))
        return dict(lambda_kind=_ops.LAMBDA_DCG, topn=self._topn or 0,
                    smooth_fraction=self._smooth_fraction, normalized=self._normalized,
                    gain_kind=kind, gains=gains,
                    discount=_ops.rank_table(self._rank_discount_fn, list_size + 1, device))


class DCGLambdaWeightV2(AbstractDCGLambdaWeight):
    """losses_impl.py:372-394; fused as TFR_LAMBDA_DCG_V2."""
    _lambda_kind = _ops.LAMBDA_DCG_V2

    def _pair_rank_discount(self, ranks, topn):
        return _mat.dcg_v2_pair_rank_discount(self, ranks, topn)

    def _kernel_args(self, labels, list_size, device):
        kind, gains = _gain_args(
            self._gain_fn,
            lambda: torch.where(labels >= 0, labels, torch.zeros_like(labels)))
        return dict(lambda_kind=self._lambda_kind, topn=self._topn or 0, smooth_fraction=0.0,
                    normalized=self._normalized, gain_kind=kind, gains=gains,
                    discount=_ops.rank_table(self._rank_discount_fn, list_size + 1, device))


class YetiDCGLambdaWeight(DCGLambdaWeightV2):
    """losses_impl.py:397-407; fused as TFR_LAMBDA_YETI_DCG."""
    _lambda_kind = _ops.LAMBDA_YETI_DCG

    def pair_weights(self, labels, ranks):
        pw = super().pair_weights(labels, ranks)
        ranks = torch.as_tensor(ranks, device=pw.device)
        return pw * (torch.abs(ranks.unsqueeze(2) - ranks.unsqueeze(1)) == 1).to(pw.dtype)


def _is_greater_equal_1(label):
    return label >= 1.0


class PrecisionLambdaWeight(_LambdaWeight):
    """losses_impl.py:410-454; fused as TFR_LAMBDA_PRECISION (positive_fn evaluated on the host side
    into a 0/1 gain array)."""

    def __init__(self, topn, positive_fn=_is_greater_equal_1):
        self._topn = topn
        self._positive_fn = positive_fn

    def pair_weights(self, labels, ranks):
        _check_tensor_shapes([labels, ranks])
        return _mat.precision_pair_weights(self, labels, ranks)

    def _kernel_args(self, labels, list_size, device):
        if not self._topn:
            raise ValueError('PrecisionLambdaWeight needs topn')
        clean = torch.where(labels >= 0, labels, torch.zeros_like(labels))
        gains = self._positive_fn(clean).to(torch.float32).contiguous()
        return dict(lambda_kind=_ops.LAMBDA_PRECISION, topn=self._topn, smooth_fraction=0.0, normalized=False,
                    gain_kind=_ops.GAIN_CUSTOM, gains=gains,
                    discount=torch.ones(list_size + 1, dtype=torch.float32, device=device))


def _lambda_kernel_args(lambda_weight, labels, list_size, device):
    if lambda_weight is None:
        return dict(lambda_kind=_ops.LAMBDA_NONE)
    if isinstance(lambda_weight, (DCGLambdaWeight, DCGLambdaWeightV2, PrecisionLambdaWeight)):
        return lambda_weight._kernel_args(labels, list_size, device)
    if isinstance(lambda_weight, LabelDiffLambdaWeight):
        return dict(lambda_kind=_ops.LAMBDA_LABELDIFF)
    return None   # not fusable: caller falls back to the materialised API-parity path


def _compute_ranks(logits, is_valid):
    """losses_impl.py:483-500 via the sort kernel (invalid strictly last)."""
    _check_tensor_shapes([logits, is_valid])
    ranks, _ = _ops.sort_ranks(logits, None, is_valid, None, want_ranks=True, want_order=False)
    return ranks


# -------------------------------------------------------------------- sampler
class _GumbelSampleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, uniform, seed, offset, sample_size, temperature):
        out = _ops.gumbel_sample(logits, labels, None, uniform, seed, offset, sample_size, temperature)
        ctx.save_for_backward(out, labels)
        ctx.sample_size, ctx.temperature = sample_size, temperature
        return out

    @staticmethod
    def backward(ctx, g):
        out, labels = ctx.saved_tensors
        d = _ops.gumbel_sample_bwd(out, labels, None, g.contiguous(), ctx.sample_size, ctx.temperature)
        return d, None, None, None, None, None, None


class GumbelSampler(object):
    """losses_impl.py:540-644.  Noise: in-kernel Philox4x32-10 keyed by ``seed``
    and a per-call offset, or an injected ``uniform`` [B, S, L] tensor (the
    reference's TF random stream cannot be reproduced)."""

    def __init__(self, name=None, sample_size=8, temperature=1.0, seed=None, ragged=False):
        self._name = name
        self._sample_size = sample_size
        self._temperature = temperature
        self._seed = seed
        self._ragged = ragged
        self._calls = 0

    def sample(self, labels, logits, weights=None, uniform=None):
        if self._ragged:
            labels, logits, weights, _ = utils.ragged_to_dense(labels, logits, weights)
        labels = _ops.require_device(torch.as_tensor(labels), 'labels').to(torch.float32)
        logits = _ops.require_device(logits, 'logits').to(torch.float32)
        if labels.dim() != 2:
            raise NotImplementedError('3-D (subtopic) labels are outside the hot path.')
        b, l = labels.shape
        s = self._sample_size
        seed = self._seed
        if seed is None:
            seed = int(torch.empty((), dtype=torch.int64).random_().item())
        offset = self._calls
        self._calls += 1
        sampled = _GumbelSampleFn.apply(logits, labels.contiguous(), uniform, seed, offset, s,
                                        float(self._temperature))
        expanded_labels = labels.unsqueeze(1).expand(b, s, l).reshape(b * s, l)
        expanded_weights = weights
        if expanded_weights is not None:
            w = torch.as_tensor(expanded_weights, dtype=torch.float32, device=labels.device)
            w = w.reshape(b, 1, 1) if w.dim() == 1 else w.unsqueeze(1)
            expanded_weights = w.expand(b, s, w.shape[-1]).reshape(b * s, -1)
        return expanded_labels, sampled, expanded_weights


# --------------------------------------------------------------- autograd glue
class _PerListLossFn(torch.autograd.Function):
    """Wraps a fused fwd+bwd kernel: the kernel already produced
    d(per_list[b])/d(logits[b, :]); backward only scales rows by the upstream."""

    @staticmethod
    def forward(ctx, logits, runner):
        per_list, dlogits, aux = runner(logits.detach(), logits.requires_grad)
        ctx.has_grad = dlogits is not None
        if ctx.has_grad:
            ctx.save_for_backward(dlogits)
        ctx.mark_non_differentiable(*[t for t in aux if t is not None])
        return (per_list,) + tuple(aux)

    @staticmethod
    def backward(ctx, g, *unused):
        if not ctx.has_grad:
            return None, None
        (dlogits,) = ctx.saved_tensors
        return dlogits * g.unsqueeze(1), None


# -------------------------------------------------------------------- losses
class _RankingLoss(object, metaclass=abc.ABCMeta):
    """losses_impl.py:652-860."""

    def __init__(self, name, lambda_weight=None, temperature=1.0, ragged=False):
        self._name = name
        self._lambda_weight = lambda_weight
        self._temperature = temperature
        self._ragged = ragged

    @property
    def name(self):
        return self._name

    def _prepare_and_validate_params(self, labels, logits, weights, mask):
        if self._ragged:
            labels, logits, weights, mask = utils.ragged_to_dense(labels, logits, weights)
        logits = _ops.require_device(torch.as_tensor(logits), 'logits')
        labels = torch.as_tensor(labels, dtype=torch.float32, device=logits.device)
        if weights is None:
            weights = 1.0
        weights = torch.as_tensor(weights, dtype=torch.float32, device=logits.device)
        if mask is not None:
            mask = torch.as_tensor(mask, device=logits.device).to(torch.bool)
        _check_tensor_shapes([labels, logits] + ([mask] if mask is not None else []))
        return labels, logits.to(torch.float32), weights, mask

    def compute_unreduced_loss(self, labels, logits, mask=None):
        labels, logits, _, mask = self._prepare_and_validate_params(labels, logits, None, mask)
        return self._compute_unreduced_loss_impl(labels, logits, mask)

    @abc.abstractmethod
    def _compute_unreduced_loss_impl(self, labels, logits, mask=None):
        raise NotImplementedError('Calling an abstract method.')

    def normalize_weights(self, labels, weights):
        if self._ragged:
            labels, _, weights, _ = utils.ragged_to_dense(labels, None, weights)
        labels = torch.as_tensor(labels, dtype=torch.float32)
        if weights is not None:
            weights = torch.as_tensor(weights, dtype=torch.float32, device=labels.device)
        return self._normalize_weights_impl(labels, weights)

    def _normalize_weights_impl(self, labels, weights):
        del labels
        return 1.0 if weights is None else weights

    def get_logits(self, logits):
        return torch.as_tensor(logits) / self._temperature

    def compute(self, labels, logits, weights, reduction, mask=None):
        """losses_impl.py:787-814 (temperature applied inside the kernel)."""
        labels, logits, _, mask = self._prepare_and_validate_params(labels, logits, None, mask)
        if weights is not None:
            weights = torch.as_tensor(weights, dtype=torch.float32, device=logits.device)
        return self._compute_reduced(labels, logits, weights, reduction, mask)

    @abc.abstractmethod
    def _compute_reduced(self, labels, logits, weights, reduction, mask):
        raise NotImplementedError

    @abc.abstractmethod
    def compute_per_list(self, labels, logits, weights, mask=None):
        raise NotImplementedError('Calling an abstract method.')

    def eval_metric(self, labels, logits, weights, mask=None):
        """losses_impl.py:838-860: weighted mean of the losses."""
        return self.compute(labels, logits, weights, Reduction.MEAN, mask)


# ------------------------------------------------------------------ pairwise
class _PairwiseLoss(_RankingLoss, metaclass=abc.ABCMeta):
    """losses_impl.py:863-930.  Equal scores: the ranks behind a lambda weight are ``_compute_ranks(logits,
    shuffle_ties=True)`` in the reference (:483-500, a random order of tied scores in every call); the fused kernels rank
    ties by index unless ``shuffle_ties`` is set on the loss object -- then by a counter-based hash of a tie seed, list and
    item (``seed``: None = a new seed per call from torch's host generator), on the general workgroup kernel instead of the
    LambdaRank fast paths.  Only a lambda weight looks at ranks."""

    _fused_kind = None   # subclasses with a fused kernel set this
    shuffle_ties = False
    seed = None
    # 'analytic' (default): d loss / d t of a pair at t = s_i - s_j = 0 is the function's derivative, -sigma(0) = -1/2.
    # 'reference': what TF autodiff returns for the reference's own formula relu(-t) + log1p(exp(-|t|)) (:936-940) at
    # exactly t = 0 -- zero (relu'(0) = 0, sign(0) = 0): an all-equal logit vector (a zero-initialised output layer) then
    # gets no pairwise-logistic gradient, as in a reference run.  Only PairwiseLogisticLoss differs; the mode runs the general
    # kernels (TFR_PAIR_TIED_ZERO declines the LambdaRank fast paths).  TFR_TIED_GRADIENT=reference sets the default.
    tied_gradient = os.environ.get('TFR_TIED_GRADIENT', 'analytic')

    def _kind(self):
        if self.tied_gradient not in ('analytic', 'reference'):
            raise ValueError("tied_gradient must be 'analytic' or 'reference', got %r" % (self.tied_gradient,))
        if self._fused_kind == _ops.PAIR_LOGISTIC and self.tied_gradient == 'reference':
            return _ops.PAIR_LOGISTIC | _ops.PAIR_TIED_ZERO
        return self._fused_kind

    def _tie_seed(self):
        if not self.shuffle_ties or self._lambda_weight is None:
            return 0
        if self.seed is None:
            return _fresh_tie_seed()
        return (int(self.seed) & 0x7fffffff) or 1

    @abc.abstractmethod
    def _pairwise_loss(self, pairwise_logits):
        raise NotImplementedError('Calling an abstract method.')

    def _normalize_weights_impl(self, labels, weights):
        if weights is None:
            weights = 1.
        weights = torch.where(utils.is_label_valid(labels), torch.ones_like(labels) * weights,
                              torch.zeros_like(labels))
        return weights.unsqueeze(2)

    def _compute_unreduced_loss_impl(self, labels, logits, mask=None):
        """[B, L, L] losses and weights: API-parity path (torch device ops)."""
        return _mat.pairwise_unreduced(self, labels, logits, mask)

    # fused path -----------------------------------------------------------
    def _fused(self, labels, logits, weights, mask, apply_temperature=True, want_aux=True):
        """Returns (list_loss [B] differentiable, row_loss [B,L], row_weight [B,L], nnz [B])."""
        b, l = logits.shape
        lam = _lambda_kernel_args(self._lambda_weight, labels, l, logits.device)
        if lam is None or self._fused_kind is None:
            return None
        item_w = list_w = None
        if weights is not None and weights.dim() > 0 and weights.numel() > 1:
            if weights.dim() == 2 and weights.shape == (b, l):
                item_w = weights
            elif weights.numel() == b:
                list_w = weights.reshape(b)
            else:
                raise ValueError('weights shape %s incompatible with [%d, %d]' % (tuple(weights.shape), b, l))
        elif weights is not None:
            list_w = torch.broadcast_to(weights.reshape(()), (b,)).contiguous()
        temperature = self._temperature if apply_temperature else 1.0
        tie_seed = self._tie_seed()

        def runner(lg, want_grad):
            row_loss, row_weight, nnz, d = _ops.pairwise_logistic(
                lg, labels, mask, item_w, list_w, temperature=temperature, want_grad=want_grad,
                want_aux=want_aux, loss_kind=self._kind(), tie_seed=tie_seed, **lam)
            return row_loss.sum(dim=1), d, (row_loss, row_weight, nnz)

        return _PerListLossFn.apply(logits, runner)

    def _compute_reduced(self, labels, logits, weights, reduction, mask):
        fused = self._fused(labels, logits, weights, mask,
                            want_aux=reduction in (Reduction.MEAN, Reduction.SUM_BY_NONZERO_WEIGHTS))
        if fused is None:
            losses, loss_weights = self._compute_unreduced_loss_impl(labels, self.get_logits(logits), mask)
            w = self._normalize_weights_impl(labels, weights) * loss_weights
            return compute_weighted_loss(losses, w, reduction)
        list_loss, row_loss, row_weight, nnz = fused
        b, l = logits.shape
        total = list_loss.sum()
        if reduction == Reduction.SUM:
            return total
        if reduction == Reduction.MEAN:
            return _safe_div(total, row_weight.sum())
        if reduction == Reduction.SUM_BY_NONZERO_WEIGHTS:
            return _safe_div(total, nnz.sum())
        if reduction == Reduction.SUM_OVER_BATCH_SIZE:
            return total / float(b * l * l)
        raise ValueError('Invalid reduction: {}'.format(reduction))

    def compute_per_list(self, labels, logits, weights, mask=None):
        """losses_impl.py:886-915 (NB: no temperature, like the reference)."""
        labels, logits, weights, mask = self._prepare_and_validate_params(labels, logits, weights, mask)
        fused = self._fused(labels, logits, weights, mask, apply_temperature=False)
        if fused is None:
            losses, loss_weights = self._compute_unreduced_loss_impl(labels, logits, mask)
            w = self._normalize_weights_impl(labels, weights) * loss_weights
            per_list_weights = w.sum(dim=(1, 2))
            return _safe_div((losses * w).sum(dim=(1, 2)), per_list_weights), per_list_weights
        list_loss, _, row_weight, _ = fused
        per_list_weights = row_weight.sum(dim=1)
        return _safe_div(list_loss, per_list_weights), per_list_weights


class PairwiseLogisticLoss(_PairwiseLoss):
    """losses_impl.py:933-940; fused kernel tfr_pairwise_logistic_f32."""
    _fused_kind = _ops.PAIR_LOGISTIC

    def _pairwise_loss(self, pairwise_logits):
        return torch.relu(-pairwise_logits) + torch.log1p(torch.exp(-torch.abs(pairwise_logits)))


class PairwiseHingeLoss(_PairwiseLoss):
    """losses_impl.py:943-948; fused kernel tfr_pairwise_loss_f32(TFR_PAIR_HINGE)."""
    _fused_kind = _ops.PAIR_HINGE

    def _pairwise_loss(self, pairwise_logits):
        return torch.relu(1 - pairwise_logits)


class PairwiseSoftZeroOneLoss(_PairwiseLoss):
    """losses_impl.py:951-958; fused kernel tfr_pairwise_loss_f32(TFR_PAIR_SOFT_ZERO_ONE)."""
    _fused_kind = _ops.PAIR_SOFT_ZERO_ONE

    def _pairwise_loss(self, pairwise_logits):
        return torch.where(pairwise_logits > 0, 1. - torch.sigmoid(pairwise_logits),
                           torch.sigmoid(-pairwise_logits))


# ------------------------------------------------------------------ listwise
class PairwiseMSELoss(_PairwiseLoss):
    """losses_impl.py:961-998; fused kernel tfr_pairwise_loss_f32(TFR_PAIR_MSE): every ordered pair of two
    distinct valid items, ((s_i - s_j) - (y_i - y_j))^2."""
    _fused_kind = _ops.PAIR_MSE

    def _pairwise_loss(self, pairwise_logits):
        return None

    def _compute_unreduced_loss_impl(self, labels, logits, mask=None):
        return _mat.pairwise_mse_unreduced(self, labels, logits, mask)


class _ListwiseLoss(_RankingLoss):
    """losses_impl.py:1001-1033."""

    def _normalize_weights_impl(self, labels, weights):
        if weights is None:
            return 1.0
        weights = torch.as_tensor(weights, dtype=torch.float32, device=labels.device)
        is_valid = utils.is_label_valid(labels)
        labels = torch.where(is_valid, labels, torch.zeros_like(labels))
        return _safe_div((weights * labels).sum(dim=1, keepdim=True), labels.sum(dim=1, keepdim=True))

    def _compute_reduced(self, labels, logits, weights, reduction, mask):
        losses, loss_weights = self._unreduced(labels, logits, mask, self._temperature)
        w = self._normalize_weights_impl(labels, weights) * loss_weights
        return compute_weighted_loss(losses, w, reduction)

    def compute_per_list(self, labels, logits, weights, mask=None):
        # NB: the reference does not apply the temperature here (losses_impl.py:1017-1033).
        labels, logits, weights, mask = self._prepare_and_validate_params(labels, logits, weights, mask)
        losses, loss_weights = self._unreduced(labels, logits, mask, 1.0)
        w = self._normalize_weights_impl(labels, weights) * loss_weights
        return losses.squeeze(1), w.squeeze(1)

    def _compute_unreduced_loss_impl(self, labels, logits, mask=None):
        # `logits` are already temperature-scaled by the caller (reference protocol).
        return self._unreduced(labels, logits, mask, 1.0)

    @abc.abstractmethod
    def _unreduced(self, labels, logits, mask, temperature):
        """([B,1] losses, [B,1] weights) with `logits / temperature` applied in-kernel."""


class ApproxNDCGLoss(_ListwiseLoss):
    """losses_impl.py:1579-1603; fused kernel tfr_approx_ndcg_f32."""

    def __init__(self, name, lambda_weight=None, temperature=0.1, ragged=False):
        super().__init__(name, lambda_weight, temperature, ragged)

    def _unreduced(self, labels, logits, mask, temperature):
        def runner(lg, want_grad):
            loss, weight, d = _ops.approx_ndcg(lg, labels, mask, None, temperature, 0, want_grad)
            return loss, d, (weight,)
        loss, weight = _PerListLossFn.apply(logits, runner)
        return loss.unsqueeze(1), weight.unsqueeze(1)


class ListMLELambdaWeight(_LambdaWeight):
    """losses_impl.py:457-480."""

    def __init__(self, rank_discount_fn):
        self._rank_discount_fn = rank_discount_fn

    def pair_weights(self, labels, ranks):
        pass

    def individual_weights(self, labels, ranks):
        _check_tensor_shapes([labels, ranks])
        return torch.ones_like(labels) * self._rank_discount_fn(ranks.to(torch.float32))


_TIE_GEN = {'seed': None, 'gen': None}


def _fresh_tie_seed() -> int:
    """A new non-zero 31-bit tie seed per call from a PRIVATE host generator seeded from torch.initial_seed() (round 6,
    ADVICE r5: drawing from torch's global generator perturbed the caller's own random streams -- dropout, shuffling -- by
    one draw per loss call).  The sequence restarts when torch.manual_seed CHANGES the initial seed (re-seeding with the
    same value continues it -- nothing observable distinguishes the two states without consuming the global stream): the role TF's
    graph-level seed plays for the op seed 37 of losses_impl.py:1558-1561.  (A host draw: under hipGraph capture the seed
    of the captured step is replayed -- pass ``seed`` for a fixed order, or re-capture.)"""
    base = torch.initial_seed()
    if _TIE_GEN['seed'] != base:
        _TIE_GEN['seed'], _TIE_GEN['gen'] = base, torch.Generator().manual_seed((base ^ 0x5DEECE66D) & (2 ** 63 - 1))
    return int(torch.randint(1, 2 ** 31 - 1, (1,), generator=_TIE_GEN['gen']).item())


class ListMLELoss(_ListwiseLoss):
    """losses_impl.py:1541-1576; fused kernel tfr_list_mle_f32.  Equal labels: the reference sorts with
    shuffle_ties=True (:1558-1561) -- a new random order of the tied items in every step; here ``shuffle_ties`` (default
    True) orders them by a counter-based hash of a tie seed, list and item (``seed``: None = a new seed per call, an int =
    the same order in every call); ``shuffle_ties=False`` keeps index order.  The TF random stream is not reproducible."""

    shuffle_ties = True
    seed = None

    def _tie_seed(self):
        if not self.shuffle_ties:
            return 0
        return _fresh_tie_seed() if self.seed is None else (int(self.seed) & 0x7fffffff) or 1

    def _pos_weight(self, list_size, device):
        if isinstance(self._lambda_weight, ListMLELambdaWeight):
            return _ops.rank_table(self._lambda_weight._rank_discount_fn, list_size, device)
        return None

    def _unreduced(self, labels, logits, mask, temperature):
        pw = self._pos_weight(logits.shape[1], logits.device)
        tie_seed = self._tie_seed()

        def runner(lg, want_grad):
            loss, d = _ops.list_mle(lg, labels, mask, pw, None, temperature, want_grad, tie_seed=tie_seed)
            return loss, d, ()
        (loss,) = _PerListLossFn.apply(logits, runner)
        return loss.unsqueeze(1), torch.ones_like(loss).unsqueeze(1)


class UniqueSoftmaxLoss(_ListwiseLoss):
    """losses_impl.py:1250-1281; fused kernel tfr_unique_softmax_f32."""

    def _unreduced(self, labels, logits, mask, temperature):
        def runner(lg, want_grad):
            loss, d = _ops.unique_softmax(lg, labels, mask, None, temperature, want_grad)
            return loss, d, ()
        (loss,) = _PerListLossFn.apply(logits, runner)
        return loss.unsqueeze(1), torch.ones_like(loss).unsqueeze(1)


class CircleLoss(_ListwiseLoss):
    """losses_impl.py:1036-1116; fused kernel tfr_circle_loss_f32 (the [L, L] pair matrix collapses to a
    sort by label and two scans because the pair exponent is separable)."""

    def __init__(self, name, lambda_weight=None, gamma=64, margin=0.25, ragged=False):
        super().__init__(name, lambda_weight=lambda_weight, temperature=1.0, ragged=ragged)
        self._margin = margin
        self._gamma = gamma

    def get_logits(self, logits):
        return torch.clamp(torch.as_tensor(logits), 0., 1.)

    def _compute_reduced(self, labels, logits, weights, reduction, mask):
        # compute() is the one entry point that applies get_logits (:808); the clip runs in the kernel
        losses, loss_weights = self._unreduced(labels, logits, mask, 1.0, clip=True)
        w = self._normalize_weights_impl(labels, weights) * loss_weights
        return compute_weighted_loss(losses, w, reduction)

    def _unreduced(self, labels, logits, mask, temperature, clip=False):
        out = {}

        def runner(lg, want_grad):
            loss, weight, d = _ops.circle_loss(lg, labels, mask, None, self._gamma, self._margin, clip, want_grad)
            out['w'] = weight
            return loss, d, ()
        (loss,) = _PerListLossFn.apply(logits, runner)
        return loss.unsqueeze(1), out['w'].unsqueeze(1)


def neural_sort(logits, name=None, mask=None):
    """losses_impl.py:1716-1801."""
    return _mat.neural_sort(logits, mask)


def gumbel_neural_sort(logits, name=None, sample_size=8, temperature=1.0, seed=None):
    """losses_impl.py:1804-1847."""
    return _mat.gumbel_neural_sort(logits, sample_size, temperature, seed)


class _NeuralSortLoss(_ListwiseLoss):
    _kind = None

    def _unreduced(self, labels, logits, mask, temperature):
        def runner(lg, want_grad):
            loss, d = _ops.neural_sort_loss(self._kind, lg, labels, mask, None, temperature, want_grad)
            return loss, d, ()
        (loss,) = _PerListLossFn.apply(logits, runner)
        m = mask if mask is not None else utils.is_label_valid(labels)
        nonzero = torch.where(m, labels, torch.zeros_like(labels)).sum(dim=1, keepdim=True) > 0.0
        return loss.unsqueeze(1), nonzero.to(torch.float32)


class NeuralSortCrossEntropyLoss(_NeuralSortLoss):
    """losses_impl.py:1635-1673; fused kernel tfr_neural_sort_loss_f32(TFR_NEURAL_SORT_CE)."""
    _kind = _ops.NEURAL_SORT_CE


class NeuralSortNDCGLoss(_NeuralSortLoss):
    """losses_impl.py:1676-1713; fused kernel tfr_neural_sort_loss_f32(TFR_NEURAL_SORT_NDCG)."""
    _kind = _ops.NEURAL_SORT_NDCG


class ApproxMRRLoss(_ListwiseLoss):
    """losses_impl.py:1606-1632; fused kernel tfr_approx_mrr_f32."""

    def __init__(self, name, lambda_weight=None, temperature=0.1, ragged=False):
        super().__init__(name, lambda_weight, temperature, ragged)

    def _unreduced(self, labels, logits, mask, temperature):
        def runner(lg, want_grad):
            loss, weight, d = _ops.approx_mrr(lg, labels, mask, None, temperature, want_grad)
            return loss, d, (weight,)
        loss, weight = _PerListLossFn.apply(logits, runner)
        return loss.unsqueeze(1), weight.unsqueeze(1)


class SoftmaxLoss(_ListwiseLoss):
    """losses_impl.py:1119-1197; fused kernel tfr_softmax_loss_f32."""
    _poly_epsilon = 0.0

    def _lambda_args(self, labels, logits, mask):
        """Kernel arguments of `individual_weights`: only a DCGLambdaWeight is active in SoftmaxLoss.precompute
        (losses_impl.py:1132-1134); any other lambda weight (V2, Precision, LabelDiff, ...) is ignored there."""
        lam = (self._lambda_weight._kernel_args(labels, logits.shape[1], logits.device)
               if isinstance(self._lambda_weight, DCGLambdaWeight)
               else dict(lambda_kind=_ops.LAMBDA_NONE))
        lam.pop('smooth_fraction', None)
        if lam.get('gain_kind') == _ops.GAIN_CUSTOM:
            m = mask if mask is not None else labels >= 0
            clean = torch.where(m, labels, torch.zeros_like(labels))
            clean = torch.where(clean >= 0, clean, torch.zeros_like(clean))
            lam['gains'] = self._lambda_weight._gain_fn(clean).to(torch.float32).contiguous()
        return lam

    def _run(self, labels, logits, weights, mask, temperature):
        lam = self._lambda_args(labels, logits, mask)

        def runner(lg, want_grad):
            loss, weight, d = _ops.softmax_loss(lg, labels, mask, weights, temperature=temperature,
                                                want_grad=want_grad, poly_epsilon=self._poly_epsilon, **lam)
            # kernel's dlogits = weight * dloss/dlogits; per_list output is weight*loss.
            return loss * weight, d, (loss, weight)
        weighted, loss, weight = _PerListLossFn.apply(logits, runner)
        return weighted, loss, weight

    def precompute(self, labels, logits, weights, mask=None):
        """losses_impl.py:1122-1137 (API parity; torch device ops)."""
        return _mat.softmax_precompute(self, labels, logits, weights, mask)

    def _unreduced(self, labels, logits, mask, temperature):
        _, loss, weight = self._run(labels, logits, None, mask, temperature)
        return loss, weight

    def _compute_unreduced_loss_impl(self, labels, logits, mask=None):
        # reference protocol: inputs are the *precomputed* labels/logits.
        return _mat.softmax_unreduced(labels, logits, mask)

    def _reduce(self, weighted, loss, weight, reduction):
        # sum(loss * weight) must stay attached to the fused backward: use `weighted`.
        total = weighted.sum()
        if reduction == Reduction.NONE:
            return weighted
        if reduction == Reduction.SUM:
            return total
        if reduction == Reduction.MEAN:
            return _safe_div(total, weight.sum())
        if reduction == Reduction.SUM_BY_NONZERO_WEIGHTS:
            return _safe_div(total, (weight != 0).sum().to(total.dtype))
        if reduction == Reduction.SUM_OVER_BATCH_SIZE:
            return total / weighted.numel()
        raise ValueError('Invalid reduction: {}'.format(reduction))

    def compute(self, labels, logits, weights, reduction, mask=None):
        """losses_impl.py:1160-1167."""
        labels, logits, weights, mask = self._prepare_and_validate_params(labels, logits, weights, mask)
        weighted, loss, weight = self._run(labels, logits, weights, mask, self._temperature)
        return self._reduce(weighted, loss, weight, reduction)

    def _compute_reduced(self, labels, logits, weights, reduction, mask):   # pragma: no cover
        raise AssertionError('SoftmaxLoss overrides compute')

    def compute_per_list(self, labels, logits, weights, mask=None):
        """losses_impl.py:1178-1189 (temperature IS applied here, :1187)."""
        labels, logits, weights, mask = self._prepare_and_validate_params(labels, logits, weights, mask)
        weighted, loss, weight = self._run(labels, logits, weights, mask, self._temperature)
        return _AttachFn.apply(loss, weighted, weight), weight

    def compute_unreduced_loss(self, labels, logits, mask=None):
        """losses_impl.py:1191-1197."""
        labels, logits, _, mask = self._prepare_and_validate_params(labels, logits, None, mask)
        weighted, loss, weight = self._run(labels, logits, None, mask, self._temperature)
        return _AttachFn.apply(loss, weighted, weight), weight


class _AttachFn(torch.autograd.Function):
    """Returns `loss` (values) while routing gradients through `weighted = loss *
    weight`, whose backward is the fused kernel's: d loss = d weighted / weight."""

    @staticmethod
    def forward(ctx, loss, weighted, weight):
        ctx.save_for_backward(weight)
        return loss.clone()

    @staticmethod
    def backward(ctx, g):
        (weight,) = ctx.saved_tensors
        return None, _safe_div(g, weight), None


class _PointwiseLoss(_RankingLoss):
    """losses_impl.py:1284-1321; reduced entry points on the fused kernel tfr_pointwise_loss_f32."""
    _fused_kind = None

    def _normalize_weights_impl(self, labels, weights):
        if weights is None:
            weights = 1.
        return torch.where(utils.is_label_valid(labels), torch.ones_like(labels) * weights,
                           torch.zeros_like(labels))

    def _fused(self, labels, logits, weights, mask, temperature):
        """(list_loss [B] differentiable, list_weight [B], list_nnz [B])."""
        b, l = logits.shape
        item_w = list_w = None
        if weights is not None:
            weights = torch.as_tensor(weights, dtype=torch.float32, device=logits.device)
            if weights.dim() == 2 and weights.shape == (b, l):
                item_w = weights
            elif weights.numel() == b:
                list_w = weights.reshape(b)
            elif weights.numel() == 1:
                list_w = torch.broadcast_to(weights.reshape(()), (b,)).contiguous()
            else:
                raise ValueError('weights shape %s incompatible with [%d, %d]' % (tuple(weights.shape), b, l))

        def runner(lg, want_grad):
            loss, weight, nnz, d = _ops.pointwise_loss(self._fused_kind, lg, labels, mask, item_w, list_w,
                                                       temperature, want_grad)
            return loss, d, (weight, nnz)
        return _PerListLossFn.apply(logits, runner)

    def _compute_reduced(self, labels, logits, weights, reduction, mask):
        if self._fused_kind is None:
            losses, loss_weights = self._compute_unreduced_loss_impl(labels, self.get_logits(logits), mask)
            return compute_weighted_loss(losses, self._normalize_weights_impl(labels, weights) * loss_weights,
                                         reduction)
        list_loss, list_weight, nnz = self._fused(labels, logits, weights, mask, self._temperature)
        total = list_loss.sum()
        if reduction == Reduction.SUM:
            return total
        if reduction == Reduction.MEAN:
            return _safe_div(total, list_weight.sum())
        if reduction == Reduction.SUM_BY_NONZERO_WEIGHTS:
            return _safe_div(total, nnz.sum())
        if reduction == Reduction.SUM_OVER_BATCH_SIZE:
            return total / float(logits.numel())
        raise ValueError('Invalid reduction: {}'.format(reduction))

    def compute_per_list(self, labels, logits, weights, mask=None):
        labels, logits, weights, mask = self._prepare_and_validate_params(labels, logits, weights, mask)
        if self._fused_kind is None:
            losses, loss_weights = self._compute_unreduced_loss_impl(labels, logits, mask)
            w = self._normalize_weights_impl(labels, weights) * loss_weights
            per_list_weights = w.sum(dim=1)
            return _safe_div((losses * w).sum(dim=1), per_list_weights), per_list_weights
        list_loss, list_weight, _ = self._fused(labels, logits, weights, mask, 1.0)   # no temperature (:1300-1321)
        return _safe_div(list_loss, list_weight), list_weight


class SigmoidCrossEntropyLoss(_PointwiseLoss):
    """losses_impl.py:1425-1446."""
    _fused_kind = _ops.POINT_SIGMOID_CE

    def __init__(self, name, temperature=1.0, ragged=False):
        super().__init__(name, None, temperature, ragged)

    def _compute_unreduced_loss_impl(self, labels, logits, mask=None):
        if mask is None:
            mask = utils.is_label_valid(labels)
        labels = torch.where(mask, labels, torch.zeros_like(labels))
        logits = torch.where(mask, logits, torch.zeros_like(logits))
        losses = torch.relu(logits) - logits * labels + torch.log1p(torch.exp(-torch.abs(logits)))
        return losses, mask.to(torch.float32)


class MeanSquaredLoss(_PointwiseLoss):
    """losses_impl.py:1449-1469 (temperature is not used by this loss)."""
    _fused_kind = _ops.POINT_MSE

    def __init__(self, name, ragged=False):
        super().__init__(name, None, temperature=1.0, ragged=ragged)

    def _compute_unreduced_loss_impl(self, labels, logits, mask=None):
        if mask is None:
            mask = utils.is_label_valid(labels)
        labels = torch.where(mask, labels, torch.zeros_like(labels))
        logits = torch.where(mask, logits, torch.zeros_like(logits))
        return torch.square(labels - logits), mask.to(torch.float32)


class PolyOneSoftmaxLoss(SoftmaxLoss):
    """losses_impl.py:1200-1247; fused kernel tfr_poly1_softmax_loss_f32."""

    def __init__(self, name, lambda_weight=None, epsilon=1.0, temperature=1.0, ragged=False):
        super().__init__(name, lambda_weight=lambda_weight, temperature=temperature, ragged=ragged)
        self._epsilon = epsilon
        self._poly_epsilon = float(epsilon)

    def _compute_unreduced_loss_impl(self, labels, logits, mask=None):
        losses, weights = _mat.softmax_unreduced(labels, logits, mask)
        if mask is None:
            mask = utils.is_label_valid(labels)
        label_sum = labels.sum(dim=1, keepdim=True)
        padded = torch.where(label_sum > 0.0, labels, 1e-10 * torch.ones_like(labels))
        padded = torch.where(mask, padded, torch.zeros_like(padded))
        p = _safe_div(padded, padded.sum(dim=1, keepdim=True))
        pt = (p * torch.softmax(logits, dim=-1)).sum(dim=-1)
        return losses + self._epsilon * (1 - pt).reshape(losses.shape), weights
