"""Mirror of ``tensorflow_ranking/python/keras/losses.py`` for the hot-path losses.

Same keys, class names, constructor arguments, ``__call__(y_true, y_pred,
sample_weight)``, ``get_config`` / ``from_config`` as the reference.  ``y_pred``
must be a torch tensor on a HIP device; the returned scalar is differentiable
(torch autograd) and its backward is the fused kernel's.

Extra, not in the reference: ``loss_and_grad(y_true, y_pred, sample_weight)``
returns ``(loss, d loss / d y_pred)`` from a single kernel launch without
building an autograd graph -- the path a throughput-minded training loop (and
``bench.py``) uses: ``logits.backward(dlogits)``.
"""
from __future__ import annotations

import os
from typing import Any, Dict, List, Optional

import torch

from .. import _ops
from .. import losses_impl
from .. import utils as _tfr_utils
from . import utils


class Reduction:
    """tf.keras.losses.Reduction values."""
    AUTO = 'auto'
    NONE = 'none'
    SUM = 'sum'
    SUM_OVER_BATCH_SIZE = 'sum_over_batch_size'

    @classmethod
    def all(cls):
        return (cls.AUTO, cls.NONE, cls.SUM, cls.SUM_OVER_BATCH_SIZE)

    @classmethod
    def validate(cls, key):
        if key not in cls.all():
            raise ValueError('Invalid Reduction Key: {}. Expected keys are "{}"'.format(key, cls.all()))


class RankingLossKey(object):
    """keras/losses.py:25-48."""
    PAIRWISE_HINGE_LOSS = 'pairwise_hinge_loss'
    PAIRWISE_LOGISTIC_LOSS = 'pairwise_logistic_loss'
    PAIRWISE_SOFT_ZERO_ONE_LOSS = 'pairwise_soft_zero_one_loss'
    PAIRWISE_MSE_LOSS = 'pairwise_mse_loss'
    YETI_LOGISTIC_LOSS = 'yeti_logistic_loss'
    SOFTMAX_LOSS = 'softmax_loss'
    CALIBRATED_SOFTMAX_LOSS = 'calibrated_softmax_loss'
    UNIQUE_SOFTMAX_LOSS = 'unique_softmax_loss'
    SIGMOID_CROSS_ENTROPY_LOSS = 'sigmoid_cross_entropy_loss'
    MEAN_SQUARED_LOSS = 'mean_squared_loss'
    ORDINAL_LOSS = 'ordinal_loss'
    LIST_MLE_LOSS = 'list_mle_loss'
    APPROX_NDCG_LOSS = 'approx_ndcg_loss'
    APPROX_MRR_LOSS = 'approx_mrr_loss'
    GUMBEL_APPROX_NDCG_LOSS = 'gumbel_approx_ndcg_loss'
    COUPLED_RANKDISTIL_LOSS = 'coupled_rankdistil_loss'

    @classmethod
    def all_keys(cls) -> List[str]:
        return [v for k, v in vars(cls).items() if k.isupper()]


def get(loss: str, reduction: str = Reduction.AUTO, lambda_weight=None, name: Optional[str] = None,
        **kwargs):
    """keras/losses.py:51-111.  Keys outside the hot path raise ValueError like
    any unsupported key does in the reference (:109)."""
    loss_kwargs = {'reduction': reduction, 'name': name}
    loss_kwargs.update(kwargs)
    with_lambda = {'lambda_weight': lambda_weight}
    with_lambda.update(loss_kwargs)
    key_to_cls = {
        RankingLossKey.SIGMOID_CROSS_ENTROPY_LOSS: SigmoidCrossEntropyLoss,
        RankingLossKey.MEAN_SQUARED_LOSS: MeanSquaredLoss,
        RankingLossKey.APPROX_NDCG_LOSS: ApproxNDCGLoss,
        RankingLossKey.APPROX_MRR_LOSS: ApproxMRRLoss,
        RankingLossKey.GUMBEL_APPROX_NDCG_LOSS: GumbelApproxNDCGLoss,
    }
    key_to_cls_with_lambda = {
        RankingLossKey.PAIRWISE_HINGE_LOSS: PairwiseHingeLoss,
        RankingLossKey.PAIRWISE_LOGISTIC_LOSS: PairwiseLogisticLoss,
        RankingLossKey.PAIRWISE_SOFT_ZERO_ONE_LOSS: PairwiseSoftZeroOneLoss,
        RankingLossKey.PAIRWISE_MSE_LOSS: PairwiseMSELoss,
        RankingLossKey.YETI_LOGISTIC_LOSS: YetiLogisticLoss,
        RankingLossKey.SOFTMAX_LOSS: SoftmaxLoss,
        RankingLossKey.CALIBRATED_SOFTMAX_LOSS: CalibratedSoftmaxLoss,
        RankingLossKey.LIST_MLE_LOSS: ListMLELoss,
        RankingLossKey.UNIQUE_SOFTMAX_LOSS: UniqueSoftmaxLoss,
    }
    if loss in key_to_cls:
        return key_to_cls[loss](**loss_kwargs)
    if loss in key_to_cls_with_lambda:
        return key_to_cls_with_lambda[loss](**with_lambda)
    raise ValueError('unsupported loss: {}'.format(loss))


# ------------------------------------------------------------- lambda weights
@utils.register_keras_serializable()
class LabelDiffLambdaWeight(losses_impl.LabelDiffLambdaWeight):
    """keras/losses.py:114-123."""

    def __init__(self, **kwargs):
        super().__init__()

    def get_config(self) -> Dict[str, Any]:
        return {}


@utils.register_keras_serializable()
class DCGLambdaWeight(losses_impl.DCGLambdaWeight):
    """keras/losses.py:126-148."""

    def __init__(self, topn=None, gain_fn=None, rank_discount_fn=None, normalized=False,
                 smooth_fraction=0., **kwargs):
        gain_fn = gain_fn or utils.identity
        rank_discount_fn = rank_discount_fn or utils.inverse
        super().__init__(topn, gain_fn, rank_discount_fn, normalized, smooth_fraction)

    def get_config(self) -> Dict[str, Any]:
        return {'topn': self._topn, 'gain_fn': self._gain_fn,
                'rank_discount_fn': self._rank_discount_fn, 'normalized': self._normalized,
                'smooth_fraction': self._smooth_fraction}


@utils.register_keras_serializable()
class NDCGLambdaWeight(DCGLambdaWeight):
    """keras/losses.py:197-212."""

    def __init__(self, topn=None, gain_fn=None, rank_discount_fn=None, smooth_fraction=0., **kwargs):
        super().__init__(topn, gain_fn or utils.pow_minus_1, rank_discount_fn or utils.log2_inverse,
                         normalized=True, smooth_fraction=smooth_fraction)


@utils.register_keras_serializable()
class NDCGLambdaWeightV2(losses_impl.DCGLambdaWeightV2):
    """keras/losses.py:151-169."""

    def __init__(self, topn=None, gain_fn=None, rank_discount_fn=None, **kwargs):
        super().__init__(topn, gain_fn or utils.pow_minus_1, rank_discount_fn or utils.log2_inverse, normalized=True)

    def get_config(self) -> Dict[str, Any]:
        return {'topn': self._topn, 'gain_fn': self._gain_fn, 'rank_discount_fn': self._rank_discount_fn}


@utils.register_keras_serializable()
class YetiDCGLambdaWeight(losses_impl.YetiDCGLambdaWeight):
    """keras/losses.py:172-195."""

    def __init__(self, topn=None, gain_fn=None, rank_discount_fn=None, normalized=False, **kwargs):
        super().__init__(topn, gain_fn or utils.pow_minus_1, rank_discount_fn or utils.log2_inverse,
                         normalized=normalized)

    def get_config(self) -> Dict[str, Any]:
        return {'topn': self._topn, 'gain_fn': self._gain_fn, 'rank_discount_fn': self._rank_discount_fn,
                'normalized': self._normalized}


@utils.register_keras_serializable()
class PrecisionLambdaWeight(losses_impl.PrecisionLambdaWeight):
    """keras/losses.py:215-231."""

    def __init__(self, topn=None, positive_fn=None, **kwargs):
        super().__init__(topn, positive_fn or utils.is_greater_equal_1)

    def get_config(self) -> Dict[str, Any]:
        return {'topn': self._topn, 'positive_fn': self._positive_fn}


@utils.register_keras_serializable()
class ListMLELambdaWeight(losses_impl.ListMLELambdaWeight):
    """keras/losses.py:233-244: the serialisable form of the position weights of ListMLELoss."""

    def __init__(self, rank_discount_fn=None, **kwargs):
        super().__init__(rank_discount_fn)

    def get_config(self) -> Dict[str, Any]:
        return {'rank_discount_fn': self._rank_discount_fn}


# ------------------------------------------------------------------- helpers
# The reduced scalar of loss_and_grad (TFR_LOSS_SUM_FUSED): 0 = a tfr_list_dot_f32 launch over the per-list vectors behind
# every loss launch (rounds 1-3); 1 (default) = the same, except that Softmax leaves per-wavefront partials and adds 8 192 of
# them instead of 2 x 65 536 values (-8 us of a 36 us launch); 2 = every loss takes the scalar from its own launch
# (tfr_*_sum_f32, ApproxNDCG included).  The in-launch form was round 4's default for ApproxNDCG (-4 us then) and measured
# SLOWER everywhere in round 5 (profiles/r05_headline_and_sum_ab.txt): the last wavefront's store -> ticket -> group sum -> ticket ->
# final sum is five dependent device-memory round trips -- +16 us behind a 5-25 us softmax launch, +2.6 us behind the 44 us
# LambdaRank kernel, and +10.8 us behind the ApproxNDCG kernel once that got faster and the step stopped relaunching the
# ordering kernels (0.1188 vs 0.1080 ms per step, twice each on one box) -- against ~3 us for the reduction launch.
_LOSS_SUM_MODE = int(os.environ.get('TFR_LOSS_SUM_FUSED', '1'))
_LOSS_SUM_FUSED = _LOSS_SUM_MODE >= 2                  # ApproxNDCG family: in-launch sum
_LOSS_SUM_PARTIALS = _LOSS_SUM_MODE >= 1               # Softmax: per-wavefront partials + a short dot
_LOSS_SUM_ALL = _LOSS_SUM_MODE >= 2                    # every loss
_CONST_CACHE = _ops.DeviceConstCache(64)        # graph-safe: entries a hipGraph capture has read are never evicted


def _const_vector(n: int, value: float, device) -> torch.Tensor:
    """A cached, read-only [n] fp32 tensor filled with `value` (the per-list scale of an unweighted batch):
    saves a fill launch in every loss_and_grad call."""
    return _CONST_CACHE.get((int(n), float(value), str(device)),
                            lambda: torch.full((n,), float(value), dtype=torch.float32, device=device))


def _keras_reduce(weighted, reduction):
    """tf.keras losses_utils.reduce_weighted_loss."""
    if reduction == Reduction.NONE:
        return weighted
    if reduction == Reduction.SUM:
        return weighted.sum()
    return weighted.sum() / weighted.numel()          # AUTO / SUM_OVER_BATCH_SIZE


def _apply_sample_weight(losses, sample_weight):
    """Keras weight broadcasting: squeeze a trailing 1 of a rank+1 weight,
    expand a rank-1 weight."""
    if sample_weight is None:
        return losses
    w = torch.as_tensor(sample_weight, dtype=losses.dtype, device=losses.device)
    if w.dim() == losses.dim() + 1 and w.shape[-1] == 1:
        w = w.squeeze(-1)
    elif w.dim() == losses.dim() - 1 and w.dim() > 0:
        w = w.unsqueeze(-1)
    return losses * w


def _densify(loss, y_true, y_pred, sample_weight):
    """Ragged -> dense + mask (keras path: reductions then see the padded shape)."""
    mask = None
    if loss._ragged:
        y_true, y_pred, sample_weight, mask = _tfr_utils.ragged_to_dense(y_true, y_pred, sample_weight)
    y_pred = _ops.require_device(torch.as_tensor(y_pred), 'y_pred').to(torch.float32)
    y_true = torch.as_tensor(y_true, dtype=torch.float32, device=y_pred.device)
    if sample_weight is not None:
        sample_weight = torch.as_tensor(sample_weight, dtype=torch.float32, device=y_pred.device)
    losses_impl._check_tensor_shapes([y_true, y_pred])
    return y_true, y_pred, sample_weight, mask


class _RankingLoss(object):
    """keras/losses.py:247-285."""

    def __init__(self, reduction=Reduction.AUTO, name=None, ragged=False):
        Reduction.validate(reduction)
        self.reduction = reduction
        self.name = name
        self._loss = None
        self._ragged = ragged

    def _scale(self, numel):
        return 1.0 if self.reduction == Reduction.SUM else 1.0 / float(numel)

    def __call__(self, y_true, y_pred, sample_weight=None):
        if self._loss is None:
            raise ValueError('self._loss is not defined. Please use a subclass.')
        return self._call_impl(y_true, y_pred, sample_weight)

    def call(self, y_true, y_pred):
        """keras/losses.py:274-280: unreduced losses * weights."""
        y_pred = self._loss.get_logits(y_pred)
        losses, weights = self._loss.compute_unreduced_loss(labels=y_true, logits=y_pred)
        return losses * weights

    def _call_impl(self, y_true, y_pred, sample_weight):
        sw = self._loss.normalize_weights(y_true, sample_weight)
        return _keras_reduce(_apply_sample_weight(self.call(y_true, y_pred), sw), self.reduction)

    def get_config(self) -> Dict[str, Any]:
        return {'reduction': self.reduction, 'name': self.name, 'ragged': self._ragged}

    @classmethod
    def from_config(cls, config, custom_objects=None):
        return cls(**config)


class _LambdaConfigMixin(object):
    def get_config(self) -> Dict[str, Any]:
        config = super().get_config()
        config.update({'lambda_weight': utils.serialize_keras_object(self._lambda_weight),
                       'temperature': self._temperature})
        return config

    @classmethod
    def from_config(cls, config, custom_objects=None):
        config = dict(config)
        config['lambda_weight'] = utils.deserialize_keras_object(config.get('lambda_weight'),
                                                                 custom_objects)
        return cls(**config)


# ------------------------------------------------------------------ pairwise
class _PairwiseLoss(_LambdaConfigMixin, _RankingLoss):
    """keras/losses.py:288-335."""
    _impl_cls = None

    def __init__(self, reduction=Reduction.AUTO, name=None, lambda_weight=None, temperature=1.0,
                 ragged=False, **kwargs):
        super().__init__(reduction, name, ragged)
        self._lambda_weight = lambda_weight
        self._temperature = temperature
        self._loss = self._impl_cls(name='{}_impl'.format(name) if name else None,
                                    lambda_weight=lambda_weight, temperature=temperature, ragged=ragged)

    def call(self, y_true, y_pred):
        """keras/losses.py:324-335: sum over j -> [B, L] (materialised API path)."""
        y_pred = self._loss.get_logits(y_pred)
        losses, weights = self._loss.compute_unreduced_loss(labels=y_true, logits=y_pred)
        return (losses * weights).sum(dim=2)

    def _call_impl(self, y_true, y_pred, sample_weight):
        y_true, y_pred, sample_weight, mask = _densify(self, y_true, y_pred, sample_weight)
        fused = None
        if self.reduction != Reduction.NONE:
            fused = self._loss._fused(y_true, y_pred, sample_weight, mask, want_aux=False)
        if fused is None:   # NONE reduction or a loss without a fused kernel
            saved, self._loss._ragged = self._loss._ragged, False
            try:
                sw = self._loss.normalize_weights(y_true, sample_weight)
                losses, weights = self._loss.compute_unreduced_loss(
                    labels=y_true, logits=self._loss.get_logits(y_pred), mask=mask)
            finally:
                self._loss._ragged = saved
            rows = (losses * weights).sum(dim=2)
            return _keras_reduce(_apply_sample_weight(rows, sw), self.reduction)
        list_loss = fused[0]
        b, l = y_pred.shape
        return list_loss.sum() * self._scale(b * l)

    def loss_and_grad(self, y_true, y_pred, sample_weight=None):
        """Single-launch training path: (scalar loss, dloss/dy_pred [B, L])."""
        if self.reduction == Reduction.NONE:
            raise ValueError('loss_and_grad needs a scalar reduction')
        y_true, y_pred, sample_weight, mask = _densify(self, y_true, y_pred, sample_weight)
        b, l = y_pred.shape
        scale = self._scale(b * l)
        lam = losses_impl._lambda_kernel_args(self._lambda_weight, y_true, l, y_pred.device)
        if lam is None or self._loss._fused_kind is None:
            raise NotImplementedError('no fused kernel for this loss / lambda weight')
        item_w = None
        list_w = _const_vector(b, scale, y_pred.device)
        if sample_weight is not None:
            if sample_weight.dim() == 2 and sample_weight.shape == (b, l):
                item_w = sample_weight
            else:
                list_w = list_w * torch.broadcast_to(sample_weight.reshape(-1), (b,))
        tie_seed = self._loss._tie_seed()
        if not _LOSS_SUM_ALL:
            _, _, _, dlogits, list_loss = _ops.pairwise_logistic(
                y_pred.detach(), y_true, mask, item_w, list_w, temperature=self._temperature,
                want_grad=True, want_rows=False, want_aux=False, want_list=True, loss_kind=self._loss._kind(),
                tie_seed=tie_seed, **lam)
            return _ops.list_dot(list_loss), dlogits       # [B] per-list sums: nothing [B, L]-sized for the loss
        # the reduced scalar out of the same launch (round 5: every loss, not only ApproxNDCG)
        _, _, _, dlogits, _, total = _ops.pairwise_logistic(
            y_pred.detach(), y_true, mask, item_w, list_w, temperature=self._temperature,
            want_grad=True, want_rows=False, want_aux=False, want_list=True, loss_kind=self._loss._kind(),
            want_sum=True, tie_seed=tie_seed, **lam)
        return total, dlogits


@utils.register_keras_serializable()
class PairwiseLogisticLoss(_PairwiseLoss):
    """keras/losses.py:405-469."""
    _impl_cls = losses_impl.PairwiseLogisticLoss


@utils.register_keras_serializable()
class PairwiseHingeLoss(_PairwiseLoss):
    """keras/losses.py:338-402 (materialised path; SURVEY 8f "next")."""
    _impl_cls = losses_impl.PairwiseHingeLoss


@utils.register_keras_serializable()
class PairwiseSoftZeroOneLoss(_PairwiseLoss):
    """keras/losses.py:472-536 (materialised path; SURVEY 8f "next")."""
    _impl_cls = losses_impl.PairwiseSoftZeroOneLoss


@utils.register_keras_serializable()
class PairwiseMSELoss(_PairwiseLoss):
    """keras/losses.py:540-606."""
    _impl_cls = losses_impl.PairwiseMSELoss


@utils.register_keras_serializable()
class YetiLogisticLoss(_PairwiseLoss):
    """keras/losses.py:608-718: Gumbel-perturbed scores (sample_size copies of every list) into
    PairwiseLogisticLoss weighted by YetiDCGLambdaWeight (neighbour pairs only)."""
    _impl_cls = losses_impl.PairwiseLogisticLoss

    def __init__(self, reduction=Reduction.AUTO, name=None, lambda_weight=None, temperature=1.0,
                 sample_size=8, gumbel_temperature=1.0, seed=None, ragged=False):
        lambda_weight = lambda_weight or YetiDCGLambdaWeight()
        super().__init__(reduction, name, lambda_weight, temperature, ragged)
        self._sample_size = sample_size
        self._gumbel_temperature = gumbel_temperature
        self._seed = seed
        self._gumbel_sampler = losses_impl.GumbelSampler(name=name, sample_size=sample_size,
                                                         temperature=gumbel_temperature, seed=seed, ragged=ragged)

    def get_config(self) -> Dict[str, Any]:
        config = super().get_config()
        config.update({'sample_size': self._sample_size, 'gumbel_temperature': self._gumbel_temperature,
                       'seed': self._seed})
        return config

    def __call__(self, y_true, y_pred, sample_weight=None):
        gbl_labels, gbl_logits, gbl_weights = self._gumbel_sampler.sample(y_true, y_pred, weights=sample_weight)
        return super().__call__(gbl_labels, gbl_logits, gbl_weights)

    def loss_and_grad(self, y_true, y_pred, sample_weight=None):
        raise NotImplementedError('YetiLogisticLoss: use __call__ (the Gumbel sampler needs autograd)')


# ------------------------------------------------------------------ listwise
class _ListwiseLoss(_LambdaConfigMixin, _RankingLoss):
    """keras/losses.py:721-756."""

    def __init__(self, reduction=Reduction.AUTO, name=None, lambda_weight=None, temperature=1.0,
                 ragged=False, **kwargs):
        super().__init__(reduction, name, ragged)
        self._lambda_weight = lambda_weight
        self._temperature = temperature


@utils.register_keras_serializable()
class SoftmaxLoss(_ListwiseLoss):
    """keras/losses.py:759-832."""

    def __init__(self, reduction=Reduction.AUTO, name=None, lambda_weight=None, temperature=1.0,
                 ragged=False):
        super().__init__(reduction, name, lambda_weight, temperature, ragged)
        self._loss = losses_impl.SoftmaxLoss(name='{}_impl'.format(name) if name else None,
                                             lambda_weight=lambda_weight, temperature=temperature,
                                             ragged=ragged)

    def _call_impl(self, y_true, y_pred, sample_weight):
        """keras/losses.py:824-832: compute_per_list, then the Keras reduction."""
        y_true, y_pred, sample_weight, mask = _densify(self, y_true, y_pred, sample_weight)
        weighted, _, _ = self._loss._run(y_true, y_pred, sample_weight, mask, self._temperature)
        return _keras_reduce(weighted, self.reduction)

    def loss_and_grad(self, y_true, y_pred, sample_weight=None):
        if self.reduction == Reduction.NONE:
            raise ValueError('loss_and_grad needs a scalar reduction')
        y_true, y_pred, sample_weight, mask = _densify(self, y_true, y_pred, sample_weight)
        b, l = y_pred.shape
        scale = self._scale(b)
        w = _const_vector(b, scale, y_pred.device) if sample_weight is None else sample_weight * scale
        lam = self._loss._lambda_args(y_true, y_pred, mask)      # only a DCGLambdaWeight is active, like __call__
        if not _LOSS_SUM_PARTIALS:
            loss, weight, dlogits = _ops.softmax_loss(y_pred.detach(), y_true, mask, w,
                                                      temperature=self._temperature, want_grad=True,
                                                      poly_epsilon=self._loss._poly_epsilon, **lam)
            return _ops.list_dot(loss, weight), dlogits
        _, _, dlogits, total = _ops.softmax_loss(y_pred.detach(), y_true, mask, w,
                                                 temperature=self._temperature, want_grad=True,
                                                 poly_epsilon=self._loss._poly_epsilon,
                                                 want_sum=True if _LOSS_SUM_ALL else 'partials', **lam)
        return total, dlogits


@utils.register_keras_serializable()
class CalibratedSoftmaxLoss(SoftmaxLoss):
    """keras/losses.py:835-936: softmax over the list PLUS one virtual item with score 0 and label
    ``virtual_label`` (item weight 1 when per-item weights are given):
    ``loss = -sum_i y_i log(e^{s_i} / (1 + sum_j e^{s_j})) - y_0 log(1 / (1 + sum_j e^{s_j}))``.
    It is the softmax kernel on a list of L + 1 items; the virtual item's gradient column is dropped."""

    def __init__(self, reduction=Reduction.AUTO, name=None, lambda_weight=None, temperature=1.0,
                 virtual_label=0.0):
        super().__init__(reduction, name, lambda_weight, temperature, False)
        assert virtual_label >= 0, 'Virtual label must be non-negative.'
        self._virtual_label = virtual_label

    def get_config(self) -> Dict[str, Any]:
        config = super().get_config()
        config.pop('ragged', None)                      # not a constructor argument of this class (:876-883)
        config.update({'virtual_label': self._virtual_label})
        return config

    def _with_virtual_item(self, y_true, y_pred, sample_weight):
        y_pred = _ops.require_device(torch.as_tensor(y_pred), 'y_pred').to(torch.float32)
        y_true = torch.as_tensor(y_true, dtype=torch.float32, device=y_pred.device)
        losses_impl._check_tensor_shapes([y_true, y_pred])
        b = y_true.shape[0]
        y_true = torch.cat([y_true, y_true.new_full((b, 1), float(self._virtual_label))], dim=1)
        y_pred = torch.cat([y_pred, y_pred.new_zeros((b, 1))], dim=1)
        if sample_weight is not None:
            sample_weight = torch.as_tensor(sample_weight, dtype=torch.float32, device=y_pred.device)
            if sample_weight.dim() == 2 and sample_weight.shape[1] > 1:
                sample_weight = torch.cat([sample_weight, sample_weight.new_ones((b, 1))], dim=1)
        return y_true, y_pred, sample_weight

    def __call__(self, y_true, y_pred, sample_weight=None):
        return super().__call__(*self._with_virtual_item(y_true, y_pred, sample_weight))

    def loss_and_grad(self, y_true, y_pred, sample_weight=None):
        loss, dlogits = super().loss_and_grad(*self._with_virtual_item(y_true, y_pred, sample_weight))
        return loss, dlogits[:, :-1].contiguous()


@utils.register_keras_serializable()
class ApproxNDCGLoss(_ListwiseLoss):
    """keras/losses.py:1165-1237."""

    def __init__(self, reduction=Reduction.AUTO, name=None, lambda_weight=None, temperature=0.1,
                 ragged=False):
        super().__init__(reduction, name, lambda_weight, temperature, ragged)
        self._loss = losses_impl.ApproxNDCGLoss(name='{}_impl'.format(name) if name else None,
                                                lambda_weight=lambda_weight, temperature=temperature,
                                                ragged=ragged)

    def _call_impl(self, y_true, y_pred, sample_weight):
        y_true, y_pred, sample_weight, mask = _densify(self, y_true, y_pred, sample_weight)
        sw = self._loss._normalize_weights_impl(y_true, sample_weight)      # [B,1] or 1.0
        losses, weights = self._loss._unreduced(y_true, y_pred, mask, self._temperature)
        return _keras_reduce(_apply_sample_weight(losses * weights, sw), self.reduction)

    def loss_and_grad(self, y_true, y_pred, sample_weight=None):
        if self.reduction == Reduction.NONE:
            raise ValueError('loss_and_grad needs a scalar reduction')
        y_true, y_pred, sample_weight, mask = _densify(self, y_true, y_pred, sample_weight)
        b = y_pred.shape[0]
        scale = self._scale(b)
        sw = self._loss._normalize_weights_impl(y_true, sample_weight)
        if torch.is_tensor(sw):
            list_scale = (sw.reshape(b) * scale).contiguous()
        else:
            list_scale = _const_vector(b, scale * float(sw), y_pred.device)
        # weight = [sum of labels > 0]; the kernel's loss of such a list is exactly 0 (all gains are 0): no `* weight`.
        # The reduced scalar sum_b loss_b * list_scale_b: one short reduction launch (default; TFR_LOSS_SUM_FUSED=2: out of
        # the loss launch itself, a fixed-order sum by the last workgroup to finish its forward pass -- see _LOSS_SUM_MODE).
        if not _LOSS_SUM_FUSED:
            loss, _, dlogits = _ops.approx_ndcg(y_pred.detach(), y_true, mask, list_scale, self._temperature, 0, True)
            return _ops.list_dot(loss, list_scale), dlogits
        _, _, dlogits, total = _ops.approx_ndcg(y_pred.detach(), y_true, mask, list_scale,
                                                self._temperature, 0, True, want_sum=True)
        return total, dlogits


@utils.register_keras_serializable()
class UniqueSoftmaxLoss(ApproxNDCGLoss):
    """keras/losses.py:939-1010."""

    def __init__(self, reduction=Reduction.AUTO, name=None, lambda_weight=None, temperature=1.0, ragged=False):
        _ListwiseLoss.__init__(self, reduction, name, lambda_weight, temperature, ragged)
        self._loss = losses_impl.UniqueSoftmaxLoss(name='{}_impl'.format(name) if name else None,
                                                   lambda_weight=lambda_weight, temperature=temperature,
                                                   ragged=ragged)

    def loss_and_grad(self, y_true, y_pred, sample_weight=None):
        if self.reduction == Reduction.NONE:
            raise ValueError('loss_and_grad needs a scalar reduction')
        y_true, y_pred, sample_weight, mask = _densify(self, y_true, y_pred, sample_weight)
        b = y_pred.shape[0]
        scale = self._scale(b)
        sw = self._loss._normalize_weights_impl(y_true, sample_weight)
        if torch.is_tensor(sw):
            list_scale = (sw.reshape(b) * scale).contiguous()
        else:
            list_scale = _const_vector(b, scale * float(sw), y_pred.device)
        if not _LOSS_SUM_ALL:
            loss, dlogits = _ops.unique_softmax(y_pred.detach(), y_true, mask, list_scale, self._temperature, True)
            return _ops.list_dot(loss, list_scale), dlogits
        _, dlogits, total = _ops.unique_softmax(y_pred.detach(), y_true, mask, list_scale, self._temperature, True,
                                                want_sum=True)
        return total, dlogits


@utils.register_keras_serializable()
class ListMLELoss(ApproxNDCGLoss):
    """keras/losses.py:1013-1091.  ``shuffle_ties`` / ``seed`` (not in the reference signature): see
    ``losses_impl.ListMLELoss`` -- equal labels in a new random order per call by default, as the reference sorts them."""

    def __init__(self, reduction=Reduction.AUTO, name=None, lambda_weight=None, temperature=1.0, ragged=False,
                 shuffle_ties=True, seed=None):
        _ListwiseLoss.__init__(self, reduction, name, lambda_weight, temperature, ragged)
        self._loss = losses_impl.ListMLELoss(name='{}_impl'.format(name) if name else None,
                                             lambda_weight=lambda_weight, temperature=temperature, ragged=ragged)
        self._loss.shuffle_ties, self._loss.seed = bool(shuffle_ties), seed

    def get_config(self):
        config = super().get_config()
        config.update({'shuffle_ties': self._loss.shuffle_ties, 'seed': self._loss.seed})
        return config

    def loss_and_grad(self, y_true, y_pred, sample_weight=None):
        if self.reduction == Reduction.NONE:
            raise ValueError('loss_and_grad needs a scalar reduction')
        y_true, y_pred, sample_weight, mask = _densify(self, y_true, y_pred, sample_weight)
        b = y_pred.shape[0]
        scale = self._scale(b)
        sw = self._loss._normalize_weights_impl(y_true, sample_weight)
        if torch.is_tensor(sw):
            list_scale = (sw.reshape(b) * scale).contiguous()
        else:
            list_scale = _const_vector(b, scale * float(sw), y_pred.device)
        pw = self._loss._pos_weight(y_pred.shape[1], y_pred.device)
        tie_seed = self._loss._tie_seed()
        if not _LOSS_SUM_ALL:
            loss, dlogits = _ops.list_mle(y_pred.detach(), y_true, mask, pw, list_scale, self._temperature, True,
                                          tie_seed=tie_seed)
            return _ops.list_dot(loss, list_scale), dlogits
        _, dlogits, total = _ops.list_mle(y_pred.detach(), y_true, mask, pw, list_scale, self._temperature, True,
                                          want_sum=True, tie_seed=tie_seed)
        return total, dlogits


@utils.register_keras_serializable()
class ApproxMRRLoss(ApproxNDCGLoss):
    """keras/losses.py:1094-1162."""

    def __init__(self, reduction=Reduction.AUTO, name=None, lambda_weight=None, temperature=0.1,
                 ragged=False):
        _ListwiseLoss.__init__(self, reduction, name, lambda_weight, temperature, ragged)
        self._loss = losses_impl.ApproxMRRLoss(name='{}_impl'.format(name) if name else None,
                                               lambda_weight=lambda_weight, temperature=temperature,
                                               ragged=ragged)

    def loss_and_grad(self, y_true, y_pred, sample_weight=None):
        if self.reduction == Reduction.NONE:
            raise ValueError('loss_and_grad needs a scalar reduction')
        y_true, y_pred, sample_weight, mask = _densify(self, y_true, y_pred, sample_weight)
        b = y_pred.shape[0]
        scale = self._scale(b)
        sw = self._loss._normalize_weights_impl(y_true, sample_weight)
        if torch.is_tensor(sw):
            list_scale = (sw.reshape(b) * scale).contiguous()
        else:
            list_scale = _const_vector(b, scale * float(sw), y_pred.device)
        loss, weight, dlogits = _ops.approx_mrr(y_pred.detach(), y_true, mask, list_scale, self._temperature, True)
        return torch.dot(loss * weight, list_scale), dlogits


@utils.register_keras_serializable()
class GumbelApproxNDCGLoss(ApproxNDCGLoss):
    """keras/losses.py:1241-1341."""

    def __init__(self, reduction=Reduction.AUTO, name=None, lambda_weight=None, temperature=0.1,
                 sample_size=8, gumbel_temperature=1.0, seed=None, ragged=False):
        super().__init__(reduction, name, lambda_weight, temperature=temperature, ragged=ragged)
        self._sample_size = sample_size
        self._gumbel_temperature = gumbel_temperature
        self._seed = seed
        self._gumbel_sampler = losses_impl.GumbelSampler(name=name, sample_size=sample_size,
                                                         temperature=gumbel_temperature, seed=seed,
                                                         ragged=ragged)

    def get_config(self) -> Dict[str, Any]:
        config = super().get_config()
        config.update({'sample_size': self._sample_size,
                       'gumbel_temperature': self._gumbel_temperature, 'seed': self._seed})
        return config

    def _call_impl(self, y_true, y_pred, sample_weight, uniform=None):
        gbl_labels, gbl_logits, gbl_weights = self._gumbel_sampler.sample(
            y_true, y_pred, weights=sample_weight, uniform=uniform)
        saved, self._ragged = self._ragged, False      # the sampler returned dense tensors
        try:
            return super()._call_impl(gbl_labels, gbl_logits, gbl_weights)
        finally:
            self._ragged = saved

    def __call__(self, y_true, y_pred, sample_weight=None, uniform=None):
        """``uniform``: optional injected U(0,1) noise [B, S, L] (parity runs)."""
        return self._call_impl(y_true, y_pred, sample_weight, uniform)

    def loss_and_grad(self, y_true, y_pred, sample_weight=None, uniform=None):
        y_true, y_pred, sample_weight, _ = _densify(self, y_true, y_pred, sample_weight)
        s = self._sample_size
        sampler = self._gumbel_sampler
        seed = sampler._seed if sampler._seed is not None else 0
        offset = sampler._calls
        sampler._calls += 1
        b, l = y_true.shape
        # Round 6: the Philox offset of the training step also has a device part that the backward launch advances -- a step
        # replayed from a hipGraph draws new noise (the host-side `offset` is frozen into the graph).  The counter is created
        # by the first call outside a capture; the sampler kernel also writes the labels of the S copies of every list.
        step = None
        if uniform is None and l <= 1024:
            key = str(y_pred.device)
            steps = sampler.__dict__.setdefault('_device_steps', {})
            step = steps.get(key)
            if step is None and not torch.cuda.is_current_stream_capturing():
                step = steps[key] = torch.zeros(1, dtype=torch.int64, device=y_pred.device)
        sampled, gl = _ops.gumbel_sample(y_pred.detach(), y_true, None, uniform, seed, offset, s,
                                         self._gumbel_temperature, step=step, want_labels=True)
        gw = None
        if sample_weight is not None:
            w = sample_weight.reshape(b, 1, 1) if sample_weight.dim() == 1 else sample_weight.unsqueeze(1)
            gw = w.expand(b, s, w.shape[-1]).reshape(b * s, -1)
        saved, self._ragged = self._ragged, False
        try:
            loss, d_sampled = ApproxNDCGLoss.loss_and_grad(self, gl, sampled, gw)
        finally:
            self._ragged = saved
        dlogits = _ops.gumbel_sample_bwd(sampled, y_true, None, d_sampled, s, self._gumbel_temperature, step_inc=step)
        return loss, dlogits


class _PointwiseLoss(_RankingLoss):
    """Keras pointwise losses on the fused kernel tfr_pointwise_loss_f32 (scalar reductions)."""

    def _weights_args(self, sample_weight, b, l, device):
        item_w = list_w = None
        if sample_weight is not None:
            w = torch.as_tensor(sample_weight, dtype=torch.float32, device=device)
            if w.dim() == 2 and w.shape == (b, l):
                item_w = w
            elif w.numel() == b:
                list_w = w.reshape(b)
            elif w.numel() == 1:
                list_w = torch.broadcast_to(w.reshape(()), (b,)).contiguous()
            else:
                raise ValueError('sample_weight shape %s incompatible with [%d, %d]' % (tuple(w.shape), b, l))
        return item_w, list_w

    def _call_impl(self, y_true, y_pred, sample_weight):
        if self.reduction == Reduction.NONE:
            saved_ragged = self._loss._ragged
            y_true, y_pred, sample_weight, mask = _densify(self, y_true, y_pred, sample_weight)
            self._loss._ragged = False
            try:
                sw = self._loss.normalize_weights(y_true, sample_weight)
                losses, weights = self._loss.compute_unreduced_loss(labels=y_true, logits=self._loss.get_logits(y_pred),
                                                                    mask=mask)
            finally:
                self._loss._ragged = saved_ragged
            return _apply_sample_weight(losses * weights, sw)
        y_true, y_pred, sample_weight, mask = _densify(self, y_true, y_pred, sample_weight)
        b, l = y_pred.shape
        list_loss, _, _ = self._loss._fused(y_true, y_pred, sample_weight, mask, self._loss._temperature)
        return list_loss.sum() * self._scale(b * l)

    def loss_and_grad(self, y_true, y_pred, sample_weight=None):
        """Single-launch training path: (scalar loss, dloss/dy_pred [B, L])."""
        if self.reduction == Reduction.NONE:
            raise ValueError('loss_and_grad needs a scalar reduction')
        y_true, y_pred, sample_weight, mask = _densify(self, y_true, y_pred, sample_weight)
        b, l = y_pred.shape
        scale = self._scale(b * l)
        item_w, list_w = self._weights_args(sample_weight, b, l, y_pred.device)
        list_w = _const_vector(b, scale, y_pred.device) if list_w is None else list_w * scale
        if not _LOSS_SUM_ALL:
            loss, _, _, dlogits = _ops.pointwise_loss(self._loss._fused_kind, y_pred.detach(), y_true, mask, item_w,
                                                      list_w, self._loss._temperature, True)
            return _ops.list_dot(loss), dlogits
        _, _, _, dlogits, total = _ops.pointwise_loss(self._loss._fused_kind, y_pred.detach(), y_true, mask, item_w,
                                                      list_w, self._loss._temperature, True, want_sum=True)
        return total, dlogits


@utils.register_keras_serializable()
class SigmoidCrossEntropyLoss(_PointwiseLoss):
    """keras/losses.py:1493-1556."""

    def __init__(self, reduction=Reduction.AUTO, name=None, ragged=False):
        super().__init__(reduction, name, ragged)
        self._loss = losses_impl.SigmoidCrossEntropyLoss(
            name='{}_impl'.format(name) if name else None, ragged=ragged)


@utils.register_keras_serializable()
class MeanSquaredLoss(_PointwiseLoss):
    """keras/losses.py:1550-1601."""

    def __init__(self, reduction=Reduction.AUTO, name=None, ragged=False):
        super().__init__(reduction, name, ragged)
        self._loss = losses_impl.MeanSquaredLoss(name='{}_impl'.format(name) if name else None, ragged=ragged)
