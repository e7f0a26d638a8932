"""Training-loop shell: mirror of the single-task path of
``tensorflow_ranking/python/keras/pipeline.py`` -- ``PipelineHparams`` (:262-335),
``DatasetHparams`` (:338-366), ``SimpleDatasetBuilder`` (:1026-1117), ``NullDatasetBuilder``
(:827-863), ``ModelFitPipeline.train_and_validate`` (:561-650) and ``SimplePipeline`` (:659-730).

The reference hands everything to ``tf.keras.Model.compile / fit`` under a ``tf.distribute``
strategy.  Here the loop is explicit and MI355X-shaped: batches are parsed on the host by
``libtfr_io`` (``ranking_amd.data``), copied to the GPU, scored by the fused tower, the loss
kernel returns loss AND d loss / d logits in one launch (``loss_and_grad``), the scorer backward
lands in one flat gradient bucket, ONE all-reduce per step when ``torch.distributed`` is
initialised (one process per GPU), then the optimizer.  Validation = running weighted means of the
Keras metrics (NDCG@{1,5,10,all}, like ``SimplePipeline.build_metrics``) and of the loss; best
checkpoint by ``best_exporter_metric``; ReduceLROnPlateau / early stopping like
``build_callbacks`` (:472-531).  SavedModel export is out of scope: checkpoints are
``torch.save`` state dicts under ``model_dir``.
"""
from __future__ import annotations

import abc
import dataclasses
import os
from typing import Any, Callable, Dict, Iterator, List, Optional, Tuple, Union

import torch

from .. import data as data_lib
from .. import distributed as dist_lib
from . import losses as losses_lib
from . import metrics as metrics_lib
from . import model as model_lib


@dataclasses.dataclass
class PipelineHparams:
    """keras/pipeline.py:262-335 (strategy / TPU knobs dropped: one process per GPU)."""
    model_dir: str
    num_epochs: int
    steps_per_epoch: int
    validation_steps: int
    learning_rate: float
    loss: Union[str, Dict[str, str]]
    loss_reduction: str = losses_lib.Reduction.AUTO
    optimizer: str = 'adam'
    loss_weights: Optional[Union[float, Dict[str, float]]] = None
    steps_per_execution: int = 10
    automatic_reduce_lr: bool = False
    early_stopping_patience: int = 0
    early_stopping_min_delta: float = 0.0
    use_weighted_metrics: bool = False
    export_best_model: bool = False
    best_exporter_metric_higher_better: bool = False
    best_exporter_metric: str = 'loss'


@dataclasses.dataclass
class DatasetHparams:
    """keras/pipeline.py:338-366."""
    train_input_pattern: str
    valid_input_pattern: str
    train_batch_size: int
    valid_batch_size: int
    list_size: Optional[int] = None
    valid_list_size: Optional[int] = None
    dataset_reader: Any = None
    convert_labels_to_binary: bool = False


def _convert_label(label, convert_labels_to_binary=False):
    """keras/pipeline.py:1010-1023."""
    label = label.to(torch.float32)
    if label.dim() == 3:
        label = label.squeeze(2)
    if convert_labels_to_binary:
        label = torch.where(label > 0, torch.ones_like(label), label)
    return label


class AbstractDatasetBuilder(metaclass=abc.ABCMeta):
    """keras/pipeline.py:159-259."""

    @abc.abstractmethod
    def build_train_dataset(self, *arg, **kwargs):
        raise NotImplementedError('Calling an abstract method.')

    @abc.abstractmethod
    def build_valid_dataset(self, *arg, **kwargs):
        raise NotImplementedError('Calling an abstract method.')


class NullDatasetBuilder(AbstractDatasetBuilder):
    """keras/pipeline.py:827-863: wraps ready-made iterables of (features, labels[, weights])."""

    def __init__(self, train_dataset, valid_dataset, signatures=None):
        self._train_dataset = train_dataset
        self._valid_dataset = valid_dataset

    def build_train_dataset(self, *arg, **kwargs):
        return self._train_dataset

    def build_valid_dataset(self, *arg, **kwargs):
        return self._valid_dataset


class SimpleDatasetBuilder(AbstractDatasetBuilder):
    """keras/pipeline.py:1026-1117 over ELWC TFRecords (``ranking_amd.data``)."""

    def __init__(self, context_feature_spec, example_feature_spec, mask_feature_name, label_spec,
                 hparams: DatasetHparams, sample_weight_spec=None, data_format=data_lib.ELWC,
                 example_dtype=torch.float32):
        """``data_format``: any key of ``data.make_parsing_fn``.  ``example_dtype=torch.bfloat16``: the example features
        are parsed, pinned and shipped as bfloat16 (label and sample weight stay float32; ``data._parse_batch``)."""
        self._data_format = data_format
        self._example_dtype = example_dtype
        self._context_feature_spec = context_feature_spec or {}
        self._example_feature_spec = example_feature_spec
        self._mask_feature_name = mask_feature_name
        self._label_spec = label_spec
        self._sample_weight_spec = sample_weight_spec
        self._hparams = hparams

    def _features_and_labels(self, features):
        name, _ = self._label_spec
        label = _convert_label(features.pop(name), self._hparams.convert_labels_to_binary)
        if self._sample_weight_spec:
            wname, _ = self._sample_weight_spec
            weight = features.pop(wname).squeeze(2).to(torch.float32)
            return features, label, weight
        return features, label

    def _build_dataset(self, file_pattern, batch_size, list_size, randomize_input, num_epochs, shard=None):
        spec = dict(self._example_feature_spec)
        spec[self._label_spec[0]] = self._label_spec[1]
        if self._sample_weight_spec:
            spec[self._sample_weight_spec[0]] = self._sample_weight_spec[1]
        keep32 = ()
        if self._example_dtype == torch.bfloat16:
            keep32 = (self._label_spec[0],) + ((self._sample_weight_spec[0],) if self._sample_weight_spec else ())
        ds = data_lib.build_ranking_dataset(
            file_pattern, self._data_format, batch_size, self._context_feature_spec, spec, list_size=list_size,
            mask_feature_name=self._mask_feature_name, shuffle=randomize_input,
            num_epochs=num_epochs, drop_final_batch=randomize_input, shard=shard, example_dtype=self._example_dtype,
            float32_features=keep32)
        return (self._features_and_labels(f) for f in ds)

    def build_train_dataset(self):
        # data parallel: `train_batch_size` is the GLOBAL batch (as under tf.distribute); rank r trains on its
        # 1 / world of the identically shuffled stream -- the validation set stays whole on every rank
        h = self._hparams
        rank, world = dist_lib.world()
        return self._build_dataset(h.train_input_pattern, max(1, h.train_batch_size // world), h.list_size, True, None,
                                   shard=(rank, world) if world > 1 else None)

    def build_valid_dataset(self):
        h = self._hparams
        return self._build_dataset(h.valid_input_pattern, h.valid_batch_size, h.valid_list_size or h.list_size,
                                   False, None)


class SimpleModelBuilder:
    """The role of ``model.ModelBuilder`` (keras/model.py:316-399) for numeric features: example (and
    context) features are concatenated in sorted-name order (keras/model.py:803-813) and scored by a
    ``DNNScorer``; ``build()`` returns ``fn(features) -> logits [B, L]``."""

    def __init__(self, context_feature_spec, example_feature_spec, mask_feature_name, **dnn_kwargs):
        self._context_names = sorted(context_feature_spec or {})
        self._example_names = sorted(example_feature_spec)
        self._mask_feature_name = mask_feature_name
        width = sum(data_lib._spec_width(example_feature_spec[n]) for n in self._example_names)
        width += sum(data_lib._spec_width(context_feature_spec[n]) for n in self._context_names)
        self._input_dim = width
        self._dnn_kwargs = dnn_kwargs

    def build(self) -> torch.nn.Module:
        scorer = model_lib.DNNScorer(input_dim=self._input_dim, **self._dnn_kwargs)
        builder = self

        class _Model(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.scorer = scorer

            def forward(self, features):
                ctx = {n: features[n] for n in builder._context_names}
                ex = {n: features[n] for n in builder._example_names}
                return self.scorer(ctx, ex, features[builder._mask_feature_name])
        return _Model()


_OPTIMIZERS = {'adam': torch.optim.Adam, 'adagrad': torch.optim.Adagrad, 'sgd': torch.optim.SGD,
               'rmsprop': torch.optim.RMSprop}


class AbstractPipeline(metaclass=abc.ABCMeta):
    """keras/pipeline.py:32-156."""

    @abc.abstractmethod
    def build_loss(self) -> Any:
        raise NotImplementedError('Calling an abstract method.')

    @abc.abstractmethod
    def build_metrics(self) -> Any:
        raise NotImplementedError('Calling an abstract method.')

    @abc.abstractmethod
    def build_weighted_metrics(self) -> Any:
        raise NotImplementedError('Calling an abstract method.')

    @abc.abstractmethod
    def train_and_validate(self, *arg, **kwargs) -> Any:
        raise NotImplementedError('Calling an abstract method.')


class ModelFitPipeline(AbstractPipeline):
    """keras/pipeline.py:369-650."""

    def __init__(self, model_builder, dataset_builder: AbstractDatasetBuilder, hparams: PipelineHparams,
                 device: Optional[torch.device] = None):
        self._validate_parameters(model_builder, dataset_builder)
        self._model_builder = model_builder
        self._dataset_builder = dataset_builder
        self._hparams = hparams
        if hparams.optimizer not in _OPTIMIZERS:
            raise ValueError('unsupported optimizer: {}'.format(hparams.optimizer))
        self._device = device or torch.device('cuda', torch.cuda.current_device())

    def _validate_parameters(self, model_builder, dataset_builder):
        """keras/pipeline.py:440-470."""
        if not hasattr(model_builder, 'build'):
            raise ValueError('The `model_builder` cannot be empty.')
        if not isinstance(dataset_builder, AbstractDatasetBuilder):
            raise ValueError('The `dataset_builder` cannot be empty.')

    def _to_device(self, batch):
        features, rest = batch[0], batch[1:]
        features = {k: v.to(self._device, non_blocking=True) for k, v in features.items()}
        return (features,) + tuple(t.to(self._device, non_blocking=True) for t in rest)

    def train_and_validate(self, verbose=0) -> Dict[str, List[float]]:
        """keras/pipeline.py:561-650.  Returns the history (per-epoch loss / val_loss / val metrics)."""
        h = self._hparams
        model = self._model_builder.build().to(self._device)
        dist_lib.broadcast_module(model)        # replicas start from rank 0's initialisation (mirrored variables)
        loss = self.build_loss()
        metrics = self.build_weighted_metrics() if h.use_weighted_metrics else self.build_metrics()
        opt = _OPTIMIZERS[h.optimizer](model.parameters(), lr=h.learning_rate)
        sched = (torch.optim.lr_scheduler.ReduceLROnPlateau(
            opt, mode='max' if h.best_exporter_metric_higher_better else 'min', factor=0.1,
            patience=max(1, h.early_stopping_patience // 2 or 1)) if h.automatic_reduce_lr else None)
        rank, world = dist_lib.world()
        bucket = dist_lib.FlatGradBucket(model.parameters(), n_scalars=2).attach(model) if world > 1 else None
        # batches are staged on the GPU two ahead, on a copy stream (pinned buffer -> async copy -> event): the
        # host-to-device copy of the next batch overlaps this step (data.Prefetcher)
        train_it = data_lib.Prefetcher(self._dataset_builder.build_train_dataset(), buffer_size=2, device=self._device)
        history: Dict[str, List[float]] = {}
        best, since_best = None, 0
        os.makedirs(h.model_dir, exist_ok=True)
        fused = hasattr(loss, 'loss_and_grad') and h.loss_reduction != losses_lib.Reduction.NONE
        for epoch in range(h.num_epochs):
            model.train()
            total = torch.zeros((), device=self._device)
            for _ in range(h.steps_per_epoch):
                batch = next(train_it)
                features, labels = batch[0], batch[1]
                weights = batch[2] if len(batch) > 2 else None
                opt.zero_grad(set_to_none=bucket is None)
                if bucket is not None:
                    bucket.zero()
                logits = model(features)
                if fused:                       # one launch: loss AND d loss / d logits
                    try:
                        value, dlogits = loss.loss_and_grad(labels, logits.detach(), weights)
                    except NotImplementedError:  # e.g. YetiLogisticLoss, a pairwise loss with a non-fusable
                        fused = False            # lambda weight: the autograd path serves them
                    else:
                        logits.backward(dlogits)
                if not fused:
                    value = loss(labels, logits, weights)
                    value.backward()
                if bucket is not None:          # ONE all-reduce of the flat gradient bucket per step
                    s = bucket.all_reduce(torch.stack([value.detach(), value.new_tensor(1.0)]), average=True)
                    value = s[0] / world
                opt.step()
                total += value.detach()
            history.setdefault('loss', []).append(float(total) / max(1, h.steps_per_epoch))
            # ---- validation
            model.eval()
            for m in metrics:
                m.reset_state()
            vloss, vcount = torch.zeros((), device=self._device), 0
            valid_it = iter(self._dataset_builder.build_valid_dataset())
            with torch.no_grad():
                for _ in range(h.validation_steps):
                    try:
                        batch = self._to_device(next(valid_it))
                    except StopIteration:
                        break
                    features, labels = batch[0], batch[1]
                    weights = batch[2] if len(batch) > 2 else None
                    logits = model(features)
                    vloss += loss(labels, logits, weights)
                    vcount += 1
                    # (objects that differ only in their cut-off share one launch: keras/metrics.py update_metrics)
                    metrics_lib.update_metrics(metrics, labels, logits, weights if h.use_weighted_metrics else None)
            # Every rank must see the same numbers: the early-stopping / ReduceLROnPlateau / best-checkpoint decisions
            # below decide whether this rank enters the next all-reduce.  ONE collective carries the validation loss
            # and every metric's Mean sums (a metric without a batch on this rank contributes zeros); BatchNorm moving
            # statistics differ between ranks after the initial broadcast, so per-rank metric values would too.
            sums = [vloss, vcount]
            for m in metrics:
                sums += [m.total if m.total is not None else 0.0, m.count if m.count is not None else 0.0]
            if world > 1:
                sums = dist_lib.all_reduce_scalars(sums, self._device)
            sums = [float(x) for x in sums]
            history.setdefault('val_loss', []).append(sums[0] / max(1.0, sums[1]))
            for i, m in enumerate(metrics):
                t, c = sums[2 + 2 * i], sums[3 + 2 * i]
                history.setdefault('val_' + m.name, []).append(t / c if c != 0 else 0.0)
            monitor = history['val_' + h.best_exporter_metric][-1] if h.best_exporter_metric != 'loss' \
                else history['val_loss'][-1]
            if verbose and rank == 0:
                print('epoch %d: ' % (epoch + 1) + ', '.join('%s %.5f' % (k, v[-1]) for k, v in history.items()))
            improved = best is None or (
                monitor > best + h.early_stopping_min_delta if h.best_exporter_metric_higher_better
                else monitor < best - h.early_stopping_min_delta)
            if improved:
                best, since_best = monitor, 0
                if h.export_best_model and rank == 0:
                    os.makedirs(os.path.join(h.model_dir, 'best_checkpoint'), exist_ok=True)
                    torch.save(model.state_dict(), os.path.join(h.model_dir, 'best_checkpoint', 'ckpt.pt'))
            else:
                since_best += 1
            if sched is not None:
                sched.step(monitor)
            if h.early_stopping_patience and since_best >= h.early_stopping_patience:
                break
        train_it.close()
        if rank == 0:
            os.makedirs(os.path.join(h.model_dir, 'export', 'latest_model'), exist_ok=True)
            torch.save(model.state_dict(), os.path.join(h.model_dir, 'export', 'latest_model', 'model.pt'))
        self.model = model
        return history


def _get_metric(prefix, key, topn=None, **kwargs):
    """keras/pipeline.py:653-656."""
    name = '{}{}{}'.format(prefix, key, '_%s' % topn if topn else '')
    return metrics_lib.get(key, name=name, topn=topn, **kwargs)


class SimplePipeline(ModelFitPipeline):
    """keras/pipeline.py:659-730."""

    def build_loss(self):
        if not isinstance(self._hparams.loss, str):
            raise TypeError('In the simple pipeline, losses are expected to be specified in a str.')
        return losses_lib.get(loss=self._hparams.loss, reduction=self._hparams.loss_reduction)

    def build_metrics(self):
        return [_get_metric('metric/', metrics_lib.RankingMetricKey.NDCG, topn=topn) for topn in [1, 5, 10, None]]

    def build_weighted_metrics(self):
        return [_get_metric('weighted_metric/', metrics_lib.RankingMetricKey.NDCG, topn=topn)
                for topn in [1, 5, 10, None]]
