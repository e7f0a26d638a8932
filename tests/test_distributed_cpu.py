"""world_size-2 gloo tests (CPU) of the data-parallel path: list
sharding, the single flat-bucket all-reduce, global normalisers, metric sync."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ranking_amd import distributed as D


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _toy_model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(5, 8), torch.nn.ReLU(), torch.nn.Linear(8, 1))


def _toy_loss(model, feats, labels):
    """A list-wise stand-in loss with a data-dependent GLOBAL normaliser (count of
    lists with a relevant item), like SUM_BY_NONZERO_WEIGHTS (losses.py:66-67)."""
    logits = model(feats).squeeze(-1)                       # [B, L]
    w = (labels.sum(dim=1) > 0).float()
    per_list = -(torch.log_softmax(logits, dim=1) * labels).sum(dim=1)
    return (per_list * w).sum(), w.sum()


def _worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    g = torch.Generator().manual_seed(1)
    feats = torch.randn(7, 4, 5, generator=g)
    labels = (torch.rand(7, 4, generator=g) > 0.6).float()
    labels[2] = 0.0
    model = _toy_model()
    bucket = D.FlatGradBucket(model.parameters(), n_scalars=2)
    f, l = D.shard_lists([feats, labels])
    num, den = _toy_loss(model, f, l)
    bucket.zero()
    num.backward()
    loss = D.global_normalizer_step(bucket, num, den)
    # metric sync (keras/metrics.py Mean accumulators)
    from ranking_amd.keras.metrics import _RankingMetric
    m = _RankingMetric()
    m.total, m.count = torch.tensor(float(rank + 1)), torch.tensor(2.0)
    out[rank] = (loss.item(), bucket.flat[:bucket.numel].clone(), m.result().item(), f.shape[0])
    dist.destroy_process_group()


def test_dp_world2_matches_single_process():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    g = torch.Generator().manual_seed(1)
    feats = torch.randn(7, 4, 5, generator=g)
    labels = (torch.rand(7, 4, generator=g) > 0.6).float()
    labels[2] = 0.0
    model = _toy_model()
    num, den = _toy_loss(model, feats, labels)
    (num / den).backward()
    want = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    assert out[0][3] + out[1][3] == 7 and out[0][3] == 4           # remainder goes to the first ranks
    for r in range(world):
        assert abs(out[r][0] - (num / den).item()) < 1e-5
        assert torch.allclose(out[r][1], want, atol=1e-6)
        assert abs(out[r][2] - (1.0 + 2.0) / 4.0) < 1e-6           # (sum totals)/(sum counts)


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 16, 4096):
        for w in (1, 2, 3, 8):
            spans = [D.shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_flat_bucket_single_process():
    model = _toy_model()
    bucket = D.FlatGradBucket(model.parameters(), n_scalars=3)
    assert bucket.flat.numel() == sum(p.numel() for p in model.parameters()) + 3
    model(torch.ones(2, 5)).sum().backward()
    assert bucket.flat[:bucket.numel].abs().sum() > 0             # grads landed in the bucket
    s = bucket.all_reduce(torch.tensor([1., 2., 3.]))
    assert s.tolist() == [1., 2., 3.]
    with pytest.raises(ValueError):
        D.FlatGradBucket([])


def _avg_worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    g = torch.Generator().manual_seed(3)
    feats = torch.randn(8, 4, 5, generator=g)
    labels = torch.rand(8, 4, generator=g)
    model = _toy_model()
    bucket = D.FlatGradBucket(model.parameters(), n_scalars=2)
    f, l = D.shard_lists([feats, labels])
    value = -(torch.log_softmax(model(f).squeeze(-1), dim=1) * l).sum(dim=1).mean()    # Keras AUTO on the shard
    bucket.zero()
    value.backward()
    s = bucket.all_reduce(torch.stack([value.detach(), value.new_tensor(1.0)]), average=True)   # bench.py / pipeline.py
    out[rank] = ((s[0] / world).item(), s[1].item(), bucket.flat[:bucket.numel].clone())
    dist.destroy_process_group()


def test_averaged_all_reduce_of_the_training_step_world2():
    """The step of bench.py (e2e workloads) and keras/pipeline.py: every rank back-propagates the AUTO (mean) loss
    of its equal shard, ONE all_reduce(average=True) of the flat bucket carries the gradients and (loss, 1): the
    gradients are those of the global-batch mean, the scalars come back summed."""
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_avg_worker, args=(world, port, out), nprocs=world, join=True)
    g = torch.Generator().manual_seed(3)
    feats = torch.randn(8, 4, 5, generator=g)
    labels = torch.rand(8, 4, generator=g)
    model = _toy_model()
    value = -(torch.log_softmax(model(feats).squeeze(-1), dim=1) * labels).sum(dim=1).mean()
    value.backward()
    want = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    for r in range(world):
        assert abs(out[r][0] - value.item()) < 1e-6 and out[r][1] == float(world)
        assert torch.allclose(out[r][2], want, atol=1e-6)


# ---------------------------------------------------------------------------- keras/pipeline.py DP branch, 2 ranks
class _ToyLoss:
    """Torch-only stand-in with the Keras loss protocol (the product losses need a HIP device): softmax
    cross-entropy per list, AUTO (mean over lists) reduction.  No `loss_and_grad` -> the autograd branch."""

    def __call__(self, y_true, y_pred, sample_weight=None):
        return -(torch.log_softmax(y_pred, dim=1) * y_true).sum(dim=1).mean()


class _ToyMetric:
    """Torch-only stand-in with the Keras metric protocol (name, total / count Mean sums, update_state, reset_state):
    the mean top-1 label of the lists.  Per-rank values differ when the ranks validate on different data."""
    name = 'toy_metric'

    def __init__(self):
        self.total = self.count = None

    def reset_state(self):
        self.total = self.count = None

    def update_state(self, y_true, y_pred, sample_weight=None):
        top = y_true.gather(1, y_pred.argmax(dim=1, keepdim=True)).sum()
        self.total = top if self.total is None else self.total + top
        n = torch.tensor(float(y_true.shape[0]))
        self.count = n if self.count is None else self.count + n

    def result(self):
        raise AssertionError('the pipeline reduces the sums itself (one collective), result() is not its path')


class _ToyModelBuilder:
    def __init__(self, seed):
        self._seed = seed

    def build(self):
        torch.manual_seed(self._seed)                          # a DIFFERENT initialisation on every rank

        class _M(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.net = torch.nn.Sequential(torch.nn.Linear(5, 8), torch.nn.ReLU(), torch.nn.Linear(8, 1))
                self.register_buffer('steps_seen', torch.full((1,), float(torch.initial_seed())))

            def forward(self, features):
                return self.net(features['x']).squeeze(-1)
        return _M()


def _toy_batches(n_steps, batch, lo, hi):
    g = torch.Generator().manual_seed(11)
    feats = torch.randn(n_steps, batch, 4, 5, generator=g)
    labels = torch.rand(n_steps, batch, 4, generator=g)
    return [({'x': feats[s, lo:hi]}, labels[s, lo:hi]) for s in range(n_steps)]


def _run_toy_pipeline(tmp, rank, world, seed, n_steps=6, batch=8, with_metric=False):
    from ranking_amd.keras import pipeline as P
    lo, hi = D.shard_bounds(batch, rank, world)

    class _Pipe(P.ModelFitPipeline):
        def build_loss(self):
            return _ToyLoss()

        def build_metrics(self):
            return [_ToyMetric()] if with_metric else []

        def build_weighted_metrics(self):
            return []
    # rank-dependent validation data: the early-stopping / best-checkpoint decisions must still agree
    valid = _toy_batches(2, batch, 0, batch) if rank == 0 else _toy_batches(2, batch, 0, batch // 2)
    hp = P.PipelineHparams(model_dir=os.path.join(tmp, 'rank%d' % rank), num_epochs=3, steps_per_epoch=2,
                           validation_steps=2, learning_rate=0.05, loss='toy', optimizer='sgd',
                           early_stopping_patience=2, automatic_reduce_lr=True,
                           **(dict(best_exporter_metric='toy_metric', best_exporter_metric_higher_better=True)
                              if with_metric else {}))
    pipe = _Pipe(_ToyModelBuilder(seed), P.NullDatasetBuilder(iter(_toy_batches(n_steps, batch, lo, hi)), valid), hp,
                 device=torch.device('cpu'))
    history = pipe.train_and_validate()
    flat = torch.cat([p.detach().reshape(-1) for p in pipe.model.parameters()])
    return history, flat, float(pipe.model.steps_seen)


def _pipeline_worker(rank, world, port, tmp, out, with_metric=False):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    history, flat, buf = _run_toy_pipeline(tmp, rank, world, seed=100 + rank, with_metric=with_metric)
    out[rank] = (history, flat, buf)
    dist.destroy_process_group()


def test_model_fit_pipeline_data_parallel_world2(tmp_path):
    """`ModelFitPipeline.train_and_validate` with 2 ranks (keras/pipeline.py:605-632 under a tf.distribute strategy):
    replicas are initialised differently, rank 0's parameters AND buffers are broadcast, each rank trains on its half
    of every batch, one all-reduce per step; the result equals the single-process run on the whole batches from rank
    0's initialisation, and both ranks record the same (globally reduced) validation history."""
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_pipeline_worker, args=(world, port, str(tmp_path), out), nprocs=world, join=True)
    want_hist, want_flat, want_buf = _run_toy_pipeline(str(tmp_path / 'single'), 0, 1, seed=100)
    for r in range(world):
        hist, flat, buf = out[r]
        assert torch.allclose(flat, want_flat, atol=1e-6), (flat - want_flat).abs().max()
        assert buf == want_buf                                   # buffers are broadcast too
        assert hist['loss'] == pytest.approx(want_hist['loss'], abs=1e-6)
    assert out[0][0]['val_loss'] == out[1][0]['val_loss']       # the same decisions on every rank
    assert len(out[0][0]['val_loss']) == len(out[1][0]['val_loss'])


def test_model_fit_pipeline_monitors_a_metric_identically_on_every_rank(tmp_path):
    """with `best_exporter_metric` set to a metric, the monitor that drives early stopping, ReduceLROnPlateau
    and the best checkpoint must be the SAME number on every rank although the ranks validate on different data: the
    pipeline reduces the metric's Mean sums in the collective that carries the validation loss."""
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_pipeline_worker, args=(world, port, str(tmp_path), out, True), nprocs=world, join=True)
    h0, h1 = out[0][0], out[1][0]
    assert 'val_toy_metric' in h0 and len(h0['val_toy_metric']) >= 1
    assert h0['val_toy_metric'] == h1['val_toy_metric'] and h0['val_loss'] == h1['val_loss']
    assert len(h0['loss']) == len(h1['loss'])                    # both ranks stopped after the same epoch
    assert torch.equal(out[0][1], out[1][1])                     # ... with identical parameters
    # rank 0 validated 8 lists per batch, rank 1 the first 4 of them: the reduced value is the mean over all 12
    single = _run_toy_pipeline(str(tmp_path / 'single'), 0, 1, seed=100, with_metric=True)[0]
    assert h0['val_toy_metric'] != single['val_toy_metric']


def test_flat_params_sgd_equals_per_tensor_sgd():
    """FlatGradBucket(flatten_params=True): the parameters become views of one buffer (values kept, module still
    works) and `sgd_step` equals torch.optim.SGD on the separate tensors."""
    ref = _toy_model()
    flat = _toy_model()
    opt = torch.optim.SGD(ref.parameters(), lr=0.1)
    bucket = D.FlatGradBucket(flat.parameters(), n_scalars=2, flatten_params=True)
    assert all(torch.equal(a, b) for a, b in zip(ref.parameters(), flat.parameters()))
    x = torch.randn(6, 5, generator=torch.Generator().manual_seed(2))
    for _ in range(3):
        opt.zero_grad(); bucket.zero()
        ref(x).pow(2).sum().backward(); flat(x).pow(2).sum().backward()
        opt.step(); bucket.sgd_step(0.1)
    for a, b in zip(ref.parameters(), flat.parameters()):
        assert torch.allclose(a, b, atol=1e-7)
    assert all(p.data_ptr() >= bucket.flat_params.data_ptr() for p in flat.parameters())
    with pytest.raises(ValueError):
        D.FlatGradBucket(_toy_model().parameters()).sgd_step(0.1)


def test_completion_order_puts_the_early_gradients_in_one_prefix():
    """Round 6: distributed.completion_order -- the bucket order behind the overlapped exchange (SplitStep): output layer and
    hidden layers >= split first (final first in a backward), then the layers below; every parameter exactly once."""
    import torch
    from ranking_amd import distributed as D
    from ranking_amd.tower import FusedTower
    t = FusedTower(136, [64, 32, 16], output_units=1, activation=torch.relu, use_batch_norm=True)
    for split in (1, 2):
        order, early = D.completion_order(t, split)
        assert len(order) == len(list(t.parameters())) and len({id(p) for p in order}) == len(order)
        assert order[0] is t.out_weight and order[1] is t.out_bias
        upper = [t.weights[l] for l in range(2, split - 1, -1)]
        assert [p for p in order if any(p is w for w in t.weights)][:len(upper)] == upper or all(
            a is b for a, b in zip([p for p in order if any(p is w for w in t.weights)][:len(upper)], upper))
        n_early = t.out_weight.numel() + t.out_bias.numel() + sum(
            t.weights[l].numel() + t.biases[l].numel() + t.gammas[l].numel() + t.betas[l].numel() for l in range(split, 3))
        assert early == n_early
        # a bucket on this order: the early gradients are the contiguous prefix [0, early)
        b = D.FlatGradBucket(order, n_scalars=2)
        assert t.out_weight.grad.data_ptr() == b.flat.data_ptr()
        last_early = order[[i for i, p in enumerate(order) if sum(q.numel() for q in order[:i + 1]) == early][0]]
        assert last_early.grad.data_ptr() + last_early.numel() * 4 == b.flat.data_ptr() + early * 4
    import pytest
    with pytest.raises(ValueError):
        D.completion_order(t, 3)
    with pytest.raises(ValueError):
        D.completion_order(t, 0)
