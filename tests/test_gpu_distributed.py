"""The N > 1 path on real devices (VERDICT r3 "Next round" #6; the reference's multi-device entry is
python/keras/strategy_utils.py:45-116).  Runs only where two GPUs are visible: two ranks on the `nccl` backend (= RCCL),
each scoring its shard of one global batch with the fused bf16 tower (per-replica BatchNorm statistics, like
`tf.distribute`), ONE all-reduce of the flat gradient bucket -- against the same two shards run one after the other in a
single process and averaged.  On a 1-GPU box the test is skipped (and says so); the world-2 `gloo` tests of
tests/test_distributed_cpu.py cover the host logic there."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _batch(B, L, F):
    from ranking_amd.synthetic import make_batch
    labels, _ = make_batch(B, L, seed=11)
    feats = torch.rand((B, L, F), generator=torch.Generator().manual_seed(12)) * 2 - 1
    return labels, feats


def _shard_step(labels, feats, dev, world_reduce=None):
    """One config-2 training step (scaled) on one shard; returns (loss value, flat gradients, BN moving means)."""
    import ranking_amd as ra
    from ranking_amd import distributed as D
    torch.manual_seed(0)
    scorer = ra.keras.model.DNNScorer(input_dim=feats.shape[2], hidden_layer_dims=[512, 512, 512], output_units=1,
                                      activation=torch.relu, use_batch_norm=True, dropout=0.0,
                                      compute_dtype=torch.bfloat16).to(dev)
    scorer.train()
    D.broadcast_module(scorer)
    bucket = D.FlatGradBucket(scorer.parameters(), n_scalars=2, flatten_params=True)
    bucket.attach(scorer)
    labels, feats = labels.to(dev), feats.to(dev)
    loss = ra.keras.losses.SoftmaxLoss()
    bucket.zero()
    logits = scorer({}, {'x': feats}, labels >= 0)
    value, dlogits = loss.loss_and_grad(labels, logits.detach())
    logits.backward(dlogits)
    s = torch.stack([value, value.new_tensor(1.0)])
    if world_reduce is not None:
        s = bucket.all_reduce(s, average=True)
        value = s[0] / world_reduce
    mm = torch.cat([b.reshape(-1).float() for n, b in scorer.named_buffers() if 'moving_mean' in n])
    return value.detach().cpu(), bucket.flat[:bucket.numel].detach().cpu().clone(), mm.cpu()


def _worker(rank, world, port, out):
    import torch.distributed as dist
    from ranking_amd import distributed as D
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(rank)
    dev = torch.device('cuda', rank)
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    labels, feats = _batch(128, 100, 136)
    lb, ft = D.shard_lists([labels, feats])
    out[rank] = _shard_step(lb, ft, dev, world_reduce=world)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_rccl_match_the_two_shards_run_in_one_process():
    if torch.cuda.device_count() < 2:
        pytest.skip('needs 2 visible GPUs (%d here): the nccl path runs in the driver\'s multi-GPU bench' %
                    torch.cuda.device_count())
    import torch.multiprocessing as mp
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    labels, feats = _batch(128, 100, 136)
    dev = torch.device('cuda', 0)
    v0, g0, m0 = _shard_step(labels[:64], feats[:64], dev)
    v1, g1, m1 = _shard_step(labels[64:], feats[64:], dev)
    want_v, want_g = (v0 + v1) / 2, (g0 + g1) / 2
    for r, (mm, ) in ((0, (m0,)), (1, (m1,))):
        v, g, m = out[r]
        assert abs(v.item() - want_v.item()) <= 1e-6 * max(1.0, abs(want_v.item())), (r, v, want_v)
        assert torch.allclose(g, want_g, rtol=1e-5, atol=1e-7 * want_g.abs().max().item()), r
        assert torch.equal(m, mm), r                               # per-replica BatchNorm statistics: the shard's own
