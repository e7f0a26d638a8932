"""The scorer at the BASELINE configurations' REAL sizes, and every bench.py workload, inside `-m gpu`
(VERDICT r3 "Next round" #2; weak #2: the tower tests stopped at M = 4 100 rows while bench.py runs 25 600 .. 819 200,
and the driver's bench was the first thing to exercise the full-size hipGraph step -- and faulted).

* `test_fused_tower_at_baseline_rows`: one forward + backward of `FusedTower` at M = 409 600 (config 2), 819 200
  (config 3), 512 000 (config 4) rows of 136 features and at 25 600 rows of 272 (config 5's group tower), Dropout 0 and
  the reference default 0.5 (keep masks rebuilt on the host side of the test from the counter hash), eagerly AND replayed
  from a hipGraph, against the bf16-aware fp32 replica (tests/test_gpu_tower.py: ref_tower / ref_tower_dropout) evaluated
  on the same device: logits on every row, every parameter gradient.
* `test_every_bench_workload_runs`: `bench.build_step` for every key of `bench.WORKLOADS` exactly as the driver's
  command builds it (hipGraph, reference dropout), three replays, synchronise, finite outputs.
* `test_graph_replay_survives_constant_cache_overflow`: the cached constants a captured step has read stay alive when the
  caches overflow.
"""
import pytest
import torch

from tests.margins import record_margin
from tests.test_gpu_tower import ref_tower, ref_tower_dropout, rnd

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _tower(F, O, rate, seed=0):
    from ranking_amd.tower import FusedTower
    torch.manual_seed(seed)
    t = FusedTower(F, [512, 512, 512], O, activation='relu', use_batch_norm=True, dropout=rate).to(DEV)
    with torch.no_grad():
        for p in list(t.biases) + [t.out_bias]:
            p.normal_(0, 0.1)
        for g in t.gammas:
            g.uniform_(0.5, 1.5)
        for b in t.betas:
            b.normal_(0, 0.2)
    t.train()
    return t


def _masks(tower, M, rate):
    from ranking_amd import _tower_ops as T
    return [T.dropout_mask(d, M, h, DEV) for d, h in zip(tower.dropout_structs(), tower.hidden_layer_dims)]


def _compare(tag, got, want, g_got, g_want, names):
    scale = max(1.0, want.abs().max().item())
    err = (got - want).abs().max().item() / scale
    record_margin('full-size fused tower logits vs bf16-aware fp32 replica (%s)' % tag, err, 3e-2, pin=True)
    assert torch.isfinite(got).all()
    assert err <= 3e-2, (tag, err)
    gscale = max(b.abs().max().item() for b in g_want)
    for n, a, b in zip(names, g_got, g_want):
        assert torch.isfinite(a).all(), (tag, n)
        rel = (a - b).norm().item() / (b.norm().item() + 1e-6 * b.numel() ** 0.5)
        mx = (a - b).abs().max().item() / gscale
        record_margin('full-size fused tower gradients (%s): min(rel / 3e-2, max-norm / 2e-2)' % tag,
                      min(rel / 3e-2, mx / 2e-2), 1.0, pin=True)
        assert rel <= 3e-2 or mx <= 2e-2, (tag, n, rel, mx)


@pytest.mark.parametrize('M,F,O', [(409600, 136, 1), (819200, 136, 1), (512000, 136, 1), (25600, 272, 2)])
@pytest.mark.parametrize('rate', [0.0, 0.5])
def test_fused_tower_at_baseline_rows(M, F, O, rate):
    tower = _tower(F, O, rate)
    names = [n for n, _ in tower.named_parameters()]
    x = rnd((M, F), 150).to(DEV)
    up = (rnd((M, O), 151) / M ** 0.5).to(DEV)                # d loss / d logits of a mean-like loss

    # eager
    got = tower(x)
    got.backward(up)
    g_got = [p.grad.clone() for p in tower.parameters()]
    masks = _masks(tower, M, rate) if rate > 0.0 else None
    tower.zero_grad(set_to_none=True)
    want = ref_tower_dropout(x, tower, masks) if rate > 0.0 else ref_tower(x, tower)
    want.backward(up)
    g_want = [p.grad.clone() for p in tower.parameters()]
    want = want.detach()
    tower.zero_grad(set_to_none=True)
    _compare('M=%d dropout=%g eager' % (M, rate), got.detach(), want, g_got, g_want, names)
    del got, g_got
    torch.cuda.synchronize()

    # the same step captured in a hipGraph and replayed (what bench.py and a production loop do)
    for p in tower.parameters():
        p.grad = torch.zeros_like(p)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            tower(x).backward(up)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for p in tower.parameters():
            p.grad.zero_()
        out = tower(x)
        out.backward(up)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    g_rep = [p.grad.clone() for p in tower.parameters()]
    if rate > 0.0:                                            # the replica with the masks the LAST replay drew
        masks = _masks(tower, M, rate)
        tower.zero_grad(set_to_none=True)
        want = ref_tower_dropout(x, tower, masks)
        want.backward(up)
        g_want = [p.grad.clone() for p in tower.parameters()]
        want = want.detach()
    _compare('M=%d dropout=%g hipGraph replay' % (M, rate), out.detach(), want, g_rep, g_want, names)


def test_every_bench_workload_runs():
    """Every workload of bench.py, built the way the driver's command builds it (hipGraph replay, the reference's
    Dropout 0.5 for the end-to-end ones) at its full size: three steps, a synchronise, finite results -- so that the
    driver's pytest catches a faulting workload before its bench does."""
    import bench
    for name, (B, L, _, _) in bench.WORKLOADS.items():
        if name in bench.HBM_VARIANTS:                        # the same steps on a cycled working set
            continue
        labels, logits = bench.make_inputs(B, L, seed=4, device=torch.device('cuda', 0))
        is_e2e = name.startswith('e2e_')
        info = bench.build_step(name, labels, logits, bench.REFERENCE_DROPOUT if is_e2e else 0.0, use_graph=True)
        step = info['step']
        if not is_e2e:
            step = bench.graph_of(step)
        out = None
        for _ in range(3):
            out = step()
        if info.get('kernel') is not None:
            info['kernel']()
        torch.cuda.synchronize()
        vals = out if isinstance(out, (tuple, list)) else (out,)
        flat = []
        for v in vals:
            flat += list(v) if isinstance(v, (tuple, list)) else [v]
        for v in flat:
            if torch.is_tensor(v):
                assert torch.isfinite(v.float()).all(), name
        del info, step, out
        torch.cuda.synchronize()
        torch.cuda.empty_cache()


def test_graph_replay_survives_constant_cache_overflow():
    """`_ops._table_cache` / `keras.losses._CONST_CACHE` used to clear() themselves when full; a
    captured step then replayed against freed storage.  Capture a loss step, overflow both caches, churn the
    allocator, replay, compare with the eager result."""
    import ranking_amd as ra
    from ranking_amd import _ops
    from ranking_amd.keras import losses as K
    from ranking_amd.synthetic import make_batch
    labels, logits = make_batch(256, 64, seed=21)
    labels, logits = labels.to(DEV), logits.to(DEV)
    loss = ra.keras.losses.PairwiseLogisticLoss(lambda_weight=ra.keras.losses.NDCGLambdaWeight())
    want_v, want_g = loss.loss_and_grad(labels, logits)
    want_v, want_g = want_v.clone(), want_g.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        loss.loss_and_grad(labels, logits)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        v, g = loss.loss_and_grad(labels, logits)
    pinned = _ops._table_cache.pinned() + K._CONST_CACHE.pinned()
    assert pinned >= 1                                        # the capture read cached constants: they are pinned now
    for i in range(400):                                      # overflow both caches (capacities 256 / 64)
        _ops.rank_table(lambda r, i=i: 1.0 / torch.log1p(r + i), 64, labels.device)
        K._const_vector(256, 1.0 / (i + 2), labels.device)
    assert len(_ops._table_cache) <= 256 + _ops._table_cache.pinned()
    assert len(K._CONST_CACHE) <= 64 + K._CONST_CACHE.pinned()
    torch.cuda.empty_cache()
    junk = [torch.full((1 << 16,), float('nan'), device=DEV) for _ in range(64)]     # reuse whatever was freed
    graph.replay()
    torch.cuda.synchronize()
    del junk
    assert torch.equal(v, want_v) and torch.equal(g, want_g)
