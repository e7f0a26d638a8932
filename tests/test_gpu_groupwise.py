"""GPU parity of the groupwise multi-item scorer (SURVEY.md 8a row a22, config 5 of BASELINE.json) through the C ABI:
`tfr_group_indices_i32`, `tfr_group_gather_cast_f32_bf16`, `tfr_group_scatter_avg_f32` (+ backward) and
`ranking_amd.model.GroupwiseScorer` on the device, against the oracle (oracle/tfr_ref.py: form_group_indices*,
groupwise_logits) and the reference's known answers (python/model_test.py:52-112, 223-277).

Bars: indices and counts exact; gathered bf16 values exact (RNE of the fp32 source); scatter-average exact against the
oracle's sequential scatter (same entry order, fp32); the fused bf16 tower path within the bf16 tolerance of
tests/test_gpu_tower.py against an fp32 replica of the same tower."""
import pytest

from tests.margins import record_margin
import torch

from oracle import tfr_ref as R
from tests.common import make_batch

pytestmark = pytest.mark.gpu
PINNED_BAR = 5e-2     # forward pinned to the kernel's: backward arithmetic only (bf16 dz with stochastic rounding, 3200 randomly signed rows per entry): measured 2.9e-2 of max|dW| element-wise, 5.6e-3 in norm (bar 1.5e-2)
ELEM_BAR = 2e-1       # (round 6: no longer asserted) measured 8.9e-2 (layer 0) / 1.33e-1 (layer 1) without BatchNorm: a randomly signed upstream makes dW a sum of 3200 cancelling terms; the trajectory test (test_gpu_e2e_parity.py) is the meaningful end-to-end bound
DEV = 'cuda'


def G():
    from ranking_amd import _group_ops
    return _group_ops


def M():
    from ranking_amd import model
    return model


def _valid(B, L, seed, p=0.7):
    g = torch.Generator().manual_seed(seed)
    v = torch.rand(B, L, generator=g) < p
    if B > 2:
        v[0] = False                 # a list without valid items
        v[1] = True                  # a full list
    if B > 3:
        v[2] = False
        v[2, L // 2] = True          # exactly one valid item
    return v


@pytest.mark.parametrize('B,L', [(1, 1), (5, 3), (9, 50), (7, 64), (6, 65), (5, 200), (3, 1000), (3, 2049), (2, 4100)])   # > 2048: fewer list-waves per workgroup
@pytest.mark.parametrize('gs', [1, 2, 3, 5])
def test_group_indices_no_shuffle(B, L, gs):
    """model_test.py:52-73, 96-112 semantics at scale: valid-first index order, rolling windows mod n_valid."""
    v = _valid(B, L, seed=L + gs)
    idx, mask = G().group_indices(v.to(DEV), gs)
    want_idx, want_mask = R.form_group_indices(v, gs)
    assert idx.dtype == torch.int32 and tuple(idx.shape) == (B, L, gs)
    assert torch.equal(mask.cpu(), want_mask)
    # only the groups the mask keeps are consumed downstream, but the reference defines all of them: compare all
    assert torch.equal(idx.cpu().long(), want_idx)


def test_group_indices_reference_literals():
    """model_test.py:96-112: is_valid [[T,T,T],[T,T,F]], group_size 2."""
    v = torch.tensor([[True, True, True], [True, True, False]])
    idx, mask = G().group_indices(v.to(DEV), 2)
    assert idx.cpu().tolist() == [[[0, 1], [1, 2], [2, 0]], [[0, 1], [1, 0], [0, 1]]]
    assert mask.cpu().tolist() == [[True, True, True], [True, True, False]]
    idx, mask = M()._form_group_indices_nd(v.to(DEV), 2, shuffle=False)          # the product entry point
    assert idx.cpu().tolist() == [[[0, 1], [1, 2], [2, 0]], [[0, 1], [1, 0], [0, 1]]]


@pytest.mark.parametrize('B,L,gs', [(9, 50, 2), (6, 65, 3), (4, 200, 2), (3, 777, 4), (2, 3000, 2)])
def test_group_indices_with_shuffle_keys(B, L, gs):
    """utils.py:203-230 with shuffle: stable descending order of the draws, invalid entries last; tied draws keep
    index order."""
    v = _valid(B, L, seed=3 * L)
    g = torch.Generator().manual_seed(L)
    keys = torch.rand(B, L, generator=g)
    keys[:, 1::7] = keys[:, 0:1]                                   # ties
    idx, mask = G().group_indices(v.to(DEV), gs, keys.to(DEV))
    want_idx, want_mask = R.form_group_indices_with_keys(v, gs, keys)
    assert torch.equal(mask.cpu(), want_mask) and torch.equal(idx.cpu().long(), want_idx)
    # every valid item is the FIRST member of exactly one valid group (a permutation of the valid set)
    for b in range(B):
        first = idx[b, :, 0].cpu()[want_mask[b]]
        assert sorted(first.tolist()) == torch.nonzero(v[b]).flatten().tolist()


def test_seeded_device_shuffle_advances_per_step():
    """the train-mode shuffle must differ from step to step (an op seed owns a stream) yet be reproducible
    after `set_random_seed`."""
    from ranking_amd import utils
    v = torch.ones(4, 30, dtype=torch.bool, device=DEV)
    sc = M().GroupwiseScorer(lambda c, e: None, 2)
    sc.train()
    utils.set_random_seed(5)
    a = [sc.group_indices(v)[0].cpu() for _ in range(3)]
    utils.set_random_seed(5)
    b = [sc.group_indices(v)[0].cpu() for _ in range(3)]
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    assert not torch.equal(a[0], a[1]) and not torch.equal(a[1], a[2])


@pytest.mark.parametrize('B,L,F,gs,width', [(3, 7, 136, 2, None), (3, 7, 136, 2, 320), (4, 50, 5, 3, None),
                                            (2, 9, 8, 1, 64), (5, 33, 24, 4, 128)])
def test_group_gather_cast(B, L, F, gs, width):
    g = torch.Generator().manual_seed(F)
    x = torch.randn(B, L, F, generator=g)
    v = _valid(B, L, seed=F)
    idx, _ = R.form_group_indices(v, gs)
    out = G().group_gather_cast(x.to(DEV), idx.to(torch.int32).to(DEV), width)
    Kp = width if width is not None else (gs * F + 7) // 8 * 8
    assert out.dtype == torch.bfloat16 and tuple(out.shape) == (B * L, Kp)
    want = torch.gather(x.unsqueeze(1).expand(B, L, L, F), 2, idx.unsqueeze(-1).expand(B, L, gs, F)).reshape(B * L, gs * F)
    assert torch.equal(out[:, :gs * F].cpu(), want.to(torch.bfloat16))
    assert bool((out[:, gs * F:] == 0).all())
    # a strided view (one feature of a wider tensor) is accepted
    wide = torch.randn(B, L, F + 8, generator=g).to(DEV)
    out2 = G().group_gather_cast(wide[:, :, :F], idx.to(torch.int32).to(DEV), width)
    want2 = torch.gather(wide[:, :, :F].cpu().unsqueeze(1).expand(B, L, L, F), 2,
                         idx.unsqueeze(-1).expand(B, L, gs, F)).reshape(B * L, gs * F)
    assert torch.equal(out2[:, :gs * F].cpu(), want2.to(torch.bfloat16))


def _scatter_oracle(scores, idx, mask, L):
    b, g, gs = idx.shape
    sm = mask.unsqueeze(2).expand(b, g, gs)
    counts = torch.zeros(b, L).scatter_add_(1, idx.reshape(b, -1), sm.reshape(b, -1).float())
    s = torch.where(sm, scores.reshape(b, g, gs), torch.zeros(b, g, gs))
    logits = torch.zeros(b, L).scatter_add(1, idx.reshape(b, -1), s.reshape(b, -1))
    return R._safe_div(logits, counts), counts


@pytest.mark.parametrize('B,L,gs,shuffles', [(1, 1, 1, 1), (6, 3, 2, 1), (9, 50, 2, 1), (9, 50, 2, 3), (5, 64, 3, 1),
                                             (5, 65, 3, 2), (4, 200, 2, 1), (3, 1000, 2, 1), (2, 1100, 3, 1)])
def test_group_scatter_avg_forward_and_backward(B, L, gs, shuffles):
    v = _valid(B, L, seed=L + 11 * gs)
    gen = torch.Generator().manual_seed(L)
    parts = [R.form_group_indices_with_keys(v, gs, torch.rand(B, L, generator=gen)) for _ in range(shuffles)]
    idx = torch.cat([p[0] for p in parts], dim=1)
    mask = torch.cat([p[1] for p in parts], dim=1)
    Gn = idx.shape[1]
    scores = torch.randn(B * Gn, gs, generator=gen)
    want, want_counts = _scatter_oracle(scores, idx, mask, L)
    logits, counts = G().group_scatter_avg(scores.to(DEV), idx.to(torch.int32).to(DEV), mask.to(DEV), L)
    assert torch.equal(counts.cpu(), want_counts)
    assert torch.equal(logits.cpu(), want)                       # same entry order, fp32, correctly rounded division
    # backward against autograd through the op-by-op formulation
    s = scores.clone().requires_grad_(True)
    up = torch.randn(B, L, generator=gen)
    (_scatter_oracle(s, idx, mask, L)[0] * up).sum().backward()
    sd = scores.to(DEV).requires_grad_(True)
    out = G().GroupScatterAvgFn.apply(sd, idx.to(torch.int32).to(DEV), mask.to(DEV), L)
    (out * up.to(DEV)).sum().backward()
    assert torch.equal(sd.grad.cpu(), s.grad)


def _dummy_score_fn(group_size):
    # model_test.py:232-239: context + example, plus the number of rows scored.
    def fn(ctx, ex):
        logits = ctx['context'].unsqueeze(1) + ex['example_f1']
        logits = logits.reshape(-1, group_size)
        return logits + float(logits.shape[0])
    return fn


@pytest.mark.parametrize('training', [True, False])
def test_compute_logits_known_answers_on_device(training):
    """model_test.py:223-277 on the HIP path (device index kernel, device scatter-average)."""
    t = lambda x: torch.tensor(x, device=DEV)
    gs = 2
    ctx = {'context': t([[1.]])}
    scorer = M().GroupwiseScorer(_dummy_score_fn(gs), gs)
    scorer.train(training)
    ex = {'example_f1': t([[[1.], [2.], [3.]]])}
    is_valid = t([[True, True, False]])
    assert scorer(ctx, ex, is_valid).cpu().tolist() == [[5., 6., 0.]]            # shuffle-invariant
    scorer2 = M().GroupwiseScorer(_dummy_score_fn(gs), gs, num_shuffles=2)
    scorer2.train(training)
    assert scorer2(ctx, ex, is_valid).cpu().tolist() == [[8., 9., 0.]]
    ex = {'example_f1': t([[[1.], [2.], [0.]]])}
    assert scorer2(ctx, ex, t([[True, True, True]])).cpu().tolist() == [[8., 9., 7.]]


def _tower_and_replica(gs, F, hidden, use_bn):
    import ranking_amd as ra
    torch.manual_seed(3)
    tower = ra.keras.layers.create_tower(hidden, gs, activation=torch.relu, use_batch_norm=use_bn, dropout=0.0,
                                         input_dim=gs * F, compute_dtype=torch.bfloat16).to(DEV)
    from ranking_amd.tower import FusedTower
    assert isinstance(tower, FusedTower)
    with torch.no_grad():                                       # non-trivial BatchNorm affine parameters
        for gm, bt in zip(tower.gammas, tower.betas):
            gm.uniform_(0.5, 1.5); bt.uniform_(-0.3, 0.3)
    Ws = [w.detach().cpu().t().contiguous() for w in tower.weights] + [tower.out_weight.detach().cpu().t().contiguous()]
    bs = [b.detach().cpu() for b in tower.biases] + [tower.out_bias.detach().cpu()]
    gam = [g_.detach().cpu() for g_ in tower.gammas] if use_bn else [torch.ones(h) for h in hidden]
    bet = [b.detach().cpu() for b in tower.betas] if use_bn else [torch.zeros(h) for h in hidden]
    return tower, (Ws, bs, gam, bet)


def _replica_score_fn(rep, use_bn):
    Ws, bs, gam, bet = rep
    if use_bn:
        return lambda x: R.create_tower_train(x.reshape(x.shape[0], -1), Ws, bs, gam, bet)
    return lambda x: R.dnn_tower(x.reshape(x.shape[0], -1), Ws, bs)


@pytest.mark.parametrize('shuffle', [False, True])
@pytest.mark.parametrize('use_bn', [False, True])
def test_groupwise_scorer_fused_tower_against_the_oracle(shuffle, use_bn):
    """Config 5's scorer: group_size 2, list_size 50, 136 features, fused bf16 tower 272-512-512-512-2 fed by the
    gather-cast kernel, scatter-average kernel behind it -- forward and backward against oracle.groupwise_logits
    around an fp32 replica of the same tower (the training-mode create_tower op graph when BatchNorm is on)."""
    B, L, F, gs = 64, 50, 136, 2
    labels, _ = make_batch(B, L, seed=6)
    v = labels >= 0
    x = torch.rand(B, L, F, generator=torch.Generator().manual_seed(7)) * 2 - 1
    tower, rep = _tower_and_replica(gs, F, [512, 512, 512], use_bn)
    scorer = M().GroupwiseScorer(M().FusedGroupScoreFn(tower), gs).to(DEV)
    scorer.train()
    gidx = scorer.group_indices(v.to(DEV), shuffle=None if shuffle else False)
    got = scorer({}, {'x': x.to(DEV)}, v.to(DEV), group_indices=gidx)
    cpu_idx = (gidx[0].cpu().long(), gidx[1].cpu())
    if not shuffle:
        want_idx, want_mask = R.form_group_indices(v, gs)
        assert torch.equal(cpu_idx[0], want_idx) and torch.equal(cpu_idx[1], want_mask)
    want = R.groupwise_logits(_replica_score_fn(rep, use_bn), x, v, gs, indices=cpu_idx)
    scale = max(1.0, want.abs().max().item())
    assert (got.cpu() - want).abs().max().item() < 3e-2 * scale
    assert bool((got.cpu()[~v] == 0).all())                     # no score lands on a padded item: logit 0 (:407)
    # the general path (torch gather of fp32 group features into the same tower) agrees with the fused input
    cap = {}

    def general_fn(c, e):                                       # (keeps what the pinned comparison (c) below needs)
        xin = e['x'].reshape(e['x'].shape[0], -1)
        out = tower(xin)
        out.retain_grad()
        cap.update(xin=xin.detach(), out=out, pin=(out.grad_fn.x0, list(out.grad_fn.zs)))
        return out
    general = M().GroupwiseScorer(general_fn, gs).to(DEV)
    general.train()
    got2 = general({}, {'x': x.to(DEV)}, v.to(DEV), group_indices=gidx)
    assert (got2 - got).abs().max().item() < 2e-2 * scale
    # backward: d loss / d parameters through scatter-average backward + the fused tower backward
    up = torch.randn(B, L, generator=torch.Generator().manual_seed(8))
    params = list(tower.weights) + [tower.out_weight]

    def grads_of(out):
        for p in tower.parameters():
            p.grad = None
        (out * up.to(DEV)).sum().backward()
        return [p.grad.detach().cpu().clone() for p in params]
    g_fused = grads_of(got)
    g_general = grads_of(got2)
    # (a) the fused input (gather inside the bf16 cast) feeds the tower the same bf16 matrix as the op-by-op gather:
    #     the gradients of the two paths agree far inside bf16 noise
    for i, (a_, b_) in enumerate(zip(g_fused, g_general)):
        rel = (a_ - b_).norm().item() / (b_.norm().item() + 1e-12)
        assert rel <= 2e-3, ('fused vs general', i, rel)
    # (b) against the fp32 oracle (autograd through oracle.groupwise_logits around the fp32 replica): the direction of
    #     every weight gradient is the oracle's (bf16 operands and bf16 dz between the layers leave a few per cent of
    #     noise on a randomly signed upstream; tests/test_gpu_tower.py holds the tower itself to a bf16-aware replica)
    Ws, bs, gam, bet = rep
    leaves = [w.clone().requires_grad_(True) for w in Ws]
    fn = _replica_score_fn((leaves, bs, gam, bet), use_bn)
    (R.groupwise_logits(fn, x, v, gs, indices=cpu_idx) * up).sum().backward()
    for i, a_ in enumerate(g_fused):
        b_ = leaves[i].grad.t()
        cos = (a_ * b_).sum().item() / (a_.norm().item() * b_.norm().item() + 1e-12)
        ratio = a_.norm().item() / (b_.norm().item() + 1e-12)
        assert cos >= 0.99 and 0.9 <= ratio <= 1.1, ('vs oracle', i, cos, ratio)
        # element-wise, bf16-aware: every entry of dW within ELEM_BAR * max|dW| of the fp32 oracle's.  The operands of
        # the weight-gradient GEMM (dz and the activations) are bf16 (2^-9 relative rounding each) and 3200 rows are
        # summed with a randomly signed upstream, so the error of an entry is ~ sqrt(rows) * 2^-9 * |dz| |a| -- a few
        # per cent of the LARGEST entries, not of each entry.
        # Round 6: recorded, no longer a gate -- it needed a 20 % bar (13 % used), wide enough to hide a
        # regression; direction and norm against the oracle are asserted above, the element-wise bar is (c)'s (5 %).
        err = (a_ - b_).abs().max().item() / (b_.abs().max().item() + 1e-30)
        record_margin('groupwise scorer dW element-wise / max|dW| vs fp32 oracle (bf16 operands; recorded, not a gate)', err, float('inf'))
    # (c) the tower backward alone, at the kernel's own forward point: the bf16-aware replica of
    #     tests/test_gpu_tower.py with its forward values pinned to the kernel's bf16 input and pre-activations (same ReLU
    #     gates) and the upstream gradient the kernel backward received.  What is left is backward arithmetic -- bf16 dz with
    #     stochastic rounding, MFMA summation order -- and the bar is PINNED_BAR, not the 20 % of (b), whose entries are
    #     dominated by gates that fall differently in a forward without any bf16 rounding.
    from tests.test_gpu_tower import ref_tower
    up_tower = cap['out'].grad.clone()
    for p in tower.parameters():
        p.grad = None
    ref_tower(cap['xin'].float(), tower, pin=cap['pin']).backward(up_tower)
    for i, (a_, p) in enumerate(zip(g_general, params)):
        b_ = p.grad.detach().cpu()
        err = (a_ - b_).abs().max().item() / (b_.abs().max().item() + 1e-30)
        rel = (a_ - b_).norm().item() / (b_.norm().item() + 1e-30)
        record_margin('groupwise scorer dW, forward pinned: element-wise / max|dW|', err, PINNED_BAR)
        record_margin('groupwise scorer dW, forward pinned: ||dW - dW_ref|| / ||dW_ref||', rel, 1.5e-2)
        assert err <= PINNED_BAR and rel <= 1.5e-2, ('pinned', i, err, rel)


def test_groupwise_rejects_bad_arguments():
    import ctypes
    from ranking_amd import _lib
    lib = _lib.load()
    assert lib.tfr_group_indices_i32(None, None, 1, 4, 2, None, None, None) == -1
    assert lib.tfr_group_gather_cast_f32_bf16(None, 0, None, 1, 1, 1, 1, 1, 8, None, None) == -1
    x = torch.zeros(1, 4, 8, device=DEV); idx = torch.zeros(1, 4, 2, dtype=torch.int32, device=DEV)
    out = torch.zeros(4, 16, dtype=torch.bfloat16, device=DEV)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    assert lib.tfr_group_gather_cast_f32_bf16(p(x), 8, p(idx), 1, 4, 4, 2, 8, 12, p(out), None) == -1     # Kp % 8
    assert lib.tfr_group_gather_cast_f32_bf16(p(x), 8, p(idx), 1, 4, 4, 2, 8, 8, p(out), None) == -1      # Kp < gs * F
    with pytest.raises(ValueError):
        M().GroupwiseScorer(lambda c, e: None, 0)
