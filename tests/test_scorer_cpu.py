"""Scorer pieces: groupwise gather/score/scatter-average
against the reference's known answers and the oracle (index plumbing is plain
torch indexing, so these run on CPU too); the GPU test covers the device path."""
import math

import pytest
import torch

from oracle import tfr_ref as R
from ranking_amd import model as M
import ranking_amd as ra


def _dummy_score_fn(group_size):
    # model_test.py:232-239: context + example, plus the number of rows scored.
    def fn(ctx, ex):
        logits = ctx['context'].unsqueeze(1) + ex['example_f1']
        logits = logits.reshape(-1, group_size)
        return logits + float(logits.shape[0])
    return fn


def test_rolling_window_indices():   # model_test.py:52-73
    idx, mask = M._rolling_window_indices(3, 2, [3, 2])
    assert idx.tolist() == [[[0, 1], [1, 2], [2, 0]], [[0, 1], [1, 0], [0, 1]]]
    assert mask.tolist() == [[True, True, True], [True, True, False]]
    idx, mask = M._rolling_window_indices(3, 2, [0])
    assert idx.tolist() == [[[0, 0], [0, 0], [0, 0]]] and mask.tolist() == [[False, False, False]]
    idx, mask = M._rolling_window_indices(2, 3, [2])
    assert idx.tolist() == [[[0, 1, 0], [1, 0, 1]]] and mask.tolist() == [[True, True]]


def test_form_group_indices_no_shuffle():   # model_test.py:96-112
    is_valid = torch.tensor([[True, True, True], [True, True, False]])
    idx, mask = M._form_group_indices_nd(is_valid, 2, shuffle=False)
    assert idx.tolist() == [[[0, 1], [1, 2], [2, 0]], [[0, 1], [1, 0], [0, 1]]]
    assert mask.tolist() == [[True, True, True], [True, True, False]]
    oidx, omask = R.form_group_indices(is_valid, 2)
    assert oidx.tolist() == idx.tolist() and omask.tolist() == mask.tolist()


def test_scatter_gather_indices_reference_literals():   # model_test.py:156-189 (column of the reference's nd index)
    idx, mask = M._form_group_indices_nd(torch.tensor([[True, True, False]]), 1, shuffle=False)    # group_size 1
    assert idx.tolist() == [[[0], [1], [0]]] and mask.tolist() == [[True, True, False]]
    idx, mask = M._form_group_indices_nd(torch.tensor([[True, True, True]]), 2, shuffle=False)     # PREDICT: no shuffle
    assert idx.tolist() == [[[0, 1], [1, 2], [2, 0]]] and mask.tolist() == [[True, True, True]]
    oidx, omask = R.form_group_indices(torch.tensor([[True, True, False]]), 1)
    assert oidx.tolist() == [[[0], [1], [0]]] and omask.tolist() == [[True, True, False]]
    # two shuffles of a list with two valid items (:191-221): whatever the shuffle draws, every round forms the groups
    # {[0, 1], [1, 0], [x, y]} and masks the third one
    torch.manual_seed(2)
    for _ in range(4):
        idx, mask = M._form_group_indices_nd(torch.tensor([[True, True, False]]), 2, shuffle=True)
        assert sorted(idx[0, :2].tolist()) == [[0, 1], [1, 0]] and mask.tolist() == [[True, True, False]]


@pytest.mark.parametrize('training', [True, False])
def test_compute_logits_known_answers(training):   # model_test.py:223-277
    gs = 2
    ctx = {'context': torch.tensor([[1.]])}
    scorer = M.GroupwiseScorer(_dummy_score_fn(gs), gs)
    scorer.train(training)
    ex = {'example_f1': torch.tensor([[[1.], [2.], [3.]]])}
    is_valid = torch.tensor([[True, True, False]])
    assert scorer(ctx, ex, is_valid).tolist() == [[5., 6., 0.]]           # shuffle-invariant
    scorer2 = M.GroupwiseScorer(_dummy_score_fn(gs), gs, num_shuffles=2)
    scorer2.train(training)
    assert scorer2(ctx, ex, is_valid).tolist() == [[8., 9., 0.]]
    ex = {'example_f1': torch.tensor([[[1.], [2.], [0.]]])}
    assert scorer2(ctx, ex, torch.tensor([[True, True, True]])).tolist() == [[8., 9., 7.]]


def test_groupwise_matches_oracle():
    g = torch.Generator().manual_seed(5)
    b, l, f, gs = 4, 9, 6, 3
    x = torch.randn(b, l, f, generator=g)
    is_valid = torch.rand(b, l, generator=g) > 0.3
    is_valid[0] = False                       # a list without valid items
    w = torch.randn(gs * f, gs, generator=g)

    def torch_fn(ctx, ex):
        return ex['x'].reshape(ex['x'].shape[0], -1) @ w

    got = M.GroupwiseScorer(torch_fn, gs)({}, {'x': x}, is_valid, shuffle=False)
    want = R.groupwise_logits(lambda t: t.reshape(t.shape[0], -1) @ w, x, is_valid, gs)
    assert torch.allclose(got, want, atol=1e-6)


def test_dnn_scorer_matches_oracle_tower():
    g = torch.Generator().manual_seed(6)
    b, l, f = 3, 5, 7
    x = torch.randn(b, l, f, generator=g)
    c = torch.randn(b, 2, generator=g)
    mask = torch.tensor([[True, True, True, False, False], [True] * 5, [True, False, True, False, True]])
    scorer = ra.keras.model.DNNScorer(input_dim=f + 2, hidden_layer_dims=[8, 4], output_units=1,
                                      activation=torch.relu, use_batch_norm=False, dropout=0.)
    got = scorer({'ctx': c}, {'ex': x}, mask)
    lin = [m for m in scorer._tower if isinstance(m, torch.nn.Linear)]
    fc, fe = R.flatten_list(c, x, mask)
    flat = R.dnn_tower(torch.cat([fc, fe], dim=1), [m.weight.t() for m in lin], [m.bias for m in lin])
    want = R.restore_list(flat, mask)
    assert torch.allclose(got, want, atol=1e-5)
    assert got[0, 3].item() == pytest.approx(math.log(1e-10))


@pytest.mark.gpu
def test_scorer_on_device_bf16_close_to_fp32():
    dev = 'cuda'
    g = torch.Generator().manual_seed(7)
    b, l, f = 64, 50, 136
    x = torch.rand(b, l, f, generator=g) * 2 - 1
    mask = torch.ones(b, l, dtype=torch.bool)
    torch.manual_seed(0)
    s32 = ra.keras.model.DNNScorer(input_dim=f, hidden_layer_dims=[512, 512, 512], output_units=1,
                                   activation=torch.relu, use_batch_norm=False, dropout=0.).to(dev)
    torch.manual_seed(0)
    s16 = ra.keras.model.DNNScorer(input_dim=f, hidden_layer_dims=[512, 512, 512], output_units=1,
                                   activation=torch.relu, use_batch_norm=False, dropout=0.,
                                   compute_dtype=torch.bfloat16).to(dev)
    lin = [m for m in s32._tower if isinstance(m, torch.nn.Linear)]
    from ranking_amd.tower import FusedTower
    assert isinstance(s16._tower, FusedTower)          # bf16 -> the fused MFMA tower (csrc/tower.hip)
    with torch.no_grad():
        for i, m in enumerate(lin[:-1]):
            s16._tower.weights[i].copy_(m.weight); s16._tower.biases[i].copy_(m.bias)
        s16._tower.out_weight.copy_(lin[-1].weight); s16._tower.out_bias.copy_(lin[-1].bias)
    a = s32({}, {'x': x.to(dev)}, mask.to(dev))
    bb = s16({}, {'x': x.to(dev)}, mask.to(dev))
    want = R.dnn_tower(x.reshape(b * l, f), [m.weight.t().cpu() for m in lin],
                       [m.bias.cpu() for m in lin]).reshape(b, l)
    assert torch.allclose(a.cpu(), want, atol=1e-4, rtol=1e-4)                 # fp32 path vs oracle
    assert (bb.cpu() - want).abs().max().item() < 3e-2 * max(1.0, want.abs().max().item())   # bf16 MFMA path
    loss = ra.keras.losses.SoftmaxLoss()
    labels = (torch.rand(b, l, generator=g) > 0.8).float().to(dev)
    out = loss(labels, bb)
    out.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in s16.parameters())


def test_create_tower_routes_the_keras_options_to_the_fused_tower():
    """keras/layers.py:26-77: activation (any Keras activation), input_batch_norm, use_batch_norm, dropout.  The bf16
    tower takes the options its kernels know (module construction needs no GPU); everything else gets the torch-op
    tower."""
    from ranking_amd.keras.layers import create_tower
    from ranking_amd.tower import FusedTower, _act_code
    for act, want in ((torch.relu, 'relu'), ('tanh', 'tanh'), (torch.sigmoid, 'sigmoid'), (torch.nn.functional.elu, 'elu'),
                      (torch.nn.Softplus(), 'softplus'), (torch.nn.SiLU(), 'swish'), ('swish', 'swish'), (None, None)):
        assert _act_code(act) == want
        t = create_tower([64, 32], 1, activation=act, input_dim=24, compute_dtype=torch.bfloat16)
        assert isinstance(t, FusedTower) and t.activation == want
    with pytest.raises(ValueError):
        _act_code(torch.nn.functional.gelu)
    t = create_tower([64, 32], 1, activation=torch.nn.functional.gelu, input_dim=24, compute_dtype=torch.bfloat16)
    assert not isinstance(t, FusedTower)                       # an activation the kernels do not know
    t = create_tower([64, 32], 1, activation=torch.relu, input_batch_norm=True, input_dim=24, compute_dtype=torch.bfloat16)
    assert isinstance(t, FusedTower) and t.input_batch_norm
    names = [n for n, _ in t.named_parameters()]
    assert 'gamma_in' in names and 'beta_in' in names and t.moving_var_in.shape == (24,)
    assert not isinstance(create_tower([60, 32], 1, activation=torch.relu, input_dim=24, compute_dtype=torch.bfloat16),
                          FusedTower)                        # hidden width not a multiple of 8
    assert not isinstance(create_tower([64, 32], 1, activation=torch.relu, input_dim=24, compute_dtype=torch.float32),
                          FusedTower)                        # fp32 compute


def test_dnn_scorer_concatenates_many_example_features_once():
    """No context and several example features: DNNScorer concatenates them once (sorted names, keras/model.py:803-813)
    and flattens the result -- the same logits as flatten-then-concatenate (the oracle's order), including circular
    padding of ragged lists and a [B, L] feature without a trailing axis."""
    g = torch.Generator().manual_seed(9)
    b, l = 4, 6
    feats = {'10': torch.randn(b, l, 1, generator=g), '2': torch.randn(b, l, 3, generator=g), 'a': torch.randn(b, l, generator=g)}
    mask = torch.rand(b, l, generator=g) > 0.35
    mask[0] = True
    scorer = ra.keras.model.DNNScorer(input_dim=5, hidden_layer_dims=[7], output_units=1, activation=torch.tanh,
                                      use_batch_norm=False, dropout=0.)
    got = scorer({}, feats, mask)
    lin = [m for m in scorer._tower if isinstance(m, torch.nn.Linear)]
    cols = [feats[k].reshape(b, l, -1) for k in sorted(feats)]                  # '10' < '2' < 'a'
    _, fe = R.flatten_list(torch.zeros(b, 0), torch.cat(cols, dim=2), mask)
    flat = R.dnn_tower(fe, [m.weight.t() for m in lin], [m.bias for m in lin], activation=torch.tanh)
    assert torch.allclose(got, R.restore_list(flat, mask), atol=1e-6)
    # and equal to scoring the features one by one through the generic path
    want = ra.keras.model.UnivariateScorer.forward(scorer, {}, feats, mask)
    assert torch.allclose(got, want, atol=1e-6)
