"""fp32 Dense on the matrix cores (csrc/gemm_f32.hip, VERDICT r2 missing #1 / next #6): the reference's tower is fp32
(keras/layers.py:26-77); `create_tower(compute_dtype=float32)` now runs hand-written MFMA kernels forward and backward.

Bars: every product against the same product evaluated in float64, error relative to sum_k |a_k| |b_k| (the scale
fp32 accumulation works at) <= 1e-6 -- an fp32 fma chain over K <= 4096 sits at 1e-7 (measured <= 3.4e-7) -- and the
whole fp32 tower, logits and every gradient, within 1e-5 of the fp32 torch-op replica (north_star's floating-point
bar; measured <= 1.3e-6 with a smooth activation; ReLU: see the kink note below).
"""
import pytest
import torch

from tests.margins import record_margin

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def T():
    from ranking_amd import _tower_ops
    return _tower_ops


def _rel(got, want64, scale64):
    return ((got.double() - want64).abs() / scale64.clamp(min=1e-30)).max().item()


SHAPES = [(1000, 512, 136), (333, 37, 19), (4096, 512, 512), (130, 1, 512), (1, 7, 3), (257, 129, 17), (5000, 3, 250),
          (128, 128, 16), (129, 130, 4100), (70000, 2, 300), (4097, 4, 33), (300000, 1, 64)]


@pytest.mark.parametrize('M,N,K', SHAPES)
def test_dense_forward_dgrad_wgrad_against_float64(M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    x = torch.randn(M, K, generator=g).to(DEV)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    dy = torch.randn(M, N, generator=g).to(DEV)
    x64, w64, dy64 = x.double(), w.double(), dy.double()
    e_f = _rel(T().dense_f32(x, w, b), x64 @ w64.t() + b.double(), x64.abs() @ w64.abs().t() + b.double().abs())
    e_d = _rel(T().dense_f32_dgrad(dy, w), dy64 @ w64, dy64.abs() @ w64.abs())
    e_w = _rel(T().dense_f32_wgrad(dy, x), dy64.t() @ x64, dy64.abs().t() @ x64.abs())
    e_b = _rel(T().colsum_f32(dy), dy64.sum(0), dy64.abs().sum(0))
    for name, e in (('forward', e_f), ('dgrad', e_d), ('wgrad', e_w), ('bias grad', e_b)):
        record_margin('fp32 Dense %s %dx%dx%d vs fp64 / sum|a||b|' % (name, M, N, K), e, 1e-6)
    assert max(e_f, e_d, e_w, e_b) <= 1e-6, (e_f, e_d, e_w, e_b)


def test_unaligned_pitches_and_views():
    """Row pitches that are not multiples of four floats and pointers off a 16-byte boundary take the scalar loads."""
    g = torch.Generator().manual_seed(5)
    big = torch.randn(301, 150, generator=g).to(DEV)
    x = big[:, 1:138]                                   # pitch 150, pointer + 4 bytes, K = 137
    wb = torch.randn(65, 139, generator=g).to(DEV)
    w = wb[:, 2:139]                                    # pitch 139
    y = T().dense_f32(x, w, None)
    want = x.double() @ w.double().t()
    assert _rel(y, want, x.double().abs() @ w.double().abs().t()) <= 1e-6
    dyb = torch.randn(301, 67, generator=g).to(DEV)
    dy = dyb[:, 1:66]
    assert _rel(T().dense_f32_dgrad(dy, w), dy.double() @ w.double(), dy.double().abs() @ w.double().abs()) <= 1e-6
    assert _rel(T().dense_f32_wgrad(dy, x), dy.double().t() @ x.double(), dy.double().abs().t() @ x.double().abs()) <= 1e-6


def test_wgrad_is_bitwise_reproducible_and_empty_shapes():
    g = torch.Generator().manual_seed(9)
    dy = torch.randn(20000, 64, generator=g).to(DEV)
    x = torch.randn(20000, 136, generator=g).to(DEV)
    a, b = T().dense_f32_wgrad(dy, x), T().dense_f32_wgrad(dy, x)
    assert torch.equal(a, b)
    assert T().dense_f32(torch.empty(0, 8, device=DEV), torch.randn(4, 8, device=DEV), None).shape == (0, 4)
    y = T().dense_f32(torch.empty(5, 0, device=DEV), torch.empty(4, 0, device=DEV), torch.arange(4., device=DEV))
    assert torch.equal(y, torch.arange(4., device=DEV).expand(5, 4))


def _towers(hidden, out_units, f, use_bn, input_bn, act, dtype=torch.float32):
    import ranking_amd as ra
    from ranking_amd import scorer
    torch.manual_seed(3)
    mine = ra.keras.layers.create_tower(hidden, out_units, activation=act, input_batch_norm=input_bn, use_batch_norm=use_bn,
                                        batch_norm_moment=0.9, dropout=0.0, input_dim=f, compute_dtype=dtype).to(DEV)
    assert any(isinstance(m, (scorer.DenseF32, scorer.DenseBf16)) for m in mine)
    # the replica: the same modules with torch's own linear (the op graph the reference runs in fp32)
    import copy
    rep = copy.deepcopy(mine)
    for i, m in enumerate(rep):
        if isinstance(m, torch.nn.Linear):
            lin = torch.nn.Linear(m.in_features, m.out_features).to(DEV)
            lin.load_state_dict(m.state_dict())
            rep[i] = lin
    return mine, rep


def _run_both(mine, rep, f, out_units, rows=3000):
    g = torch.Generator().manual_seed(11)
    x = torch.randn(rows, f, generator=g).to(DEV)
    dlog = torch.randn(rows, out_units, generator=g).to(DEV)
    outs = []
    for tower in (mine, rep):
        tower.train()
        xi = x.clone().requires_grad_(True)
        y = tower(xi)
        (y * dlog).sum().backward()
        outs.append((y.detach(), xi.grad, [p.grad for p in tower.parameters()]))
    return outs


@pytest.mark.parametrize('hidden,out_units,f,use_bn,input_bn', [([512, 512, 512], 1, 136, True, False),
                                                                ([100, 37], 5, 45, True, True),
                                                                ([64], 3, 10, False, False)])
def test_fp32_tower_within_1e5_of_the_fp32_replica(hidden, out_units, f, use_bn, input_bn):
    """A smooth activation (tanh): two fp32 evaluations of the same tower stay within 1e-5 everywhere.  Parameter
    gradients are measured against the largest gradient entry of the tower: the bias of a Dense layer followed by
    BatchNorm has an analytically ZERO gradient (the batch mean is subtracted), both sides hold rounding noise there."""
    mine, rep = _towers(hidden, out_units, f, use_bn, input_bn, torch.tanh)
    (y1, dx1, g1), (y2, dx2, g2) = _run_both(mine, rep, f, out_units)
    e_y = ((y1 - y2).abs().max() / y2.abs().max().clamp(min=1.0)).item()
    e_x = ((dx1 - dx2).abs().max() / dx2.abs().max()).item()
    gmax = max(b.abs().max().item() for b in g2)
    e_g = max((a - b).abs().max().item() for a, b in zip(g1, g2)) / gmax
    record_margin('fp32 tower %s -> %d (tanh): logits vs fp32 replica' % (hidden, out_units), e_y, 1e-5)
    record_margin('fp32 tower %s -> %d (tanh): input gradient / max' % (hidden, out_units), e_x, 1e-5)
    record_margin('fp32 tower %s -> %d (tanh): parameter gradients / max' % (hidden, out_units), e_g, 1e-5)
    assert e_y <= 1e-5 and e_x <= 1e-5 and e_g <= 1e-5, (e_y, e_x, e_g)


def test_fp32_relu_tower_against_the_fp32_replica():
    """ReLU (BASELINE's activation), 136-512-512-512-1 with BatchNorm: logits within 1e-5.  Of 4.6 M pre-activations
    a handful lie within fp32 rounding of zero, where the two evaluations take different sides of the kink: the input
    gradient of THOSE rows differs by one unit's worth (measured 8e-3 of the largest entry) and, through the batch
    statistics, every other row by 1 / rows of that; one weight-gradient entry moves by that row's dy * x (measured:
    one row of 3000 flipped, parameter gradients 1.9e-3 of the largest entry).  Asserted: the median row within 1e-5
    of the largest entry, at most 0.5 % of the rows beyond 1e-4, parameter gradients within 1e-2 of the largest entry;
    the smooth-activation test above is the one that holds 1e-5 on every gradient."""
    mine, rep = _towers([512, 512, 512], 1, 136, True, False, torch.relu)
    (y1, dx1, g1), (y2, dx2, g2) = _run_both(mine, rep, 136, 1)
    e_y = ((y1 - y2).abs().max() / y2.abs().max().clamp(min=1.0)).item()
    row_err = (dx1 - dx2).abs().amax(dim=1) / dx2.abs().max()
    frac = (row_err > 1e-4).float().mean().item()
    med = row_err.median().item()
    gmax = max(b.abs().max().item() for b in g2)
    e_g = max((a - b).abs().max().item() for a, b in zip(g1, g2)) / gmax
    record_margin('fp32 tower 136-512-512-512-1 (ReLU): logits vs fp32 replica', e_y, 1e-5)
    record_margin('fp32 tower 136-512-512-512-1 (ReLU): input gradient, median row / max', med, 1e-5)
    record_margin('fp32 tower 136-512-512-512-1 (ReLU): fraction of rows beyond 1e-4 (kink flips)', frac, 5e-3)
    record_margin('fp32 tower 136-512-512-512-1 (ReLU): parameter gradients / max', e_g, 1e-2)
    assert e_y <= 1e-5 and med <= 1e-5 and frac <= 5e-3 and e_g <= 1e-2, (e_y, med, frac, e_g)


def test_bf16_dense_for_shapes_the_fused_tower_does_not_take():
    """Hidden widths that are not multiples of 8 / output_units > 4 with compute_dtype=bfloat16: operands rounded to
    bf16, fp32 products and sums -- against the same rounding followed by a float64 product."""
    from ranking_amd import scorer
    mine, _ = _towers([100, 37], 5, 45, False, False, torch.relu, dtype=torch.bfloat16)
    lin = [m for m in mine if isinstance(m, scorer.DenseBf16)]
    assert len(lin) == 3
    g = torch.Generator().manual_seed(12)
    x = torch.randn(777, 45, generator=g).to(DEV)
    y = lin[0](x)
    xr, wr = x.bfloat16().double(), lin[0].weight.detach().bfloat16().double()
    want = xr @ wr.t() + lin[0].bias.detach().double()
    assert _rel(y.detach(), want, xr.abs() @ wr.abs().t() + 1.0) <= 1e-6
    y.sum().backward()
    assert lin[0].weight.grad is not None and lin[0].weight.grad.shape == lin[0].weight.shape
