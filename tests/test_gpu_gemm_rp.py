"""GPU tests of the resident-panel tower GEMM (ranking_amd/csrc/tower_gemm_rp.h, round 6) through the C ABI.

Two oracles: (1) a plain fp32 torch product of the same bf16-rounded operands (the tolerance of tests/test_gpu_tower.py);
(2) the round-5 kernels (TFR_GEMM_RP=0) on the same inputs -- both families feed every accumulator the same k order through
the same MFMA, apply the same prologue arithmetic and the same Dropout hash, so C and the written operand must be
BIT-IDENTICAL; only the BatchNorm partial sums differ (summation order inside a 64-row slab)."""
import pytest
import torch

from tests.test_gpu_tower import T, bf16_close, rnd

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.fixture
def rp_everywhere(monkeypatch):
    """the launcher only picks the resident-panel kernel from ~two tiles per CU; the tests run it from one tile"""
    monkeypatch.setenv('TFR_GEMM_RP_MIN_TILES', '1')
    monkeypatch.setenv('TFR_GEMM_RP', '1')
    monkeypatch.setenv('TFR_GEMM_RP_ROT', '0')      # every n-tile in the round-5 k order: the bit-identity below (the rotated order: test_rotated_k_order)
    yield monkeypatch


def _operands(M, N, K, seed):
    A = (rnd((M, K), seed) * 1.2).to(DEV)
    A = (A + torch.arange(M, device=DEV).unsqueeze(1) % 7 * 0.125).to(torch.bfloat16)       # row-dependent: catches swapped fragment maps
    W = (rnd((N, K), seed + 1, 0.05).to(DEV) + torch.arange(N, device=DEV).unsqueeze(1) % 5 * 0.03125).to(torch.bfloat16)
    return A, W


# one 512-row tile; several tiles per workgroup group (M / 512 = 100 -> 12-13 per XCD over 8 groups); a ragged rest that the
# older kernels finish (300 rows); N = 128 / 256 / 512 / 1024 = 1 / 2 / 4 / 8 n-tiles
@pytest.mark.parametrize('M,N', [(512, 512), (2048, 128), (51200, 512), (20480 + 300, 256), (4096 + 64, 1024)])
@pytest.mark.parametrize('pro', [0, 2])
def test_forward_against_fp32_and_against_the_round5_kernels(M, N, pro, rp_everywhere):
    t = T()
    K = 512
    A, W = _operands(M, N, K, 100 + N)
    bias = rnd((N,), 3, 0.2).to(DEV)
    sc = (rnd((K,), 4) * 0.5 + 1.0).to(DEV); sh = rnd((K,), 5, 0.3).to(DEV)
    kw = dict(prologue=pro, a_scale=sc if pro else None, a_shift=sh if pro else None, bias=bias, epilogue=t.EPI_STATS)
    C, stats = t.gemm(A, W, N, K, **kw)
    rp_everywhere.setenv('TFR_GEMM_RP', '0')
    C0, stats0 = t.gemm(A, W, N, K, **kw)
    rp_everywhere.setenv('TFR_GEMM_RP', '1')
    assert torch.equal(C.view(torch.int16), C0.view(torch.int16)), 'C differs from the round-5 kernels'
    assert stats.shape == stats0.shape == (t.stats_rows(M), 2, N)
    scale = stats0.abs().amax(dim=0, keepdim=True).clamp_min(1.0)
    assert ((stats - stats0).abs() / scale).max().item() <= 2e-6                 # same 64-row slabs, another summation order
    a = A.float()
    if pro:
        a = torch.relu(a * sc + sh).to(torch.bfloat16).float()
    want = a @ W.float().t() + bias
    bf16_close(C, want, 'C')
    s = stats.double().sum(dim=0)
    assert (s[0] - want.double().sum(dim=0)).abs().max().item() <= 2e-3 * max(1.0, want.abs().max().item()) * M ** 0.5 + 1e-2
    # plain epilogue, no bias
    C2, none = t.gemm(A, W, N, K, prologue=pro, a_scale=sc if pro else None, a_shift=sh if pro else None)
    assert none is None
    bf16_close(C2, want - bias, 'C (plain)')


@pytest.mark.parametrize('rate', [0.5, 0.25, 0.1])                  # keep-bit table / 8-bit fields / 16-bit fields
@pytest.mark.parametrize('M,N', [(1024, 512), (51200, 512), (3072, 256)])
def test_forward_dropout_and_the_written_operand(M, N, rate, rp_everywhere):
    t = T()
    K = 512
    A, W = _operands(M, N, K, 200 + N)
    sc = (rnd((K,), 6) * 0.3 + 1.0).to(DEV); sh = rnd((K,), 7, 0.3).to(DEV)
    bias = rnd((N,), 8, 0.1).to(DEV)
    d = t.Dropout.make(rate, 4321)
    kw = dict(prologue=2, a_scale=sc, a_shift=sh, bias=bias, epilogue=t.EPI_STATS, pro_dropout=d)
    a_out = torch.full((M, K), float('nan'), dtype=torch.bfloat16, device=DEV)
    C1, st1 = t.gemm(A, W, N, K, a_out=a_out, **kw)
    C2, st2 = t.gemm(A, W, N, K, **kw)
    assert torch.equal(C1.view(torch.int16), C2.view(torch.int16)) and torch.equal(st1, st2)
    rp_everywhere.setenv('TFR_GEMM_RP', '0')
    a_out0 = torch.full((M, K), float('nan'), dtype=torch.bfloat16, device=DEV)
    C0, st0 = t.gemm(A, W, N, K, a_out=a_out0, **kw)
    rp_everywhere.setenv('TFR_GEMM_RP', '1')
    assert torch.equal(a_out.view(torch.int16), a_out0.view(torch.int16)), 'written operand differs from the round-5 kernel'
    assert torch.equal(C1.view(torch.int16), C0.view(torch.int16))
    want_a = torch.relu(A.float() * sc + sh) * t.dropout_mask(d, M, K, DEV)
    bf16_close(a_out, want_a, 'a_out')
    bf16_close(C1, a_out.float() @ W.float().t() + bias, 'C')


@pytest.mark.parametrize('rate', [0.0, 0.5, 0.1])
@pytest.mark.parametrize('M,N', [(512, 512), (51200 + 300, 512), (4096, 256)])
def test_dgrad_relu_backward_epilogue(M, N, rate, rp_everywhere):
    """C = (A . B^T) [* keep mask] * 1[Zp * e_scale + e_shift > 0]; stats = per-slab (sum C, sum C * zhat)."""
    t = T()
    K = 512
    A, W = _operands(M, N, K, 300 + N)
    Zp = rnd((M, N), 42).to(DEV).to(torch.bfloat16)
    es = (rnd((N,), 43) * 0.5 + 1.0).to(DEV); eh = rnd((N,), 44, 0.3).to(DEV)
    em = rnd((N,), 45, 0.2).to(DEV); er = (rnd((N,), 46).abs() + 0.5).to(DEV)
    d = t.Dropout.make(rate, 99) if rate > 0 else None
    kw = dict(prologue=t.PRO_NONE, epilogue=t.EPI_RELU_BWD, Zp=Zp, e_scale=es, e_shift=eh, e_mean=em, e_rstd=er, epi_dropout=d)
    C, stats = t.gemm(A, W, N, K, **kw)
    rp_everywhere.setenv('TFR_GEMM_RP', '0')
    C0, stats0 = t.gemm(A, W, N, K, **kw)
    rp_everywhere.setenv('TFR_GEMM_RP', '1')
    assert torch.equal(C.view(torch.int16), C0.view(torch.int16)), 'dy differs from the round-5 kernels'
    scale = stats0.abs().amax(dim=0, keepdim=True).clamp_min(1.0)
    assert ((stats - stats0).abs() / scale).max().item() <= 4e-6
    z = Zp.float()
    want = A.float() @ W.float().t()
    if d is not None:
        want = want * t.dropout_mask(d, M, N, DEV)
    want = want * ((z * es + eh) > 0).float()
    bf16_close(C, want, 'dy')
    s = stats.sum(dim=0)
    zhat = (z - em) * er
    lim = 4e-3 * max(1.0, want.abs().max().item()) * M ** 0.5 + 1e-2
    assert (s[0] - want.sum(dim=0)).abs().max().item() <= lim
    assert (s[1] - (want * zhat).sum(dim=0)).abs().max().item() <= lim * max(1.0, zhat.abs().max().item())


def test_rotated_k_order(rp_everywhere):
    """the default: n-tile tn works through the k blocks rotated by tn * NK / tiles_n (its neighbours' L2 traffic instead of
    four simultaneous HBM misses per line) -- another fp32 summation order: C within a bf16 ulp of the unrotated kernel,
    identical for n-tile 0"""
    t = T()
    M, N, K = 4096, 512, 512
    A, W = _operands(M, N, K, 700)
    C0, _ = t.gemm(A, W, N, K)
    rp_everywhere.delenv('TFR_GEMM_RP_ROT')
    C1, _ = t.gemm(A, W, N, K)
    assert torch.equal(C0[:, :128].view(torch.int16), C1[:, :128].view(torch.int16))
    assert not torch.equal(C0.view(torch.int16), C1.view(torch.int16))
    bf16_close(C1, A.float() @ W.float().t(), 'C (rotated k order)')
    assert (C1.float() - C0.float()).abs().max().item() <= 2.0 ** -7 * C0.float().abs().max().item()


def test_strided_operands_and_row_offsets(rp_everywhere):
    """pitches larger than the logical widths (views of wider buffers), as the tower passes them"""
    t = T()
    M, N, K = 2048, 512, 512
    Abuf = torch.zeros((M, K + 64), dtype=torch.bfloat16, device=DEV)
    Wbuf = torch.zeros((N, K + 8), dtype=torch.bfloat16, device=DEV)
    A0, W0 = _operands(M, N, K, 400)
    Abuf[:, :K] = A0; Wbuf[:, :K] = W0
    out = torch.zeros((M, N + 128), dtype=torch.bfloat16, device=DEV)
    C, _ = t.gemm(Abuf[:, :K], Wbuf[:, :K], N, K, out=out[:, :N])
    bf16_close(C, A0.float() @ W0.float().t(), 'C (strided)')
    assert bool((out[:, N:] == 0).all())


# ------------------------------------------------------------------ the weight-stationary kernel (csrc/tower_gemm_bs.h)
@pytest.fixture
def bs_everywhere(monkeypatch):
    monkeypatch.setenv('TFR_GEMM_BS_MIN_TILES', '1')
    monkeypatch.setenv('TFR_GEMM_BS', '1')
    monkeypatch.setenv('TFR_GEMM_RP', '0')
    yield monkeypatch


@pytest.mark.parametrize('M,N', [(64, 256), (64 * 9, 512), (51200 + 30, 512), (4096 + 63, 1024), (64 * 257, 256)])
def test_bs_plain_against_fp32_and_the_round5_kernels(M, N, bs_everywhere):
    """plain products (with and without a bias): same k order per accumulator as the round-5 kernels -> the same bits"""
    t = T()
    K = 512
    A, W = _operands(M, N, K, 500 + N)
    bias = rnd((N,), 3, 0.2).to(DEV)
    for b in (None, bias):
        C, none = t.gemm(A, W, N, K, bias=b)
        assert none is None
        bs_everywhere.setenv('TFR_GEMM_BS', '0')
        C0, _ = t.gemm(A, W, N, K, bias=b)
        bs_everywhere.setenv('TFR_GEMM_BS', '1')
        assert torch.equal(C.view(torch.int16), C0.view(torch.int16)), 'C differs from the round-5 kernels'
        bf16_close(C, A.float() @ W.float().t() + (0 if b is None else b), 'C')


@pytest.mark.parametrize('rate', [0.0, 0.5, 0.1])
@pytest.mark.parametrize('M,N', [(64, 256), (51200 + 300, 512), (64 * 40, 1024)])
def test_bs_dgrad_relu_backward_epilogue(M, N, rate, bs_everywhere):
    t = T()
    K = 512
    A, W = _operands(M, N, K, 600 + N)
    Zp = rnd((M, N), 42).to(DEV).to(torch.bfloat16)
    es = (rnd((N,), 43) * 0.5 + 1.0).to(DEV); eh = rnd((N,), 44, 0.3).to(DEV)
    em = rnd((N,), 45, 0.2).to(DEV); er = (rnd((N,), 46).abs() + 0.5).to(DEV)
    d = t.Dropout.make(rate, 99) if rate > 0 else None
    kw = dict(prologue=t.PRO_NONE, epilogue=t.EPI_RELU_BWD, Zp=Zp, e_scale=es, e_shift=eh, e_mean=em, e_rstd=er, epi_dropout=d)
    C, stats = t.gemm(A, W, N, K, **kw)
    bs_everywhere.setenv('TFR_GEMM_BS', '0')
    C0, stats0 = t.gemm(A, W, N, K, **kw)
    bs_everywhere.setenv('TFR_GEMM_BS', '1')
    assert torch.equal(C.view(torch.int16), C0.view(torch.int16)), 'dy differs from the round-5 kernels'
    assert stats.shape == stats0.shape == (t.stats_rows(M), 2, N)
    scale = stats0.abs().amax(dim=0, keepdim=True).clamp_min(1.0)
    assert ((stats - stats0).abs() / scale).max().item() <= 4e-6
    z = Zp.float()
    want = A.float() @ W.float().t()
    if d is not None:
        want = want * t.dropout_mask(d, M, N, DEV)
    want = want * ((z * es + eh) > 0).float()
    bf16_close(C, want, 'dy')
    s = stats.sum(dim=0)
    zhat = (z - em) * er
    lim = 4e-3 * max(1.0, want.abs().max().item()) * M ** 0.5 + 1e-2
    assert (s[0] - want.sum(dim=0)).abs().max().item() <= lim
    assert (s[1] - (want * zhat).sum(dim=0)).abs().max().item() <= lim * max(1.0, zhat.abs().max().item())
