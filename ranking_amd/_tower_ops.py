"""Torch-tensor level bindings of the scorer-tower entry points of the C ABI
(include/tfr_hip.h, ranking_amd/csrc/tower.hip).  Device memory and streams only;
no CPU fallback."""
from __future__ import annotations

import ctypes
from typing import Optional

import os
import torch

from . import _lib
from . import _ops
from ._ops import _ptr, _stream, require_device

class Dropout(ctypes.Structure):
    """tfr_tower_dropout (include/tfr_hip.h): counter-based keep mask shared by forward and backward.  ``step``: an
    int32 / uint32 device tensor of one element added to the seed (times the golden ratio) by the kernels -- the
    training-step counter, in device memory so that hipGraph replays draw fresh masks."""
    _fields_ = [('seed', ctypes.c_uint32), ('threshold16', ctypes.c_uint32), ('scale', ctypes.c_float),
                ('step', ctypes.c_void_p)]

    @classmethod
    def make(cls, rate: float, seed: int, step: Optional[torch.Tensor] = None):
        thr = max(0, min(65535, int(round(float(rate) * 65536.0))))
        d = cls(int(seed) & 0xffffffff, thr, 65536.0 / (65536.0 - thr), None)
        if step is not None:
            if step.numel() != 1 or step.dtype not in (torch.int32, torch.uint32) or not step.is_cuda:
                raise ValueError('Dropout step must be a one-element int32 device tensor')
            d.step = step.data_ptr()
            d._step_tensor = step                     # keeps the storage alive as long as the struct
        return d

    def resolved(self) -> 'Dropout':
        """The struct with the step counter folded into the seed (reads the device value: tests / debugging)."""
        seed = int(self.seed)
        t = getattr(self, '_step_tensor', None)
        if t is not None:
            seed = (seed + (int(t.item()) & 0xffffffff) * 0x9E3779B9) & 0xffffffff
        return Dropout(seed, int(self.threshold16), float(self.scale), None)


def _dp(d):
    return None if d is None else ctypes.byref(d)


def dropout_field(d: 'Dropout'):
    """(lge, threshold in field units, scale) the kernels derive from the struct (csrc/tower.hip to_drop): a hash word
    serves 2^lge columns with 32 >> lge bits each -- the narrowest field (1, 2, 4 or 8 bits) that represents the rate
    exactly; a rate that is not a multiple of 1 / 256 takes 16-bit fields (the rate to 1 / 65 536, the scale following
    the threshold)."""
    t16 = min(int(d.threshold16), 65535)
    for lge in (5, 4, 3, 2):
        fb = 32 >> lge
        if t16 & ((1 << (16 - fb)) - 1) == 0:
            return lge, t16 >> (16 - fb), float(d.scale)
    return 1, t16, 65536.0 / (65536.0 - t16)


def dropout_mask(d: 'Dropout', M: int, K: int, device) -> torch.Tensor:
    """The [M, K] keep-factor matrix the kernels apply (torch restatement of drop_run; tests / debugging)."""
    lge, thr, scale = dropout_field(d)
    fb = 32 >> lge
    m = torch.arange(M, device=device, dtype=torch.int64).unsqueeze(1)
    c = torch.arange(K, device=device, dtype=torch.int64).unsqueeze(0)
    mask32 = 0xffffffff
    h = (m * 0x9E3779B1 + (c >> lge) * 0x85EBCA77 + int(d.seed)) & mask32
    h = h ^ (h >> 16); h = (h * 0x7feb352d) & mask32
    h = h ^ (h >> 15); h = (h * 0x846ca68b) & mask32
    h = h ^ (h >> 16)
    field = (h >> ((c & ((1 << lge) - 1)) * fb)) & ((1 << fb) - 1)
    return (field >= thr).to(torch.float32) * scale


PRO_NONE, PRO_AFFINE, PRO_AFFINE_RELU, PRO_AFFINE_ACT = 0, 1, 2, 3
EPI_PLAIN, EPI_STATS, EPI_RELU_BWD, EPI_ACT_BWD = 0, 1, 2, 3
# activations other than ReLU: the code rides in bits 8.. of the `prologue` / `epilogue` arguments (include/tfr_hip.h)
ACT_CODES = {'tanh': 1, 'sigmoid': 2, 'elu': 3, 'softplus': 4, 'swish': 5}


def pro_act(name: str) -> int:
    return PRO_AFFINE_ACT | (ACT_CODES[name] << 8)


def epi_act_bwd(name: str) -> int:
    return EPI_ACT_BWD | (ACT_CODES[name] << 8)


def _bf16(t, name):
    require_device(t, name)
    if t.dtype != torch.bfloat16:
        raise TypeError('%s must be bfloat16' % name)
    if t.stride(-1) != 1:
        raise ValueError('%s must be row-major' % name)
    return t


def pad8(n: int) -> int:
    return (n + 7) // 8 * 8


def pad_k(n: int) -> int:
    """Width the tower stages its INPUT features at: a multiple of the GEMM's 64-wide k step once there are at
    least two steps of it (the LDS-DMA kernels need whole k steps; the padding columns are zeros on both operands)."""
    return ((n + 63) // 64) * 64 if n > 64 else pad8(n)


def multi_add_(dsts, srcs):
    """dst += src for lists of small contiguous fp32 tensors, one launch per 16 pairs."""
    if not dsts:
        return
    n = len(dsts)
    srcs = [s.to(torch.float32).contiguous() for s in srcs]
    for d, s in zip(dsts, srcs):
        if d.dtype != torch.float32 or not d.is_contiguous() or d.numel() != s.numel():
            raise ValueError('multi_add_: contiguous fp32 tensors of equal size')
    dp = (ctypes.c_void_p * n)(*[d.data_ptr() for d in dsts])
    sp = (ctypes.c_void_p * n)(*[s.data_ptr() for s in srcs])
    nn = (ctypes.c_int * n)(*[d.numel() for d in dsts])
    _lib.check(_lib.load().tfr_tower_multi_add(dp, sp, nn, n, _stream()), 'tfr_tower_multi_add')


def flatten_row_index(mask: torch.Tensor) -> torch.Tensor:
    """int32 [B * L] row index of FlattenList's circular-padding gather (utils.py:308-356, shuffle=False):
    position p of list b reads flat row b * L + (p mod n_b)-th valid position of the list."""
    require_device(mask, 'mask')
    B, L = mask.shape
    m8 = mask.contiguous().view(torch.uint8) if mask.dtype == torch.bool else (mask != 0).view(torch.uint8).contiguous()
    rows = torch.empty((B * L,), dtype=torch.int32, device=mask.device)
    _lib.check(_lib.load().tfr_flatten_row_index(_ptr(m8), B, L, _ptr(rows), _stream()), 'tfr_flatten_row_index')
    return rows


def cast_rows(x: torch.Tensor, scale: Optional[torch.Tensor] = None, shift: Optional[torch.Tensor] = None,
              row_index: Optional[torch.Tensor] = None, width: Optional[int] = None):
    """fp32 [R, F] -> bf16 [M, pad8(F)] (zero padded), optional per-column affine; with ``row_index`` (int32 [M])
    row m of the result is row ``row_index[m]`` of ``x`` (FlattenList's gather fused into the cast).  A bfloat16 ``x``
    (bf16 feature ingest, ``data.parse_from_example_list(example_dtype=torch.bfloat16)``) is gathered and padded as it
    is: no fp32 copy of the features exists on the device."""
    require_device(x, 'x')
    from_bf16 = x.dtype == torch.bfloat16
    if not from_bf16:
        x = x.to(torch.float32)
    if x.stride(1) != 1:
        x = x.contiguous()
    F = x.shape[1]
    if row_index is not None:
        row_index = row_index.to(torch.int32).contiguous()
        M = row_index.numel()
    else:
        M = x.shape[0]
    Kp = width if width is not None else pad8(F)
    out = torch.empty((M, Kp), dtype=torch.bfloat16, device=x.device)
    entry = 'tfr_tower_cast_gather_bf16_bf16' if from_bf16 else 'tfr_tower_cast_gather_f32_bf16'
    _lib.check(getattr(_lib.load(), entry)(_ptr(x), x.stride(0), M, F, Kp, _ptr(scale), _ptr(shift),
                                           _ptr(row_index), _ptr(out), _stream()), entry)
    return out


def input_stats(x: torch.Tensor, row_index: Optional[torch.Tensor] = None, n_blocks: int = 512,
                pivot: Optional[torch.Tensor] = None):
    """Per-column partial sums [T, 2, F] (sum x, sum x^2) of the fp32 features (rows gathered through
    ``row_index``): the batch statistics of create_tower's input BatchNormalization, in bn_finalize's format.
    ``pivot`` [F]: sums of ``x - pivot`` (the variance keeps its digits when |mean| >> std; add it back to the mean).
    bfloat16 features (bf16 ingest) are read as they are."""
    require_device(x, 'x')
    from_bf16 = x.dtype == torch.bfloat16
    if not from_bf16:
        x = x.to(torch.float32)
    if x.stride(1) != 1:
        x = x.contiguous()
    if row_index is not None:
        row_index = row_index.to(torch.int32).contiguous()
        M = row_index.numel()
    else:
        M = x.shape[0]
    F = x.shape[1]
    T = max(1, min(n_blocks, (M + 63) // 64))
    partial = torch.empty((T, 2, F), dtype=torch.float32, device=x.device)
    if pivot is not None:
        pivot = pivot.to(torch.float32).contiguous()
    entry = 'tfr_tower_input_stats_bf16' if from_bf16 else 'tfr_tower_input_stats_f32'
    _lib.check(getattr(_lib.load(), entry)(_ptr(x), x.stride(0), M, F, _ptr(row_index), _ptr(partial), T,
                                           _ptr(pivot), _stream()), entry)
    return partial, M


def cast_weight(w: torch.Tensor, transpose: bool = False, pitch: Optional[int] = None):
    """fp32 [R, C] -> bf16 [R, pitch >= C] (default pad8(C)), or (transpose) bf16 [C, pad8(R)]; zero padded."""
    require_device(w, 'w')
    w = w.detach().to(torch.float32).contiguous()
    R, C = w.shape
    pitch = max(pitch or 0, pad8(R if transpose else C))
    out = torch.empty((C if transpose else R, pitch), dtype=torch.bfloat16, device=w.device)
    _lib.check(_lib.load().tfr_tower_weight_cast(_ptr(w), R, C, int(transpose), pitch, _ptr(out), _stream()),
               'tfr_tower_weight_cast')
    return out


def cast_weights(specs, step=None):
    """[(w fp32 [R, C], transpose, pitch | None), ...] -> the bf16 operands of cast_weight, one launch per 8.
    ``step`` = (counter, copy), one-element int32 device tensors: the launch also advances the Dropout step counter
    (counter += 1, copy = counter) instead of an add_ and a clone launch of their own."""
    ws, outs, Rs, Cs, trs, ps = [], [], [], [], [], []
    for w, transpose, pitch in specs:
        require_device(w, 'w')
        w = w.detach().to(torch.float32).contiguous()
        R, C = w.shape
        pitch = max(pitch or 0, pad8(R if transpose else C))
        ws.append(w); Rs.append(R); Cs.append(C); trs.append(int(bool(transpose))); ps.append(pitch)
        outs.append(torch.empty((C if transpose else R, pitch), dtype=torch.bfloat16, device=w.device))
    n = len(ws)
    if n:
        arr = lambda xs: (ctypes.c_int * n)(*xs)
        if step is not None:
            counter, copy = step
            for t in (counter, copy):
                if t.numel() != 1 or t.dtype != torch.int32 or not t.is_cuda:
                    raise ValueError('step tensors must be one-element int32 device tensors')
            _lib.check(_lib.load().tfr_tower_weight_cast_batch_step(
                (ctypes.c_void_p * n)(*[w.data_ptr() for w in ws]), arr(Rs), arr(Cs), arr(trs), arr(ps),
                (ctypes.c_void_p * n)(*[o.data_ptr() for o in outs]), n, counter.data_ptr(), copy.data_ptr(), _stream()),
                'tfr_tower_weight_cast_batch_step')
            return outs
        _lib.check(_lib.load().tfr_tower_weight_cast_batch(
            (ctypes.c_void_p * n)(*[w.data_ptr() for w in ws]), arr(Rs), arr(Cs), arr(trs), arr(ps),
            (ctypes.c_void_p * n)(*[o.data_ptr() for o in outs]), n, _stream()), 'tfr_tower_weight_cast_batch')
    elif step is not None:
        step[0].add_(1); step[1].copy_(step[0])
    return outs


def stats_rows(M: int) -> int:
    return (M + 63) // 64          # one row of partials per 64-row slab (tfr_tower_gemm_stats_rows)


def gemm_writes_operand(M: int, N: int, K: int) -> bool:
    """True where ``gemm(..., a_out=...)`` is served (the persistent 256 x 256 kernel over full tiles)."""
    return bool(_lib.load().tfr_tower_gemm_writes_operand(int(M), int(N), int(K)))


def gemm(A, B, N, K, prologue=PRO_NONE, a_scale=None, a_shift=None, bias=None, epilogue=EPI_PLAIN,
         Zp=None, e_scale=None, e_shift=None, e_mean=None, e_rstd=None, out=None, pro_dropout=None,
         epi_dropout=None, a_out=None):
    """C[M, N] = pro(A)[M, :K] . B[N, :K]^T (+ bias) as bf16; returns (C, stats_partial | None).  ``a_out`` (bf16
    [M, K], optional): the kernel also writes pro(A), the operand it forms in registers (tfr_tower_gemm_bf16_aout)."""
    _bf16(A, 'A'); _bf16(B, 'B')
    M = A.shape[0]
    C = out if out is not None else torch.empty((M, N), dtype=torch.bfloat16, device=A.device)
    stats = None
    if (epilogue & 0xff) != EPI_PLAIN:
        stats = torch.empty((stats_rows(M), 2, N), dtype=torch.float32, device=A.device)
    if a_out is not None:
        _bf16(a_out, 'a_out')
        if a_out.shape[0] != M or a_out.shape[1] < K:
            raise ValueError('a_out must be a bf16 [%d, >= %d] tensor' % (M, K))
    _lib.check(_lib.load().tfr_tower_gemm_bf16_aout(
        _ptr(A), A.stride(0), _ptr(B), B.stride(0), _ptr(C), C.stride(0), M, N, K, prologue,
        _ptr(a_scale), _ptr(a_shift), _ptr(bias), epilogue, _ptr(stats), _ptr(Zp),
        Zp.stride(0) if Zp is not None else 0, _ptr(e_scale), _ptr(e_shift), _ptr(e_mean), _ptr(e_rstd),
        _dp(pro_dropout), _dp(epi_dropout), _ptr(a_out), a_out.stride(0) if a_out is not None else 0, _stream()),
        'tfr_tower_gemm_bf16_aout')
    return C, stats


def _scratch(T, W, dev):
    """Stage-1 buffer of the two-stage column reduction (None when one stage is enough)."""
    if T <= 64:
        return None
    return torch.empty(((T + 63) // 64, W), dtype=torch.float32, device=dev)


def bn_finalize(partial, M, gamma, beta, eps, momentum, moving_mean, moving_var):
    T, _, N = partial.shape
    dev = partial.device
    scale = torch.empty(N, dtype=torch.float32, device=dev)
    shift = torch.empty_like(scale); mean = torch.empty_like(scale); rstd = torch.empty_like(scale)
    scratch = None if T <= _FUSED_ROWS else _scratch(T, 2 * N, dev)
    _lib.check(_lib.load().tfr_tower_bn_finalize(_ptr(partial), T, N, M, _ptr(gamma), _ptr(beta), eps, momentum,
                                                 _ptr(moving_mean), _ptr(moving_var), _ptr(scale), _ptr(shift),
                                                 _ptr(mean), _ptr(rstd), _ptr(scratch), _stream()),
               'tfr_tower_bn_finalize')
    return scale, shift, mean, rstd


_FUSED_ROWS = 1024          # tfr_tower_reduce_partials_coeffs / tfr_tower_bn_finalize: one launch up to this many rows


def reduce_partials(partial, bn=None, colsum_of=None):
    """[T, J, N] per-workgroup column partials -> [J, N] sums.  ``bn = (gamma, rstd, mean, M)``: also the
    BatchNorm-backward coefficients pqr [3, N] of the layer whose (sum dy, sum dy zhat) are rows 0 / 1, from the same
    launch; returns (sums, pqr) then.  ``colsum_of`` = fp32 [Mr, O <= 4] (the output layer's dlogits): its column sums
    ride in the same launch when the one-launch form serves the shape, and are appended to the result (None when not)."""
    T, J, N = partial.shape
    out = torch.empty((J, N), dtype=torch.float32, device=partial.device)
    fused = T <= _FUSED_ROWS and J <= 6
    scratch = None if fused else _scratch(T, J * N, partial.device)
    lib = _lib.load()
    db = None
    if colsum_of is not None and colsum_of.dim() == 2 and 1 <= colsum_of.shape[1] <= 4 and colsum_of.is_contiguous() \
            and lib.tfr_tower_reduce_partials_serves_db(T, J):
        db = torch.empty((colsum_of.shape[1],), dtype=torch.float32, device=partial.device)
    gamma, rstd, mean, M = bn if bn is not None else (None, None, None, 0)
    pqr = torch.empty((3, N), dtype=torch.float32, device=partial.device) if bn is not None else None
    if db is not None:
        _lib.check(lib.tfr_tower_reduce_partials_coeffs_db(
            _ptr(partial), T, J, N, _ptr(out), _ptr(scratch), _ptr(gamma.detach()) if bn is not None else None, _ptr(rstd),
            _ptr(mean), M, _ptr(pqr), _ptr(colsum_of), colsum_of.shape[0], colsum_of.shape[1], _ptr(db), _stream()),
            'tfr_tower_reduce_partials_coeffs_db')
    else:
        _lib.check(lib.tfr_tower_reduce_partials_coeffs(
            _ptr(partial), T, J, N, _ptr(out), _ptr(scratch), _ptr(gamma.detach()) if bn is not None else None, _ptr(rstd),
            _ptr(mean), M, _ptr(pqr), _stream()), 'tfr_tower_reduce_partials_coeffs')
    res = (out,) if bn is None else (out, pqr)
    if colsum_of is not None:
        return res + (db,)
    return res[0] if bn is None else res


def out_layer(z, K, prologue, scale, shift, w, b, dropout=None):
    """logits[M, O] = act(z)[M, :K] . w[O, K]^T + b (fp32)."""
    _bf16(z, 'z')
    M = z.shape[0]
    w = w.detach().to(torch.float32).contiguous()
    O = w.shape[0]
    out = torch.empty((M, O), dtype=torch.float32, device=z.device)
    _lib.check(_lib.load().tfr_tower_out_f32(_ptr(z), z.stride(0), M, K, prologue, _ptr(scale), _ptr(shift),
                                             _ptr(w), _ptr(b), O, _ptr(out), _dp(dropout), _stream()),
               'tfr_tower_out_f32')
    return out


_OUT_BWD_ROWS = int(os.environ.get('TFR_OUT_BWD_ROWS', '128'))   # rows per workgroup below which fewer blocks are launched


def out_layer_bwd(z, K, prologue, scale, shift, mean, rstd, w, dlogits, n_blocks=1024, dropout=None, bn=None):
    """Output-layer backward: returns (dy bf16 [M, K], sums [2 + O, K], db) with
    sums[0] = sum dy, sums[1] = sum dy * zhat, sums[2 + o] = d w[o, :]; with ``bn = (gamma, rstd, mean, M)`` of the
    last hidden layer (dy, sums, pqr, db): its BatchNorm-backward coefficients pqr (same launch as the sums).
    db = the column sums of dlogits [O] from the same launch, or None (large M: the caller adds them up itself)."""
    _bf16(z, 'z')
    M = z.shape[0]
    w = w.detach().to(torch.float32).contiguous()
    O = w.shape[0]
    dlogits = dlogits.to(torch.float32).contiguous()
    n_blocks = max(1, min(n_blocks, (M + 15) // 16, max(256, M // _OUT_BWD_ROWS)))
    dy = torch.empty((M, K), dtype=torch.bfloat16, device=z.device)
    partial = torch.empty((n_blocks, 2 + O, K), dtype=torch.float32, device=z.device)
    _lib.check(_lib.load().tfr_tower_out_bwd(_ptr(z), z.stride(0), M, K, prologue, _ptr(scale), _ptr(shift),
                                             _ptr(mean), _ptr(rstd), _ptr(w), _ptr(dlogits), O, _ptr(dy),
                                             dy.stride(0), _ptr(partial), n_blocks, _dp(dropout), _stream()),
               'tfr_tower_out_bwd')
    if bn is None:
        sums, db = reduce_partials(partial, None, colsum_of=dlogits)
        return dy, sums, db
    sums, pqr, db = reduce_partials(partial, bn, colsum_of=dlogits)
    return dy, sums, pqr, db


def out_layer_bwd_bn(z, K, prologue, scale, shift, mean, rstd, gamma, w, dlogits, n_blocks=1024, dropout=None):
    """Output-layer backward when the last hidden layer is BatchNorm'd, in two passes over z so the
    [M, K] gradient is written once: pass 1 = column sums only, pass 2 recomputes dy and writes
    dz = p * bf16(dy) + q * z + r.  Returns (dz bf16 [M, K], sums [2 + O, K], db | None) -- dz and sums bit-identical to
    out_layer_bwd + bn_bwd_coeffs + bn_bwd_apply_."""
    _bf16(z, 'z')
    M = z.shape[0]
    w = w.detach().to(torch.float32).contiguous()
    O = w.shape[0]
    dlogits = dlogits.to(torch.float32).contiguous()
    n_blocks = max(1, min(n_blocks, (M + 15) // 16, max(256, M // _OUT_BWD_ROWS)))
    lib = _lib.load()
    partial = torch.empty((n_blocks, 2 + O, K), dtype=torch.float32, device=z.device)
    _lib.check(lib.tfr_tower_out_bwd2(_ptr(z), z.stride(0), M, K, prologue, _ptr(scale), _ptr(shift), _ptr(mean),
                                      _ptr(rstd), _ptr(w), _ptr(dlogits), O, None, K, _ptr(partial), n_blocks,
                                      _dp(dropout), None, _stream()), 'tfr_tower_out_bwd2')
    sums, pqr, db = reduce_partials(partial, (gamma, rstd, mean, M), colsum_of=dlogits)
    dz = torch.empty((M, K), dtype=torch.bfloat16, device=z.device)
    n2 = max(1, min(4096, (M + 15) // 16))
    _lib.check(lib.tfr_tower_out_bwd2(_ptr(z), z.stride(0), M, K, prologue, _ptr(scale), _ptr(shift), _ptr(mean),
                                      _ptr(rstd), _ptr(w), _ptr(dlogits), O, _ptr(dz), dz.stride(0), None, n2,
                                      _dp(dropout), _ptr(pqr), _stream()), 'tfr_tower_out_bwd2')
    return dz, sums, db


def bn_bwd_coeffs(gamma, rstd, mean, c, M):
    """pqr [3, N] of dz = p dy + q z + r from c = [sum dy; sum dy zhat] ([2, N], contiguous)."""
    N = gamma.shape[0]
    c = c.to(torch.float32).contiguous()
    pqr = torch.empty((3, N), dtype=torch.float32, device=c.device)
    _lib.check(_lib.load().tfr_tower_bn_bwd_coeffs(_ptr(gamma.detach()), _ptr(rstd), _ptr(mean), _ptr(c), N, M,
                                                   _ptr(pqr), _stream()), 'tfr_tower_bn_bwd_coeffs')
    return pqr


def bn_bwd_apply_(dy, z, K, pqr):
    """In place: dy <- p * dy + q * z + r (per column); pqr is fp32 [3, K]."""
    _bf16(dy, 'dy'); _bf16(z, 'z')
    pqr = pqr.to(torch.float32).contiguous()
    _lib.check(_lib.load().tfr_tower_bn_bwd_apply(_ptr(dy), dy.stride(0), _ptr(z), z.stride(0), dy.shape[0], K,
                                                  _ptr(pqr), _stream()), 'tfr_tower_bn_bwd_apply')
    return dy


_WGRAD_BLOCKS = int(os.environ.get('TFR_WGRAD_BLOCKS', '512'))   # split-M target: workgroups per launch


def wgrad(dz, A, N, K, prologue=PRO_NONE, a_scale=None, a_shift=None, splits=0, dropout=None, accumulate_into=None,
          out_cols=None):
    """dW[N, K] = dz[M, :N]^T . pro(A)[M, :K] (fp32).  ``accumulate_into`` (contiguous fp32 [N, K]): the split
    reduction adds into it instead of returning a fresh tensor (gradient accumulation without another launch).
    ``out_cols`` < K: only the first out_cols columns are wanted (A is staged wider than the weight matrix: k-step
    padding) -- the result / accumulate_into is [N, out_cols], written by the reduction itself."""
    _bf16(dz, 'dz'); _bf16(A, 'A')
    M = dz.shape[0]
    if splits <= 0:
        tiles = ((N + 127) // 128) * ((K + 127) // 128)
        splits = max(1, min((M + 255) // 256, (_WGRAD_BLOCKS + tiles - 1) // tiles))
        if N % 256 == 0 and K % 64 == 0 and K >= 128 and (dropout is None or not dropout.threshold16):
            # the 256 x 256 kernel: one workgroup per CU, M split into equal runs of 64-row steps
            want = max(1, 256 // ((N // 256) * ((K + 255) // 256)))
            fit = [s for s in range(want, 0, -1) if M % (64 * s) == 0]
            if fit and fit[0] * 2 > want:
                splits = fit[0]
    slab = torch.empty((splits, N, K), dtype=torch.float32, device=dz.device)
    lib = _lib.load()
    _lib.check(lib.tfr_tower_wgrad_bf16(_ptr(dz), dz.stride(0), _ptr(A), A.stride(0), M, N, K, prologue,
                                        _ptr(a_scale), _ptr(a_shift), _ptr(slab), K, splits, _dp(dropout), _stream()),
               'tfr_tower_wgrad_bf16')
    Ko = K if out_cols is None else int(out_cols)
    if not 0 < Ko <= K:
        raise ValueError('out_cols must be in (0, K]')
    if accumulate_into is not None:
        out = accumulate_into
        if out.shape != (N, Ko) or out.dtype != torch.float32 or not out.is_contiguous():
            raise ValueError('accumulate_into must be a contiguous fp32 [%d, %d] tensor' % (N, Ko))
    else:
        out = torch.empty((N, Ko), dtype=torch.float32, device=dz.device)
    if Ko == K:
        _lib.check(lib.tfr_tower_slab_reduce(_ptr(slab), splits, N * K, _ptr(out), 1 if accumulate_into is not None else 0,
                                             _stream()), 'tfr_tower_slab_reduce')
    else:
        _lib.check(lib.tfr_tower_slab_reduce_cols(_ptr(slab), splits, N, K, Ko, _ptr(out),
                                                  1 if accumulate_into is not None else 0, _stream()),
                   'tfr_tower_slab_reduce_cols')
    return out


_ops._guard_module(globals(), __name__, skip=('pad8', 'pad_k', 'stats_rows', 'dropout_mask', 'dropout_field', 'gemm_writes_operand'))


# ---- fp32 Dense on the matrix cores (csrc/gemm_f32.hip): the reference's own precision (keras/layers.py:26-77) ----
def _f32_2d(t, name):
    require_device(t, name)
    if t.dtype != torch.float32 or t.dim() != 2:
        raise TypeError('%s must be a 2-D float32 tensor' % name)
    if t.stride(1) != 1 or (t.shape[0] > 1 and t.stride(0) < t.shape[1]):
        t = t.contiguous()
    return t


def gemm_f32(A, a_k_contiguous, B, b_k_contiguous, M, N, K, bias=None, splits=1, out=None):
    """C[M, N] = op(A)[M, K] . op(B)[K, N] (+ bias), fp32 in / fp32 accumulate (tfr_tower_gemm_f32)."""
    A = _f32_2d(A, 'A'); B = _f32_2d(B, 'B')
    C = out if out is not None else torch.empty((M, N), dtype=torch.float32, device=A.device)
    if bias is not None:
        bias = bias.detach().to(torch.float32).contiguous()
    ws = torch.empty((splits, M, N), dtype=torch.float32, device=A.device) if splits > 1 else None
    lda = A.stride(0) if A.shape[0] > 1 else max(A.shape[1], 1)
    ldb = B.stride(0) if B.shape[0] > 1 else max(B.shape[1], 1)
    _lib.check(_lib.load().tfr_tower_gemm_f32(_ptr(A), lda, int(a_k_contiguous), _ptr(B), ldb, int(b_k_contiguous),
                                              _ptr(C), C.stride(0) if M > 1 else max(N, 1), M, N, K, _ptr(bias), splits,
                                              _ptr(ws), _stream()), 'tfr_tower_gemm_f32')
    return C


def dense_f32(x, w, bias=None):
    """y[M, N] = x[M, K] . w[N, K]^T + bias."""
    return gemm_f32(x, True, w, True, x.shape[0], w.shape[0], x.shape[1], bias=bias)


def dense_f32_dgrad(dy, w):
    """dx[M, K] = dy[M, N] . w[N, K]."""
    return gemm_f32(dy, True, w, False, dy.shape[0], w.shape[1], w.shape[0])


def dense_f32_wgrad(dy, x):
    """dW[N, K] = dy[M, N]^T . x[M, K]: the contraction over the M rows cut into slabs, summed in a fixed order."""
    N, K, M = dy.shape[1], x.shape[1], x.shape[0]
    splits = int(_lib.load().tfr_tower_gemm_f32_splits(N, K, M))
    return gemm_f32(dy, False, x, False, N, K, M, splits=splits)


def colsum_f32(x):
    """out[n] = sum_m x[m, n] (the bias gradient), two deterministic stages (tfr_tower_colsum_f32)."""
    x = _f32_2d(x, 'x')
    M, N = x.shape
    out = torch.empty((N,), dtype=torch.float32, device=x.device)
    T = int(_lib.load().tfr_tower_colsum_rows(M))
    partial = torch.empty((T, max(N, 1)), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().tfr_tower_colsum_f32(_ptr(x), x.stride(0) if M > 1 else max(N, 1), M, N, _ptr(partial), _ptr(out),
                                                _stream()), 'tfr_tower_colsum_f32')
    return out
