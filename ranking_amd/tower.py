"""MI355X-native feed-forward scorer tower.

Mirror of ``tfr.keras.layers.create_tower`` (keras/layers.py:26-77): per hidden layer
``Dense -> BatchNormalization -> Activation -> Dropout``, then ``Dense(output_units)``,
evaluated on the flattened ``[B * L, F]`` matrix (keras/model.py:800-817).  The
reference runs it as separate fp32 TensorFlow ops; here a hidden layer is ONE bf16
MFMA GEMM launch with the previous layer's BatchNorm + ReLU fused into its operand
load and bias / bf16 cast / BatchNorm statistics fused into its epilogue
(``csrc/tower.hip``), forward and backward.  Parameters stay fp32 (master copy); the
bf16 operand copies are rebuilt per call.

What the fused path covers: ``activation`` in {None, relu, tanh, sigmoid, elu, softplus, swish}, ``use_batch_norm`` on or
off, training (batch statistics, moving averages updated; Dropout as a counter-based keep mask
shared by forward and backward -- the TF random stream itself is not reproducible) and inference
(moving averages, no dropout).  ``input_batch_norm``: batch statistics of the raw features by
``tfr_tower_input_stats_f32``, the normalisation inside the input cast, its two parameter gradients from
the layer-0 dgrad's column partials.
"""
from __future__ import annotations

from typing import List, Optional

import os
import torch
from torch import nn

from . import _tower_ops as T

_BN_EPS = 1e-3          # tf.keras.layers.BatchNormalization default epsilon


_ACT_ALIASES = {
    'relu': ('relu', torch.relu, torch.nn.functional.relu),
    'tanh': ('tanh', torch.tanh, torch.nn.functional.tanh),
    'sigmoid': ('sigmoid', torch.sigmoid, torch.nn.functional.sigmoid),
    'elu': ('elu', torch.nn.functional.elu),
    'softplus': ('softplus', torch.nn.functional.softplus),
    'swish': ('swish', 'silu', torch.nn.functional.silu),
}
_ACT_MODULES = {nn.ReLU: 'relu', nn.Tanh: 'tanh', nn.Sigmoid: 'sigmoid', nn.SiLU: 'swish'}


def _act_code(activation) -> Optional[str]:
    """keras/layers.py:66-70 takes any Keras activation; the fused kernels know None, relu, tanh, sigmoid, elu
    (alpha = 1), softplus and swish (= silu)."""
    if activation is None:
        return None
    for name, aliases in _ACT_ALIASES.items():
        if any(activation is a or (isinstance(activation, str) and activation == a) for a in aliases):
            return name
    for cls, name in _ACT_MODULES.items():
        if isinstance(activation, cls):
            return name
    if isinstance(activation, nn.ELU) and activation.alpha == 1.0:
        return 'elu'
    if isinstance(activation, nn.Softplus) and activation.beta == 1.0:
        return 'softplus'
    raise ValueError('FusedTower supports activation None, relu, tanh, sigmoid, elu, softplus or swish, got %r'
                     % (activation,))


def _pro_of(act: Optional[str]) -> int:
    return T.PRO_AFFINE if act is None else (T.PRO_AFFINE_RELU if act == 'relu' else T.pro_act(act))


_FUSED_LAST_MIN_ELEMS = 1 << 25
# The forward GEMM of a layer >= 1 writes its transformed operand for the layer's weight gradient (tfr_tower_gemm_bf16_aout):
# 0 never, 1 when a Dropout follows the layer below (the weight-gradient kernel would otherwise hash per element and fall
# back to the 128 x 128 form), 2 always.  TFR_TOWER_AOUT overrides (developer A/B).
_AOUT_MODE = int(os.environ.get('TFR_TOWER_AOUT', '1'))


def _grad_list(dW, db, dgam, dbet, use_bn, n_h, dw_out, db_out, extra):
    """the gradients in the order of _TowerFn's `params` (weights, biases, [gammas, betas], [input BN], out weight / bias)"""
    g = list(dW) + list(db)
    if use_bn:
        g += list(dgam) + list(dbet)
    if extra:
        g += list(extra)
    return g + [dw_out, db_out]


def _layer_of(j, n_h, use_bn):
    """hidden layer a position of `params` belongs to (the output layer counts as n_h: it is final first); valid without an
    input BatchNormalization"""
    per = 4 if use_bn else 2
    return j % n_h if j < per * n_h else n_h


class _TowerFn(torch.autograd.Function):
    """forward(x, training, tower, *params) -> logits [M, O] (fp32)."""

    @staticmethod
    def forward(ctx, x, tower, training, row_index, *params):
        n_h = len(tower.hidden_layer_dims)
        use_bn, act = tower.use_batch_norm, tower.activation
        Ws = params[0:n_h]
        bs = params[n_h:2 * n_h]
        gammas = params[2 * n_h:3 * n_h] if use_bn else [None] * n_h
        betas = params[3 * n_h:4 * n_h] if use_bn else [None] * n_h
        w_out, b_out = params[-2], params[-1]
        dev = x.device
        # d loss / d features (the tower under trainable layers, keras/model.py:755-817): the layer-0 dgrad in backward
        ctx.want_dx = bool(ctx.needs_input_grad[0])
        ctx.x_shape, ctx.x_dtype, ctx.row_index = tuple(x.shape), x.dtype, row_index
        in_bn = None                                       # (scale, shift) of create_tower's input BatchNormalization
        ctx.in_stats = None
        if tower.input_batch_norm:
            g_in, b_in = params[-4], params[-3]
            if training:                                   # batch statistics of the raw fp32 features
                # shifted by a sample of every column (the first scored row): var = E[x^2] - mean^2 from fp32 partial
                # sums keeps its digits for raw features with |mean| >> std (Keras uses two-pass moments)
                first = x[0] if row_index is None else x.index_select(0, row_index[:1].long())[0]
                pivot = first.detach().to(torch.float32)
                part, rows = T.input_stats(x, row_index=row_index, pivot=pivot)
                in_sc, in_sh, in_mean, in_rstd = T.bn_finalize(part, rows, g_in, b_in, _BN_EPS, tower.momentum,
                                                               tower.moving_mean_in, tower.moving_var_in)
                in_mean = in_mean + pivot                               # finalize saw x - pivot: mean, shift and the moving
                in_sh = in_sh - pivot * in_sc                           # mean take the pivot back; scale / rstd / var do not
                tower.moving_mean_in.add_(pivot, alpha=1.0 - tower.momentum)
            else:
                in_rstd = torch.rsqrt(tower.moving_var_in + _BN_EPS)
                in_mean = tower.moving_mean_in
                in_sc = g_in.detach() * in_rstd
                in_sh = b_in.detach() - in_mean * in_sc
            in_bn = (in_sc, in_sh)
            ctx.in_stats = (in_mean, in_rstd, x)           # backward: xhat from the fp32 features, not from the bf16 copy
            x0 = T.cast_rows(x, scale=in_sc, shift=in_sh, row_index=row_index, width=T.pad_k(x.shape[1]))
        elif row_index is not None or not (x.dtype == torch.bfloat16 and x.shape[1] == T.pad_k(x.shape[1])):
            # fp32 features, bf16-ingested features at their own width (data.parse_from_example_list(example_dtype=
            # bfloat16)), or a row gather: one launch writes the bf16 operand at the k-step pitch
            x0 = T.cast_rows(x, row_index=row_index, width=T.pad_k(x.shape[1]))
        else:
            x0 = x                                         # already staged (the groupwise gather writes this layout)
        M = x0.shape[0]
        a_in, pro, sc, sh, drop = x0, T.PRO_NONE, None, None, None
        zs, coefs = [], []
        k_in = x0.shape[1]
        rate = tower.dropout if training else 0.0
        step_t = None
        if rate > 0.0:
            # the step counter lives on the device: a hipGraph replay of this forward draws a new mask, and the backward of
            # THIS forward reads the copy it saved whatever runs in between.  Counter += 1 and the copy happen inside the
            # weight-cast launch below (round 6; they were an add_ and a clone launch of their own)
            step_t = torch.empty_like(tower._drop_counter)
            base = torch.initial_seed() & 0xffffffff
        # every weight cast of the step in one launch: [N, k_in] forward operands (k_in = staged width of the layer
        # input) and, when a backward will follow, the transposed [K, pad8(N)] dgrad operands of layers >= 1
        specs = [(Ws[l], False, k_in if l == 0 else Ws[l - 1].shape[0]) for l in range(n_h)]
        want_bwd = any(ctx.needs_input_grad[4:])
        if want_bwd:
            specs += [(Ws[l], True, None) for l in range(1, n_h)]
        cast = T.cast_weights(specs, step=(tower._drop_counter, step_t) if step_t is not None else None)
        wbs, ctx.wts = cast[:n_h], [None] + cast[n_h:]
        a_outs = [None] * n_h
        for l in range(n_h):
            n_out = Ws[l].shape[0]
            wb = wbs[l]
            # layers >= 1 with a backward to come: the GEMM also writes the operand it forms in registers --
            # act(BatchNorm(z)) times the keep mask -- for this layer's weight gradient (see _AOUT_MODE)
            if (want_bwd and pro != T.PRO_NONE and (_AOUT_MODE == 2 or (_AOUT_MODE == 1 and drop is not None))
                    and T.gemm_writes_operand(M, n_out, k_in)):
                a_outs[l] = torch.empty((M, k_in), dtype=torch.bfloat16, device=dev)
            z, stats = T.gemm(a_in, wb, n_out, k_in, prologue=pro, a_scale=sc, a_shift=sh, bias=bs[l],
                              epilogue=T.EPI_STATS if (use_bn and training) else T.EPI_PLAIN, pro_dropout=drop,
                              a_out=a_outs[l])
            if use_bn:
                if training:
                    sc, sh, mean, rstd = T.bn_finalize(stats, M, gammas[l], betas[l], _BN_EPS, tower.momentum,
                                                       tower.moving_mean[l], tower.moving_var[l])
                else:
                    rstd = torch.rsqrt(tower.moving_var[l] + _BN_EPS)
                    mean = tower.moving_mean[l]
                    sc = gammas[l].detach() * rstd
                    sh = betas[l].detach() - mean * sc
                pro = _pro_of(act)
            else:
                mean = rstd = None
                if act is not None or rate > 0.0:
                    sc = torch.ones(n_out, device=dev); sh = torch.zeros(n_out, device=dev)
                    pro = _pro_of(act)
                else:
                    sc = sh = None
                    pro = T.PRO_NONE
            # Dropout after this layer's activation (keras/layers.py:72-73): applied by its consumers
            drop = T.Dropout.make(rate, base + l * 0x632BE5AB, step=step_t) if rate > 0.0 else None
            zs.append(z); coefs.append((pro, sc, sh, mean, rstd, drop))
            a_in, k_in = z, n_out
        logits = T.out_layer(a_in, k_in, pro, sc, sh, w_out, b_out, dropout=drop)
        ctx.tower, ctx.training = tower, training
        tower._last_drops = [c[5] for c in coefs]
        ctx.x0, ctx.zs, ctx.coefs, ctx.in_bn = x0, zs, coefs, in_bn
        ctx.a_outs = a_outs
        ctx.params = params
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        tower, params = ctx.tower, ctx.params
        n_h = len(tower.hidden_layer_dims)
        use_bn = tower.use_batch_norm
        if use_bn and not ctx.training:
            raise RuntimeError('FusedTower backward in inference mode (moving statistics) is not supported')
        Ws = params[0:n_h]
        gammas = params[2 * n_h:3 * n_h] if use_bn else [None] * n_h
        w_out = params[-2]
        x0, zs, coefs = ctx.x0, ctx.zs, ctx.coefs
        M = x0.shape[0]
        dev = x0.device
        dlogits = dlogits.to(torch.float32).contiguous()
        # Opt-in (FusedTower.accumulate_grads_in_place): add into the existing .grad buffers here -- the weight
        # gradients inside the split reduction, the vectors in ONE multi-tensor launch -- and hand autograd None,
        # instead of ~14 separate "grad += g" launches.  Parameter hooks do not fire in this mode.
        direct = bool(getattr(tower, 'accumulate_grads_in_place', False)) and all(
            p.grad is not None and p.grad.dtype == torch.float32 and p.grad.is_contiguous() for p in params)
        dW, db = [None] * n_h, [None] * n_h
        dgam, dbet = [None] * n_h, [None] * n_h
        split = int(getattr(tower, 'grad_split', 0) or 0) if direct else 0
        if split and tower.input_batch_norm:
            raise RuntimeError('FusedTower.grad_split with input_batch_norm is not supported')
        params_all, flushed = params, set()
        # output layer
        pro, sc, sh, mean, rstd, drop = coefs[-1]
        n_last = zs[-1].shape[1]
        # two passes over z pay once the [M, K] gradient is HBM-sized; below that the extra launch costs more
        fused_last = (use_bn and n_h > 0 and M * n_last >= _FUSED_LAST_MIN_ELEMS
                      and not os.environ.get('TFR_TOWER_NO_FUSED_LAST'))
        pqr = None                                         # BatchNorm-backward coefficients of the layer at hand
        if fused_last:                                     # dz of the last hidden layer directly (two passes over z)
            dy, sums, db_out = T.out_layer_bwd_bn(zs[-1], n_last, pro, sc, sh, mean, rstd, gammas[-1], w_out, dlogits,
                                                  dropout=drop)
        elif use_bn:                                       # (sum dy, sum dy zhat) and the coefficients in one launch
            dy, sums, pqr, db_out = T.out_layer_bwd(zs[-1], n_last, pro, sc, sh, mean, rstd, w_out, dlogits, dropout=drop,
                                                    bn=(gammas[-1], rstd, mean, M))
        else:
            dy, sums, db_out = T.out_layer_bwd(zs[-1], n_last, pro, sc, sh, mean, rstd, w_out, dlogits, dropout=drop)
        dw_out = sums[2:].contiguous()
        if db_out is None:                                 # (large M: the partial reduction is not the one-launch form)
            db_out = dlogits.sum(dim=0)
        cc = sums                                          # rows 0 / 1: sum dy, sum dy * zhat of the layer below
        # (a bias below BatchNorm has no gradient: zeros for autograd; the in-place mode skips them -- no fill launch)
        db_zero = (torch.zeros(sum(z.shape[1] for z in zs), device=dev).split([z.shape[1] for z in zs])
                   if (use_bn and not direct) else [None] * n_h)
        for l in range(n_h - 1, -1, -1):
            n_out = zs[l].shape[1]
            pro_l, sc_l, sh_l, mean_l, rstd_l, _ = coefs[l]
            if use_bn:
                dgam[l], dbet[l] = cc[1], cc[0]
                dz = dy if (fused_last and l == n_h - 1) else T.bn_bwd_apply_(dy, zs[l], n_out, pqr)
                db[l] = db_zero[l]                         # a bias below BatchNorm has no gradient
            else:
                dz = dy
                db[l] = cc[0]
            if l > 0:
                pro_p, sc_p, sh_p, mean_p, rstd_p, drop_p = coefs[l - 1]
                a_prev, k_in = zs[l - 1], zs[l - 1].shape[1]
            else:
                pro_p, sc_p, sh_p, mean_p, rstd_p, drop_p = T.PRO_NONE, None, None, None, None, None
                a_prev, k_in = x0, x0.shape[1]
            into = Ws[l].grad if (direct and Ws[l].grad.is_contiguous()) else None
            w_cols = Ws[l].shape[1] if k_in != Ws[l].shape[1] else None      # (input staged wider than W: k-step padding)
            if l == 0:
                dz0 = dz
            if ctx.a_outs[l] is not None:                    # the forward GEMM left act(BN(z)) * mask behind: no prologue
                g = T.wgrad(dz, ctx.a_outs[l], n_out, k_in, prologue=T.PRO_NONE, accumulate_into=into, out_cols=w_cols)
                ctx.a_outs[l] = None
            else:
                g = T.wgrad(dz, a_prev, n_out, k_in, prologue=pro_p, a_scale=sc_p, a_shift=sh_p, dropout=drop_p,
                            accumulate_into=into, out_cols=w_cols)
            if into is None:
                dW[l] = g
            if l > 0:
                wt = ctx.wts[l] if len(ctx.wts) > l else T.cast_weight(Ws[l], transpose=True)   # [K, pad8(N)]
                epi = T.EPI_RELU_BWD
                if pro_p == T.PRO_AFFINE_RELU:
                    e_sc, e_sh = sc_p, sh_p
                elif (pro_p & 0xff) == T.PRO_AFFINE_ACT:             # act'(z * scale + shift)
                    e_sc, e_sh, epi = sc_p, sh_p, T.epi_act_bwd(tower.activation)
                else:                                                # identity activation: mask always on
                    e_sc, e_sh = torch.zeros(k_in, device=dev), torch.ones(k_in, device=dev)
                e_mean = mean_p if mean_p is not None else torch.zeros(k_in, device=dev)
                e_rstd = rstd_p if rstd_p is not None else torch.ones(k_in, device=dev)
                dy, partial = T.gemm(dz, wt, k_in, n_out, prologue=T.PRO_NONE, epilogue=epi,
                                     Zp=zs[l - 1], e_scale=e_sc, e_shift=e_sh, e_mean=e_mean, e_rstd=e_rstd,
                                     epi_dropout=drop_p)
                if use_bn:                                 # the column sums of layer l - 1 and its coefficients together
                    cc, pqr = T.reduce_partials(partial, (gammas[l - 1], rstd_p, mean_p, M))
                else:
                    cc = T.reduce_partials(partial)
            if direct and l == split and l > 0:
                # everything of the layers >= l and of the output layer is computed: add the vector gradients into their
                # buffers now (the weight gradients went there inside the split reduction) and tell the caller
                early = [(j, g_) for j, g_ in enumerate(_grad_list(dW, db, dgam, dbet, use_bn, n_h, dw_out, db_out, None))
                         if g_ is not None and _layer_of(j, n_h, use_bn) >= l]
                T.multi_add_([params_all[j].grad for j, _ in early], [g_ for _, g_ in early])
                flushed = set(j for j, _ in early)
                if tower.grad_split_hook is not None:
                    tower.grad_split_hook(tower)
        grads = list(dW) + list(db)
        if use_bn:
            grads += list(dgam) + list(dbet)
        dx = None
        if tower.input_batch_norm or ctx.want_dx:
            # layer-0 dgrad dxin = dz_0 . W_0 ([M, F], bf16 out of the MFMA kernel like every other dgrad)
            F = Ws[0].shape[1]
            k0 = x0.shape[1]
            wt0 = T.cast_weight(Ws[0], transpose=True, pitch=T.pad8(Ws[0].shape[0]))      # [F, pad8(N)]
            if wt0.shape[0] != k0:                                                       # rows up to the staged width
                wt0 = torch.cat([wt0, torch.zeros((k0 - wt0.shape[0], wt0.shape[1]), dtype=wt0.dtype, device=dev)])
            dxin, _ = T.gemm(dz0, wt0, k0, Ws[0].shape[0], prologue=T.PRO_NONE, epilogue=T.EPI_PLAIN)
            dxin = dxin[:, :F].to(torch.float32)
            if tower.input_batch_norm:
                # BatchNormalization backward on the raw features: xhat comes from the saved batch mean / rstd and the
                # fp32 features (round 2 recovered it from the bf16-staged input as (x0 - beta) / gamma: inf / NaN for
                # gamma = 0 and bf16 noise amplified when |beta| >> |gamma| -- ADVICE r2)
                g_in = params[-4]
                in_mean, in_rstd, x_raw = ctx.in_stats
                xr = x_raw if ctx.row_index is None else x_raw.index_select(0, ctx.row_index.long())
                xhat = (xr.to(torch.float32) - in_mean) * in_rstd
                d_gamma_in = (dxin * xhat).sum(dim=0)
                d_beta_in = dxin.sum(dim=0)
                grads += [d_gamma_in, d_beta_in]
                if ctx.want_dx:
                    gsc = g_in.detach() * in_rstd
                    if ctx.training:                       # through the batch statistics
                        dx = gsc * (dxin - d_beta_in / M - xhat * (d_gamma_in / M))
                    else:
                        dx = gsc * dxin
            elif ctx.want_dx:
                dx = dxin
            if dx is not None and ctx.row_index is not None:
                # FlattenList's circular padding scores an item several times: the adjoint of the gather adds them up
                full = torch.zeros((ctx.x_shape[0], F), dtype=torch.float32, device=dev)
                dx = full.index_add_(0, ctx.row_index.long(), dx)
            if dx is not None and dx.shape[1] != ctx.x_shape[1]:          # the caller passed a k-step padded input
                dx = torch.nn.functional.pad(dx, (0, ctx.x_shape[1] - dx.shape[1]))
        grads += [dw_out, db_out]
        if direct:
            todo = [(p.grad, g) for j, (p, g) in enumerate(zip(params, grads)) if g is not None and j not in flushed]         # (zero bias grads are None here)
            T.multi_add_([a for a, _ in todo], [g for _, g in todo])
            grads = [None] * len(grads)
        return (None if dx is None else dx.to(ctx.x_dtype), None, None, None) + tuple(grads)


class FusedTower(nn.Module):
    """create_tower(...) as one fused module; call with the flattened ``[M, F]`` features.

    ``dropout``: the keep mask is a counter-based hash evaluated inside the kernels (nothing is stored); a rate that is
    a multiple of 1/2, 1/4, 1/16 or 1/256 is applied exactly through 1- to 8-bit fields of the hash, any other rate
    through 16-bit fields, i.e. to 1 / 65 536 (0.1 -> 6554 / 65536) with the scale 1 / (1 - rate) following the threshold,
    so the mask stays unbiased (``_tower_ops.dropout_field`` returns what the kernels use).  The training-step counter of the masks lives in
    device memory: a hipGraph replay of a step draws a new mask."""

    def __init__(self, input_dim: int, hidden_layer_dims: List[int], output_units: int = 1, activation=None,
                 use_batch_norm: bool = True, batch_norm_moment: float = 0.999, dropout: float = 0.0,
                 input_batch_norm: bool = False):
        super().__init__()
        if not hidden_layer_dims:
            raise ValueError('FusedTower needs at least one hidden layer')
        if any(int(h) % 8 for h in hidden_layer_dims):
            raise ValueError('FusedTower needs hidden widths that are multiples of 8, got %r' % (hidden_layer_dims,))
        if not 1 <= int(output_units) <= 4:
            raise ValueError('FusedTower supports 1..4 output units')
        self.input_dim = int(input_dim)
        self.hidden_layer_dims = [int(h) for h in hidden_layer_dims]
        self.output_units = int(output_units)
        self.activation = _act_code(activation)
        self.use_batch_norm = bool(use_batch_norm)
        self.input_batch_norm = bool(input_batch_norm)
        self.momentum = float(batch_norm_moment)
        if not 0.0 <= float(dropout or 0.0) < 1.0:
            raise ValueError('dropout rate must be in [0, 1)')
        self.dropout = float(dropout or 0.0)
        # training-step counter of the Dropout masks, on the device (see _TowerFn.forward)
        self.register_buffer('_drop_counter', torch.zeros(1, dtype=torch.int32), persistent=False)
        self._last_drops = []
        # True: backward adds into the existing .grad buffers itself (see _TowerFn.backward); set by
        # distributed.FlatGradBucket.attach(), whose flat buffer owns every .grad for the life of the model.
        self.accumulate_grads_in_place = False
        # Overlapped gradient exchange (round 6; distributed.SplitStep): with `grad_split = k` (1 <= k < number of hidden
        # layers) and in-place accumulation on, backward calls `grad_split_hook(self)` at the moment the gradients of the
        # output layer and of the hidden layers >= k are FINAL in their .grad buffers -- the layers below are still to be
        # differentiated -- so that the caller can start their all-reduce (or end a hipGraph capture there) while the rest
        # of the backward runs.  The reference's MirroredStrategy overlaps per-variable all-reduces with the backward the
        # same way (keras/strategy_utils.py:87-116).
        self.grad_split = 0
        self.grad_split_hook = None
        self.weights = nn.ParameterList()
        self.biases = nn.ParameterList()
        self.gammas = nn.ParameterList()
        self.betas = nn.ParameterList()
        self._mm, self._mv = [], []
        width = self.input_dim
        for i, h in enumerate(self.hidden_layer_dims):
            w = torch.empty(h, width)
            nn.init.xavier_uniform_(w)                      # Keras Dense: glorot_uniform, zero bias
            self.weights.append(nn.Parameter(w))
            self.biases.append(nn.Parameter(torch.zeros(h)))
            if self.use_batch_norm:
                self.gammas.append(nn.Parameter(torch.ones(h)))
                self.betas.append(nn.Parameter(torch.zeros(h)))
                self.register_buffer('moving_mean_%d' % i, torch.zeros(h))
                self.register_buffer('moving_var_%d' % i, torch.ones(h))
            width = h
        if self.input_batch_norm:                           # keras/layers.py:57-60
            self.gamma_in = nn.Parameter(torch.ones(self.input_dim))
            self.beta_in = nn.Parameter(torch.zeros(self.input_dim))
            self.register_buffer('moving_mean_in', torch.zeros(self.input_dim))
            self.register_buffer('moving_var_in', torch.ones(self.input_dim))
        w = torch.empty(self.output_units, width)
        nn.init.xavier_uniform_(w)
        self.out_weight = nn.Parameter(w)
        self.out_bias = nn.Parameter(torch.zeros(self.output_units))

    def dropout_structs(self):
        """The Dropout structs (step counter folded into the seed) of the LAST training forward, one per hidden layer:
        what a test needs to rebuild the keep masks with ``_tower_ops.dropout_mask``."""
        return [d.resolved() if d is not None else None for d in self._last_drops]

    @property
    def moving_mean(self):
        return [getattr(self, 'moving_mean_%d' % i) for i in range(len(self.hidden_layer_dims))]

    @property
    def moving_var(self):
        return [getattr(self, 'moving_var_%d' % i) for i in range(len(self.hidden_layer_dims))]

    def forward(self, x: torch.Tensor, row_index: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``row_index`` (int [M]): score row ``row_index[m]`` of ``x`` at position m -- the gather of
        ``FlattenList``'s circular padding, fused into the input cast."""
        lead = None
        if x.dim() > 2 and row_index is None:              # Keras Dense / BatchNormalization act on the last axis:
            lead = tuple(x.shape[:-1])                     # [..., F] is [prod(...), F] (keras/layers_test.py:25-30)
            x = x.reshape(-1, x.shape[-1])
        if x.dim() != 2 or x.shape[1] not in (self.input_dim, T.pad8(self.input_dim), T.pad_k(self.input_dim)):
            raise ValueError('expected [M, %d] features, got %s' % (self.input_dim, tuple(x.shape)))
        params = list(self.weights) + list(self.biases)
        if self.use_batch_norm:
            params += list(self.gammas) + list(self.betas)
        if self.input_batch_norm:
            if x.dtype not in (torch.float32, torch.bfloat16) or x.shape[1] != self.input_dim:
                raise ValueError('input_batch_norm needs the raw fp32 (or bf16-ingested) [M, %d] features'
                                 % self.input_dim)
            params += [self.gamma_in, self.beta_in]
        params += [self.out_weight, self.out_bias]
        out = _TowerFn.apply(x, self, self.training, row_index, *params)
        return out if lead is None else out.reshape(lead + (out.shape[-1],))
