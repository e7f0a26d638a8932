"""Dense layer of ``create_tower`` outside the fused bf16 tower (ranking_amd/tower.py): any width, any output_units.

``make_dense(in, out, dtype)`` returns the Dense layer of keras/layers.py:26-77:

* ``torch.float32`` -- the reference's own precision: fp32 operands, fp32 accumulation on the matrix cores
  (``v_mfma_f32_32x32x2_f32``, csrc/gemm_f32.hip), forward, input gradient, weight gradient and bias gradient all
  on hand-written kernels behind the C ABI (``tfr_tower_gemm_f32``, ``tfr_tower_colsum_f32``);
* ``torch.bfloat16`` -- shapes the fused tower does not take (hidden widths that are not multiples of 8,
  output_units > 4): operands rounded to bf16, products and sums in fp32 -- the same arithmetic as a bf16 MFMA with
  fp32 accumulation (a product of two bf16 numbers is exact in fp32) -- through the same kernel.

Weights stay fp32 ``nn.Parameter``s (master copy) and the gradients are ordinary ``.grad`` tensors so that the
data-parallel all-reduce (ranking_amd.distributed) sees one flat fp32 bucket.  Tensors that live on the host (the CPU
unit tests of the module structure) go through ``torch.nn.functional.linear``; a device tensor never does, and a
missing HIP library raises.
"""
from __future__ import annotations

import torch
from torch import nn


class _DenseF32Fn(torch.autograd.Function):
    """y = x . W^T + b on device tensors; backward: dx = dy . W, dW = dy^T . x (slabs over the rows, fixed-order sum),
    db = column sums of dy."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        from . import _tower_ops as T
        x2 = x.reshape(-1, x.shape[-1]).to(torch.float32)
        w = weight.detach().to(torch.float32)
        y = T.dense_f32(x2, w, bias)
        ctx.save_for_backward(x2, w)
        ctx.has_bias = bias is not None
        ctx.x_shape = x.shape
        return y.reshape(tuple(x.shape[:-1]) + (w.shape[0],))

    @staticmethod
    def backward(ctx, dy):
        from . import _tower_ops as T
        x2, w = ctx.saved_tensors
        dy2 = dy.reshape(-1, w.shape[0]).to(torch.float32).contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = T.dense_f32_dgrad(dy2, w).reshape(ctx.x_shape)
        if ctx.needs_input_grad[1]:
            dw = T.dense_f32_wgrad(dy2, x2)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = T.colsum_f32(dy2)
        return dx, dw, db


class DenseF32(nn.Linear):
    """fp32 Dense (keras/layers.py:62,71): device tensors run csrc/gemm_f32.hip."""

    def forward(self, x):
        if not x.is_cuda:
            return torch.nn.functional.linear(x, self.weight, self.bias)
        return _DenseF32Fn.apply(x, self.weight, self.bias)


class _RoundBf16Fn(torch.autograd.Function):
    """x -> bf16(x) as fp32, straight-through gradient (the operand rounding of a bf16 GEMM)."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g


class DenseBf16(nn.Linear):
    """y = bf16(x) . bf16(W)^T + b with fp32 products and sums; fp32 output so that BatchNorm statistics stay fp32."""

    def forward(self, x):
        xr, wr = _RoundBf16Fn.apply(x.to(torch.float32)), _RoundBf16Fn.apply(self.weight)
        if not x.is_cuda:
            return torch.nn.functional.linear(xr, wr, self.bias)
        return _DenseF32Fn.apply(xr, wr, self.bias)


def make_dense(in_features: int, out_features: int, compute_dtype=torch.float32) -> nn.Module:
    if compute_dtype == torch.bfloat16:
        layer = DenseBf16(in_features, out_features)
    elif compute_dtype == torch.float32:
        layer = DenseF32(in_features, out_features)
    else:
        raise ValueError('compute_dtype must be torch.float32 or torch.bfloat16')
    # Keras Dense default init: glorot_uniform kernel, zero bias.
    nn.init.xavier_uniform_(layer.weight)
    nn.init.zeros_(layer.bias)
    return layer
