"""Scorer GEMM building block.

``make_dense(in, out, dtype)`` returns the Dense layer used by ``create_tower``:
fp32 (the reference's precision) or bf16-operand / fp32-accumulate on the MFMA
units (BASELINE.json config 2).  Weights are kept in fp32 (master copy) and the
gradient buffers are ordinary ``.grad`` tensors so that the data-parallel
all-reduce (ranking_amd.distributed) sees one flat fp32 bucket.
"""
from __future__ import annotations

import torch
from torch import nn


class DenseBf16(nn.Linear):
    """y = x @ W^T + b with bf16 operands and fp32 accumulation (MFMA via hipBLASLt);
    output is returned in fp32 so that BatchNorm statistics stay in fp32."""

    def forward(self, x):
        y = torch.nn.functional.linear(x.to(torch.bfloat16), self.weight.to(torch.bfloat16), None)
        return y.to(torch.float32) + self.bias


def make_dense(in_features: int, out_features: int, compute_dtype=torch.float32) -> nn.Module:
    if compute_dtype == torch.bfloat16:
        layer = DenseBf16(in_features, out_features)
    elif compute_dtype == torch.float32:
        layer = nn.Linear(in_features, out_features)
    else:
        raise ValueError('compute_dtype must be torch.float32 or torch.bfloat16')
    # Keras Dense default init: glorot_uniform kernel, zero bias.
    nn.init.xavier_uniform_(layer.weight)
    nn.init.zeros_(layer.bias)
    return layer
