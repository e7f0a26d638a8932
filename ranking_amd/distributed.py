"""Data parallelism for the ranking hot path.

The reference only selects a ``tf.distribute`` strategy
(keras/strategy_utils.py:45-116) and lets TensorFlow run the collectives.  Here:
one process per GPU, query lists sharded across ranks, and ONE RCCL all-reduce
per step over a single flat fp32 bucket that holds every scorer gradient plus
the few per-rank scalars whose GLOBAL value the reductions need (loss numerator /
denominator for SUM_BY_NONZERO_WEIGHTS / MEAN, metric sums).  The scorer
gradient is tiny (2.4 MB for 136-512-512-512-1), i.e. latency bound on xGMI, so
a single bucket and a single collective is the right shape -- no per-tensor
calls, no bucketing heuristics.  The loss / metric kernels themselves shard
with no data-path collective at all.
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_bounds(num_lists: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Rank r owns lists [r*B/W, (r+1)*B/W) (remainder spread over the first ranks)."""
    q, r = divmod(num_lists, world_size)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_lists(tensors: Sequence[torch.Tensor], rank: Optional[int] = None,
                world_size: Optional[int] = None) -> List[torch.Tensor]:
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    lo, hi = shard_bounds(tensors[0].shape[0], rank, world_size)
    return [t[lo:hi] for t in tensors]


class FlatGradBucket:
    """Re-homes every parameter's ``.grad`` into one contiguous fp32 buffer with
    ``n_scalars`` extra slots at its end, so a step needs exactly one all-reduce."""

    def __init__(self, params: Iterable[torch.nn.Parameter], n_scalars: int = 2, group=None):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError('no trainable parameters')
        dev = self.params[0].device
        self.n_scalars = n_scalars
        self.numel = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(self.numel + n_scalars, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            if p.dtype != torch.float32:
                raise ValueError('master parameters must be fp32')
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        self.group = group

    def attach(self, module: torch.nn.Module) -> 'FlatGradBucket':
        """Lets the fused towers of ``module`` accumulate straight into this bucket's views (one multi-tensor
        launch + the weight-gradient split reductions) instead of one autograd ``grad += g`` launch per
        parameter.  Valid because the bucket keeps every ``.grad`` allocated; parameter hooks do not fire."""
        from .tower import FusedTower
        for m in module.modules():
            if isinstance(m, FusedTower):
                m.accumulate_grads_in_place = True
        return self

    @property
    def scalars(self) -> torch.Tensor:
        return self.flat[self.numel:]

    def zero(self):
        self.flat.zero_()

    def all_reduce(self, scalars: Optional[torch.Tensor] = None, average: bool = False) -> torch.Tensor:
        """Sums gradients (and the appended scalars) over ranks with one collective;
        returns the globally summed scalars.  ``average`` divides gradients by W."""
        if scalars is not None:
            self.scalars.copy_(scalars.reshape(-1))
        _, w = world()
        if w > 1:
            dist.all_reduce(self.flat, group=self.group)
        if average and w > 1:
            self.flat[:self.numel].div_(w)
        return self.scalars.clone()

    def scale_grads(self, factor):
        self.flat[:self.numel].mul_(factor)


def global_normalizer_step(bucket: FlatGradBucket, local_numerator: torch.Tensor,
                           local_denominator: torch.Tensor) -> torch.Tensor:
    """For reductions that divide by a GLOBAL quantity (Keras AUTO: global batch
    size; estimator SUM_BY_NONZERO_WEIGHTS: global count of non-zero weights):
    each rank back-propagates its un-normalised numerator, then one all-reduce
    carries gradients + (numerator, denominator); gradients are divided by the
    global denominator afterwards.  Returns the global loss."""
    s = bucket.all_reduce(torch.stack([local_numerator.detach().reshape(()),
                                       local_denominator.detach().reshape(()).to(torch.float32)]))
    den = s[1]
    inv = torch.where(den != 0, 1.0 / torch.where(den != 0, den, torch.ones_like(den)),
                      torch.zeros_like(den))
    bucket.scale_grads(inv)
    return s[0] * inv
