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


def broadcast_module(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Makes every replica start from rank ``src``'s parameters and buffers (what `tf.distribute` mirrored variables
    do from a single initialisation): ONE broadcast of one flat fp32 buffer, then a scatter back.  Without it the
    ranks would average gradients of differently initialised models."""
    _, w = world()
    if w <= 1:
        return
    tensors = [p.data for p in module.parameters()] + [b.data for b in module.buffers()]
    tensors = [t for t in tensors if t.numel()]
    if not tensors:
        return
    flat = torch.cat([t.reshape(-1).to(torch.float32) for t in tensors])
    dist.broadcast(flat, src=src, group=group)
    off = 0
    for t in tensors:
        t.copy_(flat[off:off + t.numel()].view_as(t).to(t.dtype))
        off += t.numel()


def all_reduce_scalars(values: Sequence, device, average: bool = False, group=None) -> List[float]:
    """Sums (or averages) a few python / tensor scalars over the ranks with one collective -- the validation
    loss, early-stopping monitors: every rank must take the same decision or the next all-reduce hangs."""
    _, w = world()
    buf = torch.stack([torch.as_tensor(v, dtype=torch.float64, device=device).reshape(()) for v in values])
    if w > 1:
        dist.all_reduce(buf, group=group)
        if average:
            buf /= w
    return [float(x) for x in buf.tolist()]


def completion_order(module: torch.nn.Module, split: int):
    """(params in the order their gradients become final in a backward, number of elements of the EARLY part).
    Early = the output layer and the hidden layers >= `split` of every FusedTower in `module` (what
    FusedTower.grad_split_hook announces); late = everything else (hidden layers < split, parameters outside a fused
    tower).  A FlatGradBucket built on this order holds the early gradients as one contiguous prefix: their all-reduce can
    run while the rest of the backward computes."""
    from .tower import FusedTower
    early, late, seen = [], [], set()
    for m in module.modules():
        if not isinstance(m, FusedTower):
            continue
        n_h = len(m.hidden_layer_dims)
        if not 1 <= split < n_h:
            raise ValueError('split must be in [1, %d), got %d' % (n_h, split))
        for l in range(n_h - 1, -1, -1):
            group = [m.weights[l], m.biases[l]] + ([m.gammas[l], m.betas[l]] if m.use_batch_norm else [])
            (early if l >= split else late).extend(group)
        early[0:0] = [m.out_weight, m.out_bias]
        if m.input_batch_norm:
            late.extend([m.gamma_in, m.beta_in])
    for p in early + late:
        seen.add(id(p))
    late.extend(p for p in module.parameters() if id(p) not in seen)
    early = [p for p in early if p.requires_grad]
    late = [p for p in late if p.requires_grad]
    return early + late, sum(p.numel() for p in early)


class FlatGradBucket:
    """Re-homes every parameter's ``.grad`` into one contiguous fp32 buffer with
    ``n_scalars`` extra slots at its end, so a step needs exactly one all-reduce."""

    def __init__(self, params: Iterable[torch.nn.Parameter], n_scalars: int = 2, group=None,
                 flatten_params: bool = False):
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
        # Optionally the fp32 master weights move into ONE flat buffer too (same order as the gradients): a plain
        # SGD step is then a single fused multiply-add over the buffer instead of a multi-tensor launch over ~14
        # small tensors (33 us -> 3 us for the 136-512-512-512-1 scorer), and a checkpoint is one tensor.
        self.flat_params = None
        if flatten_params:
            self.flat_params = torch.empty(self.numel, dtype=torch.float32, device=dev)
            off = 0
            with torch.no_grad():
                for p in self.params:
                    view = self.flat_params[off:off + p.numel()].view_as(p)
                    view.copy_(p.data)
                    p.data = view
                    off += p.numel()

    def sgd_step(self, lr: float) -> None:
        """params -= lr * grads on the flat buffers (needs ``flatten_params=True``): one launch."""
        if self.flat_params is None:
            raise ValueError('sgd_step needs FlatGradBucket(..., flatten_params=True)')
        self.flat_params.add_(self.flat[:self.numel], alpha=-float(lr))

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
        _, w = world()
        if w <= 1:                                   # a single replica: nothing to exchange, nothing to launch
            return self.scalars.clone() if scalars is None else scalars.reshape(-1)
        if scalars is not None:
            self.scalars.copy_(scalars.reshape(-1))
        dist.all_reduce(self.flat, group=self.group)
        if average:
            self.flat[:self.numel].div_(w)
        return self.scalars.clone()

    def scale_grads(self, factor):
        self.flat[:self.numel].mul_(factor)

    def all_reduce_range(self, lo: int, hi: int, average: bool = False) -> None:
        """The collective of one contiguous part of the bucket, on the CURRENT stream (a view of the flat buffer: no copy)."""
        _, w = world()
        if w <= 1 or hi <= lo:
            return
        part = self.flat[lo:hi]
        dist.all_reduce(part, group=self.group)
        if average:
            part.div_(w)


class SplitStep:
    """A data-parallel training step whose gradient exchange overlaps the backward (round 6).

        zero | forward | loss | backward of the output layer and the hidden layers >= split        <- part A (one hipGraph)
        all-reduce of the early gradients on a SIDE stream  ||  backward of the layers < split     <- part B (one hipGraph)
        all-reduce of the late gradients + the step's scalars | wait for the side stream | optimizer  (one hipGraph)

    The cut inside the backward is FusedTower.grad_split_hook: during capture the hook ends graph A and begins graph B (same
    memory pool: B consumes what A produced).  At world size 1 no collective is issued and the three graphs replay back to
    back -- the same launches in the same order as the single-graph step: bit-identical parameters.  The reference's
    MirroredStrategy overlaps per-variable all-reduces with the backward (keras/strategy_utils.py:87-116); here there are
    exactly two collectives per step and the second one is 0.3 MB.

    `fwd_bwd()` -> 0-d loss value: zeroes nothing itself (the bucket is zeroed here), runs forward, loss and
    `logits.backward(...)`; `optimizer()` applies the update.  `scalars(value)` -> 1-D tensor of per-rank scalars summed
    with the late part (default: [value, 1]).  `capture` = the context manager to capture with (bench.capture)."""

    def __init__(self, module, bucket: 'FlatGradBucket', early_numel: int, split: int, fwd_bwd, optimizer, average=True,
                 scalars=None, use_graph=True, capture=None, graph_generators=()):
        from .tower import FusedTower
        self.bucket, self.early, self.average = bucket, int(early_numel), average
        self.towers = [m for m in module.modules() if isinstance(m, FusedTower)]
        if len(self.towers) != 1:
            raise ValueError('SplitStep needs exactly one FusedTower in the module, found %d' % len(self.towers))
        self.tower = self.towers[0]
        if not self.tower.accumulate_grads_in_place:
            raise ValueError('SplitStep needs FlatGradBucket.attach(module) (gradients accumulated in place)')
        self.tower.grad_split = int(split)
        self.fwd_bwd, self.optimizer = fwd_bwd, optimizer
        self.make_scalars = scalars or (lambda v: torch.stack([v, torch.ones_like(v)]))
        self.side = torch.cuda.Stream()
        self.use_graph = use_graph
        self.value = None
        self.static_scalars = None
        self._hook_calls = 0
        if use_graph:
            self._capture(capture, graph_generators)

    # -- eager form (also the warm-up of the capture)
    def _eager(self):
        _, w = world()
        main = torch.cuda.current_stream()
        ev = {}

        def hook(_tower):
            self._hook_calls += 1
            if w > 1:
                self.side.wait_stream(main)
                with torch.cuda.stream(self.side):
                    self.bucket.all_reduce_range(0, self.early, self.average)
        self.tower.grad_split_hook = hook
        try:
            self.bucket.zero()
            value = self.fwd_bwd()
        finally:
            self.tower.grad_split_hook = None
        s = self._late(self.make_scalars(value))
        main.wait_stream(self.side)
        self.optimizer()
        return s

    def _late(self, scalars):
        _, w = world()
        b = self.bucket
        if w <= 1:
            return scalars.reshape(-1)
        b.scalars.copy_(scalars.reshape(-1))
        b.all_reduce_range(self.early, b.flat.numel(), False)
        if self.average:
            b.flat[self.early:b.numel].div_(w)
        return b.scalars.clone()

    def _capture(self, capture, graph_generators):
        main = torch.cuda.current_stream()
        warm = torch.cuda.Stream()
        warm.wait_stream(main)
        with torch.cuda.stream(warm):
            for _ in range(3):
                self._eager()
        main.wait_stream(warm)
        torch.cuda.synchronize()
        self.gA, self.gB, self.gO = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        for gen in graph_generators:
            self.gA.register_generator_state(gen)
        _, w = world()
        # 'relaxed': graph A begins on this thread and ends on the autograd engine's device thread (where the backward -- and
        # the split hook -- runs), graph B the other way round; the stricter modes tie a capture to the thread that began it
        mode = 'relaxed'
        cap_stream = torch.cuda.Stream()
        cap_stream.wait_stream(main)
        state = {'in_b': False}

        def hook(_tower):                                   # ends graph A, begins graph B: in the middle of the backward
            self._hook_calls += 1
            self.gA.capture_end()
            self.gB.capture_begin(pool=self.gA.pool(), capture_error_mode=mode)
            state['in_b'] = True
        self.tower.grad_split_hook = hook
        torch.cuda.synchronize()
        with torch.cuda.stream(cap_stream):
            self.gA.capture_begin(capture_error_mode=mode)
            try:
                self.bucket.zero()
                self.value = self.fwd_bwd()
                self.static_scalars = self.make_scalars(self.value)
            finally:
                self.tower.grad_split_hook = None
                (self.gB if state['in_b'] else self.gA).capture_end()
            if not state['in_b']:
                raise RuntimeError('SplitStep: the backward never reached the split (is grad_split inside the tower?)')
            self.gO.capture_begin(pool=self.gA.pool(), capture_error_mode=mode)
            self.optimizer()
            self.gO.capture_end()
        main.wait_stream(cap_stream)
        torch.cuda.synchronize()

    def __call__(self):
        """one step; returns the (globally summed) scalars"""
        if not self.use_graph:
            return self._eager()
        _, w = world()
        main = torch.cuda.current_stream()
        self.gA.replay()
        if w > 1:
            self.side.wait_stream(main)
            with torch.cuda.stream(self.side):
                self.bucket.all_reduce_range(0, self.early, self.average)
        self.gB.replay()
        s = self._late(self.static_scalars)
        if w > 1:
            main.wait_stream(self.side)
        self.gO.replay()
        return s

    def compute_only(self):
        """the same three graphs without the collectives (what the exposed time of the exchange is measured against)"""
        self.gA.replay(); self.gB.replay(); self.gO.replay()


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
