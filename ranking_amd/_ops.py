"""Torch-tensor level bindings of the C ABI (include/tfr_hip.h).

PyTorch is used here for device memory and streams only: every function
checks that its tensors live on a HIP device, enqueues ONE kernel on torch's
current stream and returns freshly allocated outputs.  No CPU fallback.
"""
from __future__ import annotations

import ctypes
import functools
import os
from collections import OrderedDict
from typing import Callable, Dict, Optional, Sequence, Tuple

import torch

from . import _lib

GAIN_IDENTITY, GAIN_POW2M1, GAIN_CUSTOM = 0, 1, 2
LAMBDA_NONE, LAMBDA_LABELDIFF, LAMBDA_DCG = 0, 1, 2
LAMBDA_DCG_V2, LAMBDA_YETI_DCG, LAMBDA_PRECISION = 3, 4, 5
PAIR_LOGISTIC, PAIR_HINGE, PAIR_SOFT_ZERO_ONE, PAIR_MSE = 0, 1, 2, 3
PAIR_TIED_ZERO = 0x100          # | PAIR_LOGISTIC: the reference's (TF autodiff) zero gradient at exactly tied scores (tfr_hip.h)
MAX_TOPN = 8


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_device(t: torch.Tensor, name: str) -> torch.Tensor:
    if not torch.is_tensor(t):
        raise TypeError('%s must be a torch.Tensor' % name)
    if not t.is_cuda:
        raise _lib.TfrHipError(
            '%s is on %s: ranking_amd runs on MI355X only (no CPU fallback). Move the tensor to '
            'a HIP device.' % (name, t.device))
    return t


def _f32(t, name):
    if t is None:
        return None
    require_device(t, name)
    if t.dtype != torch.float32:
        t = t.to(torch.float32)
    return t.contiguous()


def _u8(t, name):
    if t is None:
        return None
    require_device(t, name)
    if t.dtype == torch.bool:                       # same storage (one byte, 0 / 1): no conversion launch
        return t.contiguous().view(torch.uint8)
    if t.dtype != torch.uint8:
        t = t.to(torch.uint8)
    return t.contiguous()


def _check2d(t, name):
    if t.dim() != 2:
        raise ValueError('%s must have rank 2, got shape %s' % (name, tuple(t.shape)))


def _same_shape(a, b, na, nb):
    if a.shape != b.shape:
        raise ValueError('%s %s is not compatible with %s %s' % (na, tuple(a.shape), nb, tuple(b.shape)))


# ------------------------------------------------------------------ tables
class DeviceConstCache:
    """Small read-only device tensors (rank tables, per-list scale vectors) keyed by value.

    A hipGraph bakes the ADDRESS of every tensor a captured launch reads, so an entry a capture has seen must
    outlive every replay: entries touched while the current stream is capturing are pinned for the life of the
    process, only never-captured entries are evicted (least recently used first) once `capacity` is exceeded.
    An entry that would have to be CREATED during a capture is not cached at all -- its storage belongs to the
    graph's private pool and the fill is a node of the graph."""

    def __init__(self, capacity: int):
        self.capacity = int(capacity)
        self._lru: 'OrderedDict[Tuple, torch.Tensor]' = OrderedDict()
        self._pinned: Dict[Tuple, torch.Tensor] = {}

    def __len__(self):
        return len(self._lru) + len(self._pinned)

    def pinned(self) -> int:
        return len(self._pinned)

    def get(self, key, make: Callable[[], torch.Tensor]) -> torch.Tensor:
        capturing = torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
        t = self._pinned.get(key)
        if t is not None:
            return t
        t = self._lru.get(key)
        if t is not None:
            if capturing:
                self._pinned[key] = self._lru.pop(key)
            else:
                self._lru.move_to_end(key)
            return t
        t = make()
        if capturing:
            return t
        self._lru[key] = t
        while len(self._lru) > self.capacity:
            self._lru.popitem(last=False)
        return t


_table_cache = DeviceConstCache(256)


def rank_table(fn: Callable, n: int, device) -> torch.Tensor:
    """fp32 table fn(r), r = 1..n, evaluated ONCE on the host (so that it is
    bit-identical to what the CPU oracle uses) and cached on the device."""
    def make():
        r = torch.arange(1, n + 1, dtype=torch.float32)
        v = fn(r)
        if not torch.is_tensor(v):
            v = torch.as_tensor(v, dtype=torch.float32)
        v = torch.broadcast_to(v.to(torch.float32), r.shape).contiguous()
        return v.to(device)
    return _table_cache.get((fn, int(n), str(device)), make)


def _inv_log1p(rank):
    return 1.0 / torch.log1p(rank)


# -------------------------------------------------------------------- sort
def sort_ranks(scores, labels=None, mask=None, tiebreak=None, want_ranks=True, want_order=True):
    scores = _f32(scores, 'scores'); _check2d(scores, 'scores')
    labels = _f32(labels, 'labels'); mask = _u8(mask, 'mask')
    if tiebreak is not None:
        tiebreak = require_device(tiebreak, 'tiebreak').to(torch.int32).contiguous()
    B, L = scores.shape
    ranks = torch.empty((B, L), dtype=torch.int32, device=scores.device) if want_ranks else None
    order = torch.empty((B, L), dtype=torch.int32, device=scores.device) if want_order else None
    rc = _lib.load().tfr_sort_ranks_f32(_ptr(scores), _ptr(labels), _ptr(mask), _ptr(tiebreak), B, L,
                                        _ptr(ranks), _ptr(order), _stream())
    _lib.check(rc, 'tfr_sort_ranks_f32')
    return ranks, order


def _topn_array(topns: Sequence[Optional[int]]):
    if not 1 <= len(topns) <= MAX_TOPN:
        raise ValueError('between 1 and %d cutoffs per call' % MAX_TOPN)
    arr = (ctypes.c_int32 * len(topns))(*[0 if t is None else int(t) for t in topns])
    return arr


def _weights_arg(weights, labels):
    """Returns (tensor or None, per_list flag) for [B,L] / [B,1] / [B] / scalar weights."""
    if weights is None:
        return None, 0
    if not torch.is_tensor(weights):
        weights = torch.as_tensor(weights, dtype=torch.float32, device=labels.device)
    weights = _f32(weights, 'weights')
    B, L = labels.shape
    if weights.dim() == 0:
        return weights.expand(B).contiguous(), 1
    if weights.dim() == 1 and weights.shape[0] == B:
        return weights, 1
    if weights.dim() == 2 and weights.shape == (B, 1):
        return weights.reshape(B).contiguous(), 1
    if weights.dim() == 2 and weights.shape == (B, L):
        return weights, 0
    raise ValueError('weights shape %s is not compatible with labels %s'
                     % (tuple(weights.shape), (B, L)))


def ndcg_metric(labels, predictions, weights, mask, gains, discount, topns, tie_seed=0):
    """tie_seed (here and in the other metric ops): 0 = equal predictions in index order; != 0 = in the hashed order of
    `tie_keys` (the reference's shuffle_ties; NDCG then runs on its sort kernel)."""
    labels = _f32(labels, 'labels'); predictions = _f32(predictions, 'predictions')
    _check2d(predictions, 'predictions'); _same_shape(labels, predictions, 'labels', 'predictions')
    w, per_list = _weights_arg(weights, labels)
    mask = _u8(mask, 'mask'); gains = _f32(gains, 'gains'); discount = _f32(discount, 'discount')
    B, L = labels.shape
    K = len(topns)
    out = torch.empty((K, B), dtype=torch.float32, device=labels.device)
    stats = torch.empty((B, 3), dtype=torch.float32, device=labels.device)
    rc = _lib.load().tfr_ndcg_metric_f32(_ptr(labels), _ptr(predictions), _ptr(w), per_list, _ptr(mask),
                                         _ptr(gains), _ptr(discount), _topn_array(topns), K, B, L,
                                         _ptr(out), _ptr(stats), int(tie_seed) & 0xffffffff, _stream())
    _lib.check(rc, 'tfr_ndcg_metric_f32')
    return out, stats


METRIC_NDCG, METRIC_MRR, METRIC_DCG, METRIC_HITS, METRIC_RECALL, METRIC_PRECISION, METRIC_MAP, METRIC_ARP = range(8)
METRIC_BPREF, METRIC_BPREF_NONTREC, METRIC_PWA, METRIC_OPA = 8, 9, 10, 11


WS_LIST_MLE, WS_UNIQUE_SOFTMAX, WS_CIRCLE, WS_RANK_METRIC, WS_DIV_METRIC, WS_NEURAL_SORT_NDCG, WS_NEURAL_SORT_CE = range(7)
_WS_LISTS = 256                          # lists in flight of a launch that runs from the workspace


def _workspace(op, B, L, device):
    """(tensor or None, bytes): the workspace an entry point needs once a list's working arrays outgrow LDS
    (include/tfr_hip.h tfr_list_workspace_bytes): min(B, 256) slots.  The tensor must outlive the launch -- it does, as
    a local of the calling op (the caching allocator hands the block out again in stream order)."""
    slot = int(_lib.load().tfr_list_workspace_bytes(int(op), int(L)))
    if slot <= 0 or B <= 0:
        return None, 0
    n = slot * min(int(B), _WS_LISTS)
    return torch.empty((n,), dtype=torch.uint8, device=device), n


def rank_metric(kind, labels, predictions, weights, mask, topns, gains=None, discount=None, tie_seed=0):
    """tfr_rank_metric_f32: ([K, B] metric, [B, 3] stats) for any sort-based metric kind."""
    labels = _f32(labels, 'labels'); predictions = _f32(predictions, 'predictions')
    _check2d(predictions, 'predictions'); _same_shape(labels, predictions, 'labels', 'predictions')
    w, per_list = _weights_arg(weights, labels)
    mask = _u8(mask, 'mask'); gains = _f32(gains, 'gains'); discount = _f32(discount, 'discount')
    B, L = labels.shape
    K = len(topns)
    out = torch.empty((K, B), dtype=torch.float32, device=labels.device)
    stats = torch.empty((B, 3), dtype=torch.float32, device=labels.device)
    ws, ws_bytes = (None, 0) if kind in (METRIC_NDCG, METRIC_MRR) else _workspace(WS_RANK_METRIC, B, L, labels.device)
    rc = _lib.load().tfr_rank_metric_f32(int(kind), _ptr(labels), _ptr(predictions), _ptr(w), per_list, _ptr(mask),
                                         _ptr(gains), _ptr(discount), _topn_array(topns), K, B, L,
                                         _ptr(out), _ptr(stats), int(tie_seed) & 0xffffffff, _ptr(ws), ws_bytes, _stream())
    _lib.check(rc, 'tfr_rank_metric_f32')
    return out, stats


def metric_list_weights(stats):
    """tfr_metric_list_weights_f32: [B, 3] per-list statistics -> [B, 1] per-list metric weights."""
    stats = _f32(stats, 'stats')
    B = stats.shape[0]
    out = torch.empty((B, 1), dtype=torch.float32, device=stats.device)
    _lib.check(_lib.load().tfr_metric_list_weights_f32(_ptr(stats), B, _ptr(out), _stream()),
               'tfr_metric_list_weights_f32')
    return out


DIV_ALPHA_DCG, DIV_PRECISION_IA = 0, 1


def div_metric(kind, labels, predictions, weights, mask, topns, discount=None, alpha=0.5, tie_seed=0):
    """tfr_div_metric_f32: ([K, B] metric, [B, 3] stats) on subtopic labels [B, L, S]."""
    labels = _f32(labels, 'labels'); predictions = _f32(predictions, 'predictions')
    _check2d(predictions, 'predictions')
    if labels.dim() != 3 or tuple(labels.shape[:2]) != tuple(predictions.shape):
        raise ValueError('labels %s must be [batch_size, list_size, subtopic_size] matching predictions %s'
                         % (tuple(labels.shape), tuple(predictions.shape)))
    w, per_list = _weights_arg(weights, predictions)
    mask = _u8(mask, 'mask'); discount = _f32(discount, 'discount')
    B, L, S = labels.shape
    K = len(topns)
    out = torch.empty((K, B), dtype=torch.float32, device=labels.device)
    stats = torch.empty((B, 3), dtype=torch.float32, device=labels.device)
    ws, ws_bytes = _workspace(WS_DIV_METRIC, B, L, labels.device)
    rc = _lib.load().tfr_div_metric_f32(int(kind), _ptr(labels), _ptr(predictions), _ptr(w), per_list, _ptr(mask),
                                        _ptr(discount), float(alpha), _topn_array(topns), K, B, L, S,
                                        _ptr(out), _ptr(stats), int(tie_seed) & 0xffffffff, _ptr(ws), ws_bytes, _stream())
    _lib.check(rc, 'tfr_div_metric_f32')
    return out, stats


def mrr_metric(labels, predictions, weights, mask, topns, tie_seed=0):
    labels = _f32(labels, 'labels'); predictions = _f32(predictions, 'predictions')
    _check2d(predictions, 'predictions'); _same_shape(labels, predictions, 'labels', 'predictions')
    w, per_list = _weights_arg(weights, labels)
    mask = _u8(mask, 'mask')
    B, L = labels.shape
    K = len(topns)
    out = torch.empty((K, B), dtype=torch.float32, device=labels.device)
    stats = torch.empty((B, 3), dtype=torch.float32, device=labels.device)
    rc = _lib.load().tfr_mrr_metric_f32(_ptr(labels), _ptr(predictions), _ptr(w), per_list, _ptr(mask),
                                        _topn_array(topns), K, B, L, _ptr(out), _ptr(stats), int(tie_seed) & 0xffffffff,
                                        _stream())
    _lib.check(rc, 'tfr_mrr_metric_f32')
    return out, stats


# ------------------------------------------------------------------ losses
_BALANCE_MIN_LISTS = 1024      # below this the ordering launch costs more than the tail it removes


# Persistent zero-initialised int32 scratch of the kernels that synchronise their workgroups through device memory
# (last-workgroup tickets, class counters) and leave it zero.  One POOL per device (never freed: captured hipGraphs hold
# addresses inside it), cut into slots; a launch gets a slot no launch that may overlap it shares:
#   * eager launches: one slot per (kind, current stream) -- launches on one stream run in order;
#   * launches recorded by a stream capture run wherever and whenever the graph is replayed, so EVERY recorded launch takes a
#     fresh slot of its own (one state per kind shared by all graphs let two graphs replayed on different streams
#     corrupt each other's tickets).  The pool is created by the first eager call on the device (every capture is preceded
#     by an eager run of the step); a capture that finds no pool, or an exhausted one, gets None and the caller falls back
#     (a private zero-filled tensor from the graph's own pool -- the fill becomes a node of the graph -- or a path that
#     needs no state).
_STATE_SLOT_INTS = 256
_STATE_SLOTS = 2048
_state_pools: Dict[str, dict] = {}


def _state_slot(kind: str, n_ints: int, device) -> Optional[torch.Tensor]:
    if n_ints > _STATE_SLOT_INTS:
        raise ValueError('state of %d ints does not fit a slot' % n_ints)
    capturing = torch.cuda.is_current_stream_capturing()
    pool = _state_pools.get(str(device))
    if pool is None:
        if capturing:
            return None
        pool = {'buf': torch.zeros((_STATE_SLOTS, _STATE_SLOT_INTS), dtype=torch.int32, device=device), 'next': 0, 'eager': {}}
        _state_pools[str(device)] = pool
    if capturing:
        key = None
    else:
        key = (kind, int(torch.cuda.current_stream(device).cuda_stream))
        if key in pool['eager']:
            return pool['buf'][pool['eager'][key], :n_ints]
    if pool['next'] >= _STATE_SLOTS:
        return None
    i = pool['next']
    pool['next'] = i + 1
    if key is not None:
        pool['eager'][key] = i
    return pool['buf'][i, :n_ints]


def _zero_state(kind: str, n_ints: int, device) -> torch.Tensor:
    """The state of one launch (see above); when the pool cannot serve it, a zero-filled tensor of its own."""
    t = _state_slot(kind, n_ints, device)
    return t if t is not None else torch.zeros(n_ints, dtype=torch.int32, device=device)


def _sum_outputs(device):
    """(0-d result, ticket state) of a `*_sum_f32` launch.  Eagerly the result starts as NaN: if the last-arriver logic ever
    misfired (ticket state left dirty by an aborted launch) the reduced loss is loudly wrong instead of silently stale;
    inside a capture it is plain graph-pool memory (a fill would be one more node in every replay)."""
    if torch.cuda.is_current_stream_capturing():
        total = torch.empty((), dtype=torch.float32, device=device)
    else:
        total = torch.full((), float('nan'), dtype=torch.float32, device=device)
    return total, _zero_state('loss_sum', int(_lib.load().tfr_grid_sum_state_ints()), device)


def list_order(labels, mask=None):
    """tfr_list_order_i32: int32 [B] list indices, longest valid length first (launch order of the
    O(n^2) loss kernels; their results do not depend on it)."""
    labels = _f32(labels, 'labels'); mask = _u8(mask, 'mask')
    B, L = labels.shape
    order = torch.empty((B,), dtype=torch.int32, device=labels.device)
    ws = torch.empty((B,), dtype=torch.int32, device=labels.device)
    lib = _lib.load()
    rc = lib.tfr_list_order_i32(_ptr(labels), _ptr(mask), B, L, _ptr(order), _ptr(ws), _stream())
    _lib.check(rc, 'tfr_list_order_i32')
    return order


_ORDER_INTERLEAVED = os.environ.get('TFR_ORDER_INTERLEAVED', '1') != '0'


def launch_order_interleaved(labels, mask=None):
    """tfr_list_order_interleaved_i32: an approximately longest-first permutation from ONE launch (every workgroup's
    lists sorted by length class, interleaved with the other workgroups') -- what the O(n^2) losses use for load balance
    since round 6 (TFR_ORDER_INTERLEAVED=0: the exact two-launch order of ``list_order``)."""
    labels = _f32(labels, 'labels'); mask = _u8(mask, 'mask')
    B, L = labels.shape
    order = torch.empty((B,), dtype=torch.int32, device=labels.device)
    rc = _lib.load().tfr_list_order_interleaved_i32(_ptr(labels), _ptr(mask), B, L, _ptr(order), _stream())
    _lib.check(rc, 'tfr_list_order_interleaved_i32')
    return order


def _launch_order(labels, mask):
    return launch_order_interleaved(labels, mask) if _ORDER_INTERLEAVED else list_order(labels, mask)


# The launch order is a function of the LABELS (and the mask) alone and only steers load balance -- the loss kernels write
# every list to its own rows, so their results do not depend on it.  EAGER calls cache it per label tensor (round 5): keyed
# on the tensor object, its storage address and its version counter (bumped by every in-place write), so a batch whose
# labels are passed again unchanged -- an evaluation pass, an epoch over a device-resident dataset -- pays the two ordering
# launches once.  A stale entry could only ever cost balance, never correctness.
# Round 6 (ADVICE r5 high + medium): a stream capture NEVER reads or writes the cache.  The two ordering launches are
# recorded into the graph like rounds 1-4 did, so (a) a replay after `labels.copy_(next batch)` orders the NEW batch (a
# cached order baked into the graph was the warm-up batch's: a silently random order for every later batch), and (b) no
# graph ever holds the address of a cache entry -- round 5 "pinned" such entries and a later eager miss on the same key
# dropped the only reference (use-after-free under replay).  A caller who knows the labels of a captured step never
# change passes a ready order: `balance=order`.  TFR_ORDER_CACHE=0 (or order_cache(False)) turns the eager cache off too.
_ORDER_CACHE_ON = os.environ.get('TFR_ORDER_CACHE', '1') != '0'
_order_lru: 'OrderedDict[Tuple, Tuple]' = OrderedDict()
_ORDER_CACHE_CAPACITY = 64


class order_cache(object):
    """Context manager / switch: ``with order_cache(False): ...`` computes the launch order in every call."""

    def __init__(self, enabled: bool):
        self._enabled = bool(enabled)

    def __enter__(self):
        global _ORDER_CACHE_ON
        self._saved, _ORDER_CACHE_ON = _ORDER_CACHE_ON, self._enabled
        return self

    def __exit__(self, *exc):
        global _ORDER_CACHE_ON
        _ORDER_CACHE_ON = self._saved


_FIXED_ORDER = None


class launch_order(object):
    """``with launch_order(order): step()``: every automatic ordering decision inside takes this ready int32 [B] order
    (``tfr_list_order_i32``'s output for the labels in use) -- for a caller who captures a step whose labels never change
    (an evaluation set resident on the device) and does not want the two ordering launches inside every replay.  The
    caller keeps `order` alive as long as the graph."""

    def __init__(self, order):
        self._order = order

    def __enter__(self):
        global _FIXED_ORDER
        self._saved, _FIXED_ORDER = _FIXED_ORDER, self._order
        return self

    def __exit__(self, *exc):
        global _FIXED_ORDER
        _FIXED_ORDER = self._saved


def _cached_order(labels, mask):
    import weakref
    if _FIXED_ORDER is not None and _FIXED_ORDER.shape[0] == labels.shape[0]:
        return _FIXED_ORDER
    if not _ORDER_CACHE_ON or torch.cuda.is_current_stream_capturing():
        return _launch_order(labels, mask)      # inside a capture: a node of the graph, memory of the graph's pool
    key = (labels.data_ptr(), tuple(labels.shape), str(labels.device), None if mask is None else mask.data_ptr())
    stamp = (labels._version, None if mask is None else mask._version)
    ent = _order_lru.get(key)
    if ent is not None and ent[0] == stamp and ent[1]() is labels:       # (the mask is a fresh uint8 view per call: address + version)
        _order_lru.move_to_end(key)
        return ent[2]
    order = _launch_order(labels, mask)
    _order_lru[key] = (stamp, weakref.ref(labels), order)
    while len(_order_lru) > _ORDER_CACHE_CAPACITY:
        _order_lru.popitem(last=False)
    return order


def _auto_order(labels, mask, balance, min_list_size):
    """balance: None = automatic, False = index order, True = compute the order, or a ready int32 [B] order.
    Automatic: the two ordering launches cost ~20 us; they pay for themselves when the tail they remove
    (about one long list's wave time, ~ list_size^2) is longer -- measured break-even (B = 16384, valid length
    U{L/2..L}): ApproxNDCG between list_size 100 (0.063 -> 0.074 ms, a loss) and 200 (0.166 -> 0.149 ms with the
    ordering launches included, 0.137 with a ready order); the (4x heavier per pair) pairwise losses around 128."""
    if torch.is_tensor(balance):
        return balance
    B, L = labels.shape
    if balance is None:
        balance = B >= _BALANCE_MIN_LISTS and L >= min_list_size
    return _cached_order(labels, mask) if balance else None


def approx_ndcg(logits, labels, mask=None, list_scale=None, temperature=0.1, lanes_per_row=0,
                want_grad=True, balance=None, want_sum=False):
    """want_sum=True: a fourth result, the 0-d sum_b loss_b * list_scale_b, added up in a fixed order inside the same
    launch (tfr_approx_ndcg_sum_f32) -- the scalar a reduced loss returns."""
    logits = _f32(logits, 'logits'); labels = _f32(labels, 'labels')
    _check2d(logits, 'logits'); _same_shape(labels, logits, 'labels', 'logits')
    mask = _u8(mask, 'mask'); list_scale = _f32(list_scale, 'list_scale')
    B, L = logits.shape
    tab = rank_table(_inv_log1p, L, logits.device)
    loss = torch.empty((B,), dtype=torch.float32, device=logits.device)
    weight = torch.empty((B,), dtype=torch.float32, device=logits.device)
    dlogits = torch.empty((B, L), dtype=torch.float32, device=logits.device) if want_grad else None
    if want_sum:
        total, ticket = _sum_outputs(logits.device)
        rc = _lib.load().tfr_approx_ndcg_sum_f32(_ptr(logits), _ptr(labels), _ptr(mask), _ptr(tab),
                                                 _ptr(list_scale), B, L, float(temperature), int(lanes_per_row),
                                                 _ptr(loss), _ptr(weight), _ptr(dlogits),
                                                 _ptr(_auto_order(labels, mask, balance, 192)), _ptr(total),
                                                 _ptr(ticket), _stream())
        _lib.check(rc, 'tfr_approx_ndcg_sum_f32')
        return loss, weight, dlogits, total
    rc = _lib.load().tfr_approx_ndcg_f32(_ptr(logits), _ptr(labels), _ptr(mask), _ptr(tab),
                                         _ptr(list_scale), B, L, float(temperature), int(lanes_per_row),
                                         _ptr(loss), _ptr(weight), _ptr(dlogits),
                                         _ptr(_auto_order(labels, mask, balance, 192)), _stream())
    _lib.check(rc, 'tfr_approx_ndcg_f32')
    return loss, weight, dlogits


def approx_mrr(logits, labels, mask=None, list_scale=None, temperature=0.1, want_grad=True, balance=None):
    logits = _f32(logits, 'logits'); labels = _f32(labels, 'labels')
    _check2d(logits, 'logits'); _same_shape(labels, logits, 'labels', 'logits')
    mask = _u8(mask, 'mask'); list_scale = _f32(list_scale, 'list_scale')
    B, L = logits.shape
    loss = torch.empty((B,), dtype=torch.float32, device=logits.device)
    weight = torch.empty((B,), dtype=torch.float32, device=logits.device)
    dlogits = torch.empty((B, L), dtype=torch.float32, device=logits.device) if want_grad else None
    rc = _lib.load().tfr_approx_mrr_f32(_ptr(logits), _ptr(labels), _ptr(mask), _ptr(list_scale), B, L,
                                        float(temperature), _ptr(loss), _ptr(weight), _ptr(dlogits),
                                        _ptr(_auto_order(labels, mask, balance, 192)), _stream())
    _lib.check(rc, 'tfr_approx_mrr_f32')
    return loss, weight, dlogits


def tie_keys(tie_seed: int, B: int, L: int, device=None) -> torch.Tensor:
    """The [B, L] 15-bit tie keys the kernels derive from a tie seed (csrc/common.h tie_key15; smaller sorts first
    among equal scores, then the index): torch restatement for tests / debugging."""
    if not tie_seed:
        return torch.zeros((B, L), dtype=torch.int64, device=device)
    m32 = 0xffffffff
    b = torch.arange(B, dtype=torch.int64, device=device).unsqueeze(1)
    i = torch.arange(L, dtype=torch.int64, device=device).unsqueeze(0)
    h = (b * 0x9E3779B1 + i * 0x85EBCA77 + (int(tie_seed) & m32)) & m32
    h = h ^ (h >> 16); h = (h * 0x7feb352d) & m32
    h = h ^ (h >> 15); h = (h * 0x846ca68b) & m32
    h = h ^ (h >> 16)
    return h >> 17


def list_mle(logits, labels, mask=None, pos_weight=None, list_scale=None, temperature=1.0, want_grad=True,
             want_sum=False, tie_seed=0):
    """want_sum=True (here and in the other losses below): one more result, the 0-d reduced scalar of the launch
    (sum_b loss_b * list_scale_b) from the `*_sum_f32` entry point -- no reduction launch.  tie_seed != 0: equal labels
    in the hashed order of `tie_keys` (the reference's shuffle_ties), 0: index order."""
    logits = _f32(logits, 'logits'); labels = _f32(labels, 'labels')
    _check2d(logits, 'logits'); _same_shape(labels, logits, 'labels', 'logits')
    mask = _u8(mask, 'mask'); list_scale = _f32(list_scale, 'list_scale'); pos_weight = _f32(pos_weight, 'pos_weight')
    B, L = logits.shape
    loss = torch.empty((B,), dtype=torch.float32, device=logits.device)
    dlogits = torch.empty((B, L), dtype=torch.float32, device=logits.device) if want_grad else None
    ws, ws_bytes = _workspace(WS_LIST_MLE, B, L, logits.device)
    if want_sum:
        total, ticket = _sum_outputs(logits.device)
        rc = _lib.load().tfr_list_mle_sum_f32(_ptr(logits), _ptr(labels), _ptr(mask), _ptr(pos_weight), _ptr(list_scale),
                                              B, L, float(temperature), _ptr(loss), _ptr(dlogits), _ptr(total),
                                              _ptr(ticket), int(tie_seed) & 0xffffffff, _ptr(ws), ws_bytes, _stream())
        _lib.check(rc, 'tfr_list_mle_sum_f32')
        return loss, dlogits, total
    rc = _lib.load().tfr_list_mle_f32(_ptr(logits), _ptr(labels), _ptr(mask), _ptr(pos_weight), _ptr(list_scale),
                                      B, L, float(temperature), _ptr(loss), _ptr(dlogits), int(tie_seed) & 0xffffffff,
                                      _ptr(ws), ws_bytes, _stream())
    _lib.check(rc, 'tfr_list_mle_f32')
    return loss, dlogits


def unique_softmax(logits, labels, mask=None, list_scale=None, temperature=1.0, want_grad=True, want_sum=False):
    logits = _f32(logits, 'logits'); labels = _f32(labels, 'labels')
    _check2d(logits, 'logits'); _same_shape(labels, logits, 'labels', 'logits')
    mask = _u8(mask, 'mask'); list_scale = _f32(list_scale, 'list_scale')
    B, L = logits.shape
    loss = torch.empty((B,), dtype=torch.float32, device=logits.device)
    dlogits = torch.empty((B, L), dtype=torch.float32, device=logits.device) if want_grad else None
    ws, ws_bytes = _workspace(WS_UNIQUE_SOFTMAX, B, L, logits.device)
    if want_sum:
        total, ticket = _sum_outputs(logits.device)
        rc = _lib.load().tfr_unique_softmax_sum_f32(_ptr(logits), _ptr(labels), _ptr(mask), _ptr(list_scale), B, L,
                                                    float(temperature), _ptr(loss), _ptr(dlogits), _ptr(total),
                                                    _ptr(ticket), _ptr(ws), ws_bytes, _stream())
        _lib.check(rc, 'tfr_unique_softmax_sum_f32')
        return loss, dlogits, total
    rc = _lib.load().tfr_unique_softmax_f32(_ptr(logits), _ptr(labels), _ptr(mask), _ptr(list_scale), B, L,
                                            float(temperature), _ptr(loss), _ptr(dlogits), _ptr(ws), ws_bytes, _stream())
    _lib.check(rc, 'tfr_unique_softmax_f32')
    return loss, dlogits


POINT_SIGMOID_CE, POINT_MSE = 0, 1


def pointwise_loss(kind, logits, labels, mask=None, item_weights=None, list_weights=None, temperature=1.0,
                   want_grad=True, want_sum=False):
    """tfr_pointwise_loss_f32 -> (list_loss [B], list_weight [B], list_nnz [B], dlogits [B, L]) (+ 0-d sum of list_loss)."""
    logits = _f32(logits, 'logits'); labels = _f32(labels, 'labels')
    _check2d(logits, 'logits'); _same_shape(labels, logits, 'labels', 'logits')
    mask = _u8(mask, 'mask'); item_weights = _f32(item_weights, 'item_weights')
    list_weights = _f32(list_weights, 'list_weights')
    B, L = logits.shape
    dev = logits.device
    loss = torch.empty((B,), dtype=torch.float32, device=dev)
    weight = torch.empty((B,), dtype=torch.float32, device=dev)
    nnz = torch.empty((B,), dtype=torch.float32, device=dev)
    dlogits = torch.empty((B, L), dtype=torch.float32, device=dev) if want_grad else None
    if want_sum:
        total, ticket = _sum_outputs(dev)
        rc = _lib.load().tfr_pointwise_loss_sum_f32(int(kind), _ptr(logits), _ptr(labels), _ptr(mask), _ptr(item_weights),
                                                    _ptr(list_weights), B, L, float(temperature), _ptr(loss), _ptr(weight),
                                                    _ptr(nnz), _ptr(dlogits), _ptr(total), _ptr(ticket), _stream())
        _lib.check(rc, 'tfr_pointwise_loss_sum_f32')
        return loss, weight, nnz, dlogits, total
    rc = _lib.load().tfr_pointwise_loss_f32(int(kind), _ptr(logits), _ptr(labels), _ptr(mask), _ptr(item_weights),
                                            _ptr(list_weights), B, L, float(temperature), _ptr(loss), _ptr(weight),
                                            _ptr(nnz), _ptr(dlogits), _stream())
    _lib.check(rc, 'tfr_pointwise_loss_f32')
    return loss, weight, nnz, dlogits


def circle_loss(logits, labels, mask=None, list_scale=None, gamma=64.0, margin=0.25, clip=True, want_grad=True):
    """tfr_circle_loss_f32 -> (loss [B], weight [B] (NaN where a list has no pair), dlogits [B, L])."""
    logits = _f32(logits, 'logits'); labels = _f32(labels, 'labels')
    _check2d(logits, 'logits'); _same_shape(labels, logits, 'labels', 'logits')
    mask = _u8(mask, 'mask'); list_scale = _f32(list_scale, 'list_scale')
    B, L = logits.shape
    loss = torch.empty((B,), dtype=torch.float32, device=logits.device)
    weight = torch.empty((B,), dtype=torch.float32, device=logits.device)
    dlogits = torch.empty((B, L), dtype=torch.float32, device=logits.device) if want_grad else None
    ws, ws_bytes = _workspace(WS_CIRCLE, B, L, logits.device)
    rc = _lib.load().tfr_circle_loss_f32(_ptr(logits), _ptr(labels), _ptr(mask), _ptr(list_scale), B, L,
                                         float(gamma), float(margin), int(bool(clip)), _ptr(loss), _ptr(weight),
                                         _ptr(dlogits), _ptr(ws), ws_bytes, _stream())
    _lib.check(rc, 'tfr_circle_loss_f32')
    return loss, weight, dlogits


NEURAL_SORT_NDCG, NEURAL_SORT_CE = 0, 1


def neural_sort_loss(kind, logits, labels, mask=None, list_scale=None, temperature=1.0, want_grad=True):
    """tfr_neural_sort_loss_f32: per-list NeuralSortNDCG / NeuralSortCrossEntropy loss and its gradient."""
    logits = _f32(logits, 'logits'); labels = _f32(labels, 'labels')
    _check2d(logits, 'logits'); _same_shape(labels, logits, 'labels', 'logits')
    mask = _u8(mask, 'mask'); list_scale = _f32(list_scale, 'list_scale')
    B, L = logits.shape
    tab = rank_table(_inv_log1p, L, logits.device) if kind == NEURAL_SORT_NDCG else None
    loss = torch.empty((B,), dtype=torch.float32, device=logits.device)
    dlogits = torch.empty((B, L), dtype=torch.float32, device=logits.device) if want_grad else None
    ws, ws_bytes = _workspace(WS_NEURAL_SORT_NDCG if kind == NEURAL_SORT_NDCG else WS_NEURAL_SORT_CE, B, L, logits.device)
    rc = _lib.load().tfr_neural_sort_loss_f32(int(kind), _ptr(logits), _ptr(labels), _ptr(mask), _ptr(tab),
                                              _ptr(list_scale), B, L, float(temperature), _ptr(loss),
                                              _ptr(dlogits), _ptr(ws), ws_bytes, _stream())
    _lib.check(rc, 'tfr_neural_sort_loss_f32')
    return loss, dlogits


def pairwise_logistic(logits, labels, mask=None, item_weights=None, list_weights=None,
                      lambda_kind=LAMBDA_NONE, topn=0, smooth_fraction=0.0, normalized=False,
                      gain_kind=GAIN_IDENTITY, gains=None, discount=None, temperature=1.0,
                      want_grad=True, want_rows=True, want_aux=True, loss_kind=0, balance=None, want_list=False,
                      want_sum=False, tie_seed=0):
    """loss_kind: PAIR_LOGISTIC / PAIR_HINGE / PAIR_SOFT_ZERO_ONE.  want_aux=False skips the per-row weight sums and the non-zero pair counts (only the MEAN /
    SUM_BY_NONZERO_WEIGHTS reductions and compute_per_list need them): a leaner kernel variant.  want_list=True
    returns the per-list loss sums [B] in place of the [B, L] row losses (5-tuple: rows, weights, nnz, dlogits, list).
    tie_seed != 0: equal scores are ranked in the hashed order of `tie_keys` (the reference's _compute_ranks with
    shuffle_ties) -- on the workgroup kernel, not the LambdaRank fast paths."""
    logits = _f32(logits, 'logits'); labels = _f32(labels, 'labels')
    _check2d(logits, 'logits'); _same_shape(labels, logits, 'labels', 'logits')
    mask = _u8(mask, 'mask'); item_weights = _f32(item_weights, 'item_weights')
    list_weights = _f32(list_weights, 'list_weights'); gains = _f32(gains, 'gains')
    discount = _f32(discount, 'discount')
    B, L = logits.shape
    dev = logits.device
    row_loss = torch.empty((B, L), dtype=torch.float32, device=dev) if want_rows else None
    row_weight = torch.empty((B, L), dtype=torch.float32, device=dev) if (want_rows and want_aux) else None
    nnz = torch.empty((B,), dtype=torch.float32, device=dev) if want_aux else None
    dlogits = torch.empty((B, L), dtype=torch.float32, device=dev) if want_grad else None
    list_loss = torch.empty((B,), dtype=torch.float32, device=dev) if (want_list or want_sum) else None
    if want_sum:                                   # (6-tuple: ..., list_loss, 0-d sum of list_loss)
        total, ticket = _sum_outputs(dev)
        rc = _lib.load().tfr_pairwise_loss_sum_f32(
            int(loss_kind), _ptr(logits), _ptr(labels), _ptr(mask), _ptr(item_weights), _ptr(list_weights),
            int(lambda_kind), int(topn or 0), float(smooth_fraction), int(bool(normalized)), int(gain_kind),
            _ptr(gains), _ptr(discount), B, L, float(temperature), _ptr(row_loss), _ptr(row_weight),
            _ptr(nnz), _ptr(dlogits), _ptr(_auto_order(labels, mask, balance, 128)), _ptr(list_loss), _ptr(total),
            _ptr(ticket), int(tie_seed) & 0xffffffff, _stream())
        _lib.check(rc, 'tfr_pairwise_loss_sum_f32')
        return row_loss, row_weight, nnz, dlogits, list_loss, total
    rc = _lib.load().tfr_pairwise_loss_f32(
        int(loss_kind), _ptr(logits), _ptr(labels), _ptr(mask), _ptr(item_weights), _ptr(list_weights),
        int(lambda_kind), int(topn or 0), float(smooth_fraction), int(bool(normalized)), int(gain_kind),
        _ptr(gains), _ptr(discount), B, L, float(temperature), _ptr(row_loss), _ptr(row_weight),
        _ptr(nnz), _ptr(dlogits), _ptr(_auto_order(labels, mask, balance, 128)), _ptr(list_loss),
        int(tie_seed) & 0xffffffff, _stream())
    _lib.check(rc, 'tfr_pairwise_loss_f32')
    if want_list:
        return row_loss, row_weight, nnz, dlogits, list_loss
    return row_loss, row_weight, nnz, dlogits


def softmax_loss(logits, labels, mask=None, weights=None, lambda_kind=LAMBDA_NONE, topn=0,
                 normalized=False, gain_kind=GAIN_IDENTITY, gains=None, discount=None,
                 temperature=1.0, want_grad=True, poly_epsilon=0.0, want_sum=False):
    """poly_epsilon != 0: PolyOneSoftmaxLoss (loss += epsilon * (1 - sum p softmax)).  want_sum=True: a fourth result,
    the 0-d sum_b loss_b * weight_b from the same launch (tfr_softmax_loss_sum_f32); want_sum='partials': the launch
    leaves per-contributor partial sums and ONE short tfr_list_dot_f32 adds them (the default of SoftmaxLoss.loss_and_grad)."""
    logits = _f32(logits, 'logits'); labels = _f32(labels, 'labels')
    _check2d(logits, 'logits'); _same_shape(labels, logits, 'labels', 'logits')
    mask = _u8(mask, 'mask'); gains = _f32(gains, 'gains'); discount = _f32(discount, 'discount')
    w, per_list = _weights_arg(weights, labels)
    B, L = logits.shape
    dev = logits.device
    loss = torch.empty((B,), dtype=torch.float32, device=dev)
    weight = torch.empty((B,), dtype=torch.float32, device=dev)
    dlogits = torch.empty((B, L), dtype=torch.float32, device=dev) if want_grad else None
    if want_sum:
        scratch = torch.empty((max(B, 1),), dtype=torch.float32, device=dev)
        if want_sum == 'partials':
            # the launch leaves one value per contributor (a list, or a wavefront of the streaming form: 8 192 of them for
            # any batch); one short tfr_list_dot_f32 adds them -- no ticket chain behind a 5-25 us kernel
            total, ticket = None, None
        else:
            total, ticket = _sum_outputs(dev)
        rc = _lib.load().tfr_softmax_loss_sum_f32(
            _ptr(logits), _ptr(labels), _ptr(mask), _ptr(w), per_list, int(lambda_kind), int(topn or 0),
            int(bool(normalized)), int(gain_kind), _ptr(gains), _ptr(discount), B, L, float(temperature),
            float(poly_epsilon), _ptr(loss), _ptr(weight), _ptr(dlogits), _ptr(total), _ptr(scratch), _ptr(ticket),
            _stream())
        _lib.check(rc, 'tfr_softmax_loss_sum_f32')
        if total is None:
            n = _lib.load().tfr_softmax_sum_contributors(B, L, int(mask is not None),
                                                         0 if w is None else (1 if per_list else 2),
                                                         int(lambda_kind), int(want_grad))
            if n < 0:
                _lib.check(n, 'tfr_softmax_sum_contributors')
            total = list_dot(scratch[:n]) if n > 0 else torch.zeros((), dtype=torch.float32, device=dev)
        return loss, weight, dlogits, total
    rc = _lib.load().tfr_poly1_softmax_loss_f32(
        _ptr(logits), _ptr(labels), _ptr(mask), _ptr(w), per_list, int(lambda_kind), int(topn or 0),
        int(bool(normalized)), int(gain_kind), _ptr(gains), _ptr(discount), B, L, float(temperature),
        float(poly_epsilon), _ptr(loss), _ptr(weight), _ptr(dlogits), _stream())
    _lib.check(rc, 'tfr_poly1_softmax_loss_f32')
    return loss, weight, dlogits


def gumbel_sample(logits, labels, mask=None, uniform=None, seed=0, offset=0, sample_size=8,
                  gumbel_temperature=1.0, step=None, want_labels=False):
    """``step`` (one-element int64 device tensor): the Philox offset of the draw is ``offset + step[0]`` -- a step replayed
    from a hipGraph draws new noise when ``gumbel_sample_bwd(..., step_inc=step)`` advances it (list_size <= 1024).
    ``want_labels``: also returns the labels of the S copies of every list, [B * S, L], written by the same launch."""
    logits = _f32(logits, 'logits'); labels = _f32(labels, 'labels')
    _check2d(logits, 'logits'); _same_shape(labels, logits, 'labels', 'logits')
    mask = _u8(mask, 'mask'); uniform = _f32(uniform, 'uniform')
    B, L = logits.shape
    S = int(sample_size)
    if uniform is not None and tuple(uniform.shape) != (B, S, L):
        raise ValueError('uniform noise must have shape [B, S, L]')
    out = torch.empty((B * S, L), dtype=torch.float32, device=logits.device)
    if (step is not None or want_labels) and L <= 1024:
        if step is not None and (step.numel() != 1 or step.dtype != torch.int64 or not step.is_cuda):
            raise ValueError('step must be a one-element int64 device tensor')
        lab_out = torch.empty((B * S, L), dtype=torch.float32, device=logits.device) if want_labels else None
        rc = _lib.load().tfr_gumbel_sample_step_f32(_ptr(logits), _ptr(labels), _ptr(mask), _ptr(uniform),
                                                    int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1), _ptr(step),
                                                    B, S, L, float(gumbel_temperature), _ptr(out), _ptr(lab_out), _stream())
        _lib.check(rc, 'tfr_gumbel_sample_step_f32')
        return (out, lab_out) if want_labels else out
    if step is not None:
        raise ValueError('a device step counter needs list_size <= 1024')
    rc = _lib.load().tfr_gumbel_sample_f32(_ptr(logits), _ptr(labels), _ptr(mask), _ptr(uniform),
                                           int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1), B, S, L,
                                           float(gumbel_temperature), _ptr(out), _stream())
    _lib.check(rc, 'tfr_gumbel_sample_f32')
    if want_labels:
        return out, labels.unsqueeze(1).expand(B, S, L).reshape(B * S, L).contiguous()
    return out


def gumbel_sample_bwd(sampled, labels, mask, upstream, sample_size, gumbel_temperature, step_inc=None):
    sampled = _f32(sampled, 'sampled'); labels = _f32(labels, 'labels')
    upstream = _f32(upstream, 'upstream'); mask = _u8(mask, 'mask')
    B, L = labels.shape
    S = int(sample_size)
    out = torch.empty((B, L), dtype=torch.float32, device=labels.device)
    if step_inc is not None and L <= 1024 and B > 0:
        rc = _lib.load().tfr_gumbel_sample_bwd_step_f32(_ptr(sampled), _ptr(labels), _ptr(mask), _ptr(upstream),
                                                        B, S, L, float(gumbel_temperature), _ptr(out), _ptr(step_inc), _stream())
        _lib.check(rc, 'tfr_gumbel_sample_bwd_step_f32')
        return out
    rc = _lib.load().tfr_gumbel_sample_bwd_f32(_ptr(sampled), _ptr(labels), _ptr(mask), _ptr(upstream),
                                               B, S, L, float(gumbel_temperature), _ptr(out), _stream())
    _lib.check(rc, 'tfr_gumbel_sample_bwd_f32')
    return out


def list_dot(x, w=None):
    """sum(x * w) (or sum(x)) of a per-list vector as a 0-d tensor: one launch, fixed summation order
    (tfr_list_dot_f32); vectors beyond 65536 entries use the library reduction."""
    x = _f32(x, 'x').reshape(-1)
    w = None if w is None else _f32(w, 'w').reshape(-1)
    n = x.numel()
    if n > 65536 or (x.data_ptr() & 15) or (w is not None and (w.data_ptr() & 15)):
        return torch.dot(x, w) if w is not None else x.sum()
    out = torch.empty((), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().tfr_list_dot_f32(_ptr(x), _ptr(w), n, _ptr(out), _stream()), 'tfr_list_dot_f32')
    return out


def device_guarded(fn):
    """Runs ``fn`` with the HIP device of its first device-tensor argument current, so that ``_stream()`` and every
    allocation inside land on the device the tensors live on (a caller holding cuda:1 tensors while cuda:0 is current
    would otherwise launch on the wrong device's stream)."""
    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        dev = None
        for a in args:
            if isinstance(a, (list, tuple)) and a:
                a = a[0]
            if torch.is_tensor(a) and a.is_cuda:
                dev = a.device
                break
        if dev is None or dev.index == torch.cuda.current_device():
            return fn(*args, **kwargs)
        with torch.cuda.device(dev):
            return fn(*args, **kwargs)
    return wrapper


def _guard_module(namespace, module_name, skip=()):
    for name, obj in list(namespace.items()):
        if (callable(obj) and not name.startswith('_') and getattr(obj, '__module__', None) == module_name
                and isinstance(obj, type(_guard_module)) and name not in skip):
            namespace[name] = device_guarded(obj)


_guard_module(globals(), __name__, skip=('require_device', 'rank_table', 'device_guarded', 'order_cache', 'launch_order', 'tie_keys'))
