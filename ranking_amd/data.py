"""Input side of the ranking path: mirror of the numeric-feature subset of
``tensorflow_ranking/python/data.py`` and of the LibSVM loader in
``examples/tf_ranking_libsvm.py:137-195`` on top of the native ``libtfr_io.so``
(include/tfr_io.h): TFRecord framing with CRC-32C, ``ExampleListWithContext`` /
``ExampleInExample`` / ``tf.SequenceExample`` decoding, truncation / padding to ``list_size``,
list sizes and mask.

Same entry-point names and keyword arguments as the reference where they exist
(``parse_from_example_list``, ``parse_from_example_in_example``, ``parse_from_sequence_example``,
``make_parsing_fn``, ``build_ranking_dataset``,
``build_ranking_dataset_with_parsing_fn``); feature specs are ``FixedLenFeature`` objects
(numeric, with a default value).  Tensors are returned on the host (pin + copy to the GPU is
the caller's choice); ``shuffle_examples`` permutes the valid examples of each list like
``utils.shuffle_valid_indices`` (the TF random stream itself is not reproducible: SURVEY 8c).
"""
from __future__ import annotations

import collections
import ctypes
import glob as _glob
import os
import queue
import threading
from typing import Dict, Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _io_lib

EIE = 'example_in_example'
ELWC = 'example_list_with_context'
SEQ = 'sequence_example'
_PADDING_LABEL = -1.

FixedLenFeature = collections.namedtuple('FixedLenFeature', ['shape', 'dtype', 'default_value'])
FixedLenFeature.__new__.__defaults__ = (None,)


def _spec_width(spec) -> int:
    shape = tuple(spec.shape) if not isinstance(spec.shape, int) else (spec.shape,)
    w = 1
    for s in shape:
        w *= int(s)
    return max(w, 1)


def _spec_array(feature_spec: Dict[str, FixedLenFeature]):
    names = list(feature_spec)
    arr = (_io_lib.FeatureSpec * max(len(names), 1))()
    keep = []
    for i, name in enumerate(names):
        spec = feature_spec[name]
        if spec.default_value is None:
            raise ValueError('feature %r needs a default_value (padded examples take it)' % name)
        dv = spec.default_value
        if isinstance(dv, (list, tuple, np.ndarray)):
            dv = np.asarray(dv).reshape(-1)[0]
        b = name.encode('utf-8')
        keep.append(b)
        arr[i] = _io_lib.FeatureSpec(b, _spec_width(spec), float(dv))
    return names, arr, keep


def crc32c(data: bytes) -> int:
    return int(_io_lib.load().tfr_io_crc32c(ctypes.c_char_p(data), len(data)))


def masked_crc32c(data: bytes) -> int:
    return int(_io_lib.load().tfr_io_masked_crc32c(ctypes.c_char_p(data), len(data)))


def write_tfrecord(path: str, records: Iterable[bytes]) -> None:
    """tf.io.TFRecordWriter framing (used for synthetic ELWC inputs and tests)."""
    with open(path, 'wb') as f:
        for r in records:
            hdr = len(r).to_bytes(8, 'little')
            f.write(hdr); f.write(masked_crc32c(hdr).to_bytes(4, 'little'))
            f.write(r); f.write(masked_crc32c(r).to_bytes(4, 'little'))


def read_tfrecord(path: str, verify_crc: bool = True) -> List[bytes]:
    """All records of one TFRecord file (tf.data.TFRecordDataset semantics)."""
    lib = _io_lib.load()
    size = os.path.getsize(path)
    if size == 0:
        return []
    with open(path, 'rb') as f:
        buf = f.read()
    cbuf = ctypes.c_char_p(buf)
    n = _io_lib.check(lib.tfr_io_tfrecord_index(cbuf, len(buf), int(verify_crc), None, None, 0),
                      'tfr_io_tfrecord_index(%s)' % path)
    off = np.zeros(n, dtype=np.uint64); ln = np.zeros(n, dtype=np.uint64)
    _io_lib.check(lib.tfr_io_tfrecord_index(cbuf, len(buf), 0, off.ctypes.data, ln.ctypes.data, n),
                  'tfr_io_tfrecord_index')
    return [buf[int(o):int(o) + int(l)] for o, l in zip(off, ln)]


def _record_arrays(serialized: Sequence[bytes]):
    B = len(serialized)
    ptrs = (ctypes.c_char_p * max(B, 1))(*serialized)
    lens = np.asarray([len(s) for s in serialized], dtype=np.uint64)
    return ptrs, lens


def parse_from_example_list(serialized: Sequence[bytes], list_size: Optional[int] = None,
                            context_feature_spec: Optional[Dict[str, FixedLenFeature]] = None,
                            example_feature_spec: Optional[Dict[str, FixedLenFeature]] = None,
                            size_feature_name: Optional[str] = None, mask_feature_name: Optional[str] = None,
                            shuffle_examples: bool = False, seed: Optional[int] = None,
                            num_threads: int = 0, example_dtype=torch.float32,
                            float32_features: Sequence[str] = ()) -> Dict[str, torch.Tensor]:
    """data.py:391-540: a batch of serialized ELWC protos -> feature map (see ``_parse_batch``)."""
    return _parse_batch(_io_lib.FORMAT_ELWC, serialized, list_size, context_feature_spec, example_feature_spec,
                        size_feature_name, mask_feature_name, shuffle_examples, seed, num_threads, example_dtype,
                        float32_features)


def parse_from_example_in_example(serialized: Sequence[bytes], list_size: Optional[int] = None,
                                  context_feature_spec: Optional[Dict[str, FixedLenFeature]] = None,
                                  example_feature_spec: Optional[Dict[str, FixedLenFeature]] = None,
                                  size_feature_name: Optional[str] = None, mask_feature_name: Optional[str] = None,
                                  shuffle_examples: bool = False, seed: Optional[int] = None,
                                  num_threads: int = 0, example_dtype=torch.float32,
                                  float32_features: Sequence[str] = ()) -> Dict[str, torch.Tensor]:
    """data.py:211-380: a batch of serialized ExampleInExample protos -- a tf.Example whose bytes features
    ``serialized_context`` / ``serialized_examples`` hold the context and the per-item tf.Examples -- -> feature map,
    with the truncation / padding / sizes / mask / shuffle semantics of the ELWC parser (same parser class in the
    reference, data.py:133-208, 383-388)."""
    return _parse_batch(_io_lib.FORMAT_EIE, serialized, list_size, context_feature_spec, example_feature_spec,
                        size_feature_name, mask_feature_name, shuffle_examples, seed, num_threads, example_dtype,
                        float32_features)


def parse_from_sequence_example(serialized: Sequence[bytes], list_size: Optional[int] = None,
                                context_feature_spec: Optional[Dict[str, FixedLenFeature]] = None,
                                example_feature_spec: Optional[Dict[str, FixedLenFeature]] = None,
                                size_feature_name: Optional[str] = None, mask_feature_name: Optional[str] = None,
                                shuffle_examples: bool = False, seed: Optional[int] = None,
                                num_threads: int = 0, example_dtype=torch.float32,
                                float32_features: Sequence[str] = ()) -> Dict[str, torch.Tensor]:
    """data.py:713-855 (parser :572-710): a batch of serialized tf.SequenceExample protos -> feature map.  Frame t of
    feature_list k is item t's value of example feature k; a missing feature_list has no frames, positions past a
    feature's own frames take its default value, ``list_size=None`` pads to the longest feature_list of the batch, the
    list size of a record is its longest named feature_list.  ``shuffle_examples`` raises like the reference
    (:577-579)."""
    if shuffle_examples:
        raise ValueError('Shuffling examples is not supported in SequenceExample format.')
    return _parse_batch(_io_lib.FORMAT_SEQ, serialized, list_size, context_feature_spec, example_feature_spec,
                        size_feature_name, mask_feature_name, False, seed, num_threads, example_dtype,
                        float32_features)


def parse_from_tf_example(serialized: Sequence[bytes],
                          context_feature_spec: Optional[Dict[str, FixedLenFeature]] = None,
                          example_feature_spec: Optional[Dict[str, FixedLenFeature]] = None,
                          size_feature_name: Optional[str] = None, mask_feature_name: Optional[str] = None,
                          num_threads: int = 0) -> Dict[str, torch.Tensor]:
    """data.py:1348-1395: a batch of serialized ``tf.train.Example`` protos, each ONE item together with its context
    features (the serving signature of a ranking model) -> context features ``[B, width]``, example features
    ``[B, 1, width]``, sizes ``[B]`` of ones (float32 like ``tf.ones``), mask ``[B, 1]`` of True."""
    features = _parse_batch(_io_lib.FORMAT_EXAMPLE, serialized, 1, context_feature_spec, example_feature_spec,
                            size_feature_name, mask_feature_name, False, None, num_threads, torch.float32, ())
    if size_feature_name:
        features[size_feature_name] = features[size_feature_name].to(torch.float32)
    return features


def _parse_batch(fmt: int, serialized: Sequence[bytes], list_size: Optional[int],
                 context_feature_spec: Optional[Dict[str, FixedLenFeature]],
                 example_feature_spec: Optional[Dict[str, FixedLenFeature]],
                 size_feature_name: Optional[str], mask_feature_name: Optional[str],
                 shuffle_examples: bool, seed: Optional[int], num_threads: int, example_dtype,
                 float32_features: Sequence[str]) -> Dict[str, torch.Tensor]:
    """One batch through ``libtfr_io.so``'s ``tfr_io_parse_batch``.  Example features are
    ``[B, list_size, width]`` (fp32; int64 features are converted), context features ``[B, width]``.

    ``example_dtype=torch.bfloat16`` (not in the reference): the example features leave the parser rounded to bfloat16
    (``tfr_io_parse_elwc_batch_bf16``: round to nearest even, the rounding of the scorer's own input cast) -- half the
    bytes to pin and to send over the host link, and ``FusedTower`` gathers them as they are
    (``tfr_tower_cast_gather_bf16_bf16``).  Every example feature must then be a float feature; the ones named in
    ``float32_features`` (labels, real-valued targets: anything that needs more than bfloat16's 8 significant bits) come
    back unrounded as float32 from the same pass."""
    if not example_feature_spec:
        raise ValueError('example_feature_spec {} must not be empty.'.format(example_feature_spec))
    if example_dtype not in (torch.float32, torch.bfloat16):
        raise ValueError('example_dtype must be torch.float32 or torch.bfloat16, got %r' % (example_dtype,))
    as_bf16 = example_dtype == torch.bfloat16
    float32_features = tuple(float32_features)
    for name in float32_features:
        if name not in example_feature_spec:
            raise ValueError('float32_features names %r, which is not an example feature' % name)
    if as_bf16:
        for name, spec in example_feature_spec.items():
            if name in float32_features:
                continue
            if spec.dtype not in (None, torch.float32, torch.bfloat16):
                raise ValueError('example_dtype=bfloat16: feature %r is declared %r; only float features can be '
                                 'shipped as bfloat16' % (name, spec.dtype))
    lib = _io_lib.load()
    serialized = list(serialized)
    B = len(serialized)
    ptrs, lens = _record_arrays(serialized)
    out_list_size = list_size
    ex_names, ex_arr, _k1 = _spec_array(example_feature_spec)
    cx_names, cx_arr, _k2 = _spec_array(context_feature_spec or {})
    if list_size is None or list_size <= 0 or shuffle_examples:
        cur = max(1, int(_io_lib.check(lib.tfr_io_max_list_size(fmt, ptrs, lens.ctypes.data, B, ex_arr, len(ex_names)),
                                       'tfr_io_max_list_size')))
        if list_size is None or list_size <= 0:
            list_size = out_list_size = cur
        else:
            # data.py:164-182: the shuffle runs over ALL examples of the batch's longest list and the truncation to
            # list_size comes after it (a truncated list is a random sample, not the first list_size examples)
            list_size = max(cur, list_size)
    ex_w = [_spec_width(example_feature_spec[n]) for n in ex_names]
    cx_w = [_spec_width(context_feature_spec[n]) for n in cx_names]
    if as_bf16:
        ex_t = torch.empty((B, list_size, sum(ex_w)), dtype=torch.bfloat16)
        ex_ptr = ex_t.data_ptr()
    else:
        ex_out = np.empty((B, list_size, sum(ex_w)), dtype=np.float32)
        ex_ptr = ex_out.ctypes.data
    cx_out = np.empty((B, max(sum(cx_w), 1)), dtype=np.float32)
    sizes = np.zeros(B, dtype=np.int32)
    mask = np.zeros((B, list_size), dtype=np.uint8)
    if num_threads <= 0:
        num_threads = min(16, os.cpu_count() or 1) if B >= 64 else 1
    if as_bf16:
        offs = dict(zip(ex_names, np.cumsum([0] + ex_w[:-1]).tolist()))
        side_cols = [offs[n] + j for n in float32_features for j in range(_spec_width(example_feature_spec[n]))]
        cols = np.asarray(side_cols, dtype=np.int32)
        side = np.empty((B, list_size, max(len(side_cols), 1)), dtype=np.float32)
        _io_lib.check(lib.tfr_io_parse_batch(
            fmt, ptrs, lens.ctypes.data, B, list_size, ex_arr, len(ex_names), cx_arr if cx_names else None,
            len(cx_names), None, ex_ptr, cx_out.ctypes.data if cx_names else None, sizes.ctypes.data, mask.ctypes.data,
            num_threads, cols.ctypes.data if side_cols else None, len(side_cols),
            side.ctypes.data if side_cols else None), 'tfr_io_parse_batch')
        side_t = torch.from_numpy(side)
    else:
        _io_lib.check(lib.tfr_io_parse_batch(
            fmt, ptrs, lens.ctypes.data, B, list_size, ex_arr, len(ex_names), cx_arr if cx_names else None,
            len(cx_names), ex_ptr, None, cx_out.ctypes.data if cx_names else None, sizes.ctypes.data, mask.ctypes.data,
            num_threads, None, 0, None), 'tfr_io_parse_batch')
        ex_t = torch.from_numpy(ex_out)
    if shuffle_examples:
        from . import utils
        is_valid = torch.from_numpy(mask.astype(bool))
        idx = utils.shuffle_valid_indices(is_valid, seed=seed)[:, :out_list_size]
        ex_t = torch.gather(ex_t, 1, idx.unsqueeze(-1).expand(-1, -1, ex_t.shape[2]))
        if as_bf16:
            side_t = torch.gather(side_t, 1, idx.unsqueeze(-1).expand(-1, -1, side_t.shape[2]))
        mask = mask[:, :out_list_size]                     # = sequence_mask(sizes, list_size), data.py:206
    features: Dict[str, torch.Tensor] = {}
    off = soff = 0
    for name, w in zip(ex_names, ex_w):
        spec = example_feature_spec[name]
        if as_bf16 and name in float32_features:
            t = side_t[:, :, soff:soff + w]
            soff += w
            features[name] = t.to(spec.dtype) if spec.dtype not in (None, torch.float32) else t
        elif as_bf16:
            features[name] = ex_t[:, :, off:off + w]
        else:
            t = ex_t[:, :, off:off + w]
            features[name] = t.to(spec.dtype) if spec.dtype not in (None, torch.float32) else t
        off += w
    off = 0
    cx_t = torch.from_numpy(cx_out)
    for name, w in zip(cx_names, cx_w):
        spec = context_feature_spec[name]
        t = cx_t[:, off:off + w]
        features[name] = t.to(spec.dtype) if spec.dtype not in (None, torch.float32) else t
        off += w
    if size_feature_name:
        features[size_feature_name] = torch.from_numpy(sizes)
    if mask_feature_name:
        features[mask_feature_name] = torch.from_numpy(mask.astype(bool))
    return features


def make_parsing_fn(data_format, list_size=None, context_feature_spec=None, example_feature_spec=None,
                    size_feature_name=None, mask_feature_name=None, shuffle_examples=False, seed=None,
                    example_dtype=torch.float32, float32_features=()):
    """data.py:857-911: ``example_list_with_context``, ``example_in_example`` or ``sequence_example``.
    ``example_dtype`` / ``float32_features``: see ``_parse_batch``."""
    parsers = {ELWC: parse_from_example_list, EIE: parse_from_example_in_example, SEQ: parse_from_sequence_example}
    if data_format not in parsers:
        raise ValueError('Format {} is not supported.'.format(data_format))
    parse = parsers[data_format]

    def _fn(serialized):
        return parse(serialized, list_size=list_size, context_feature_spec=context_feature_spec,
                                       example_feature_spec=example_feature_spec,
                                       size_feature_name=size_feature_name, mask_feature_name=mask_feature_name,
                                       shuffle_examples=shuffle_examples, seed=seed, example_dtype=example_dtype,
                                       float32_features=float32_features)
    return _fn


class Prefetcher:
    """``dataset.prefetch(buffer_size)`` (data.py:1015) for this runtime: a background thread runs the wrapped
    iterator ``buffer_size`` batches ahead -- reading and parsing happen inside ``libtfr_io.so``, which runs
    without the GIL -- and, when ``device`` is a HIP device, stages each batch there on a dedicated copy stream
    (pinned host buffer -> ``non_blocking`` copy -> event).  The consumer's stream waits on the event only, so the
    host-to-device copy of batch n+1 overlaps the training step of batch n.  Nested dicts / tuples / lists of
    tensors are handled; anything else passes through."""

    _END = object()

    def __init__(self, iterable: Iterable, buffer_size: int = 2, device=None):
        self._it = iter(iterable)
        self._queue: 'queue.Queue' = queue.Queue(maxsize=max(1, int(buffer_size)))
        self._device = torch.device(device) if device is not None else None
        self._cuda = self._device is not None and self._device.type == 'cuda'
        if self._cuda and self._device.index is None:
            self._device = torch.device('cuda', torch.cuda.current_device())
        self._stream = torch.cuda.Stream(self._device) if self._cuda else None
        self._stop = threading.Event()
        self._done = False
        self._thread = threading.Thread(target=self._run, name='tfr-prefetch', daemon=True)
        self._thread.start()

    def _map(self, obj, fn):
        if torch.is_tensor(obj):
            return fn(obj)
        if isinstance(obj, dict):
            return {k: self._map(v, fn) for k, v in obj.items()}
        if isinstance(obj, tuple):
            return tuple(self._map(v, fn) for v in obj)
        if isinstance(obj, list):
            return [self._map(v, fn) for v in obj]
        return obj

    def _transfer(self, t: torch.Tensor) -> torch.Tensor:
        if self._cuda:
            return (t if t.is_cuda else t.pin_memory()).to(self._device, non_blocking=True)
        return t.to(self._device)

    def _stage(self, batch):
        """Device copies of every tensor of the batch.  Tensors that are views of ONE host storage -- the per-feature
        slices ``parse_from_example_list`` cuts out of its dense ``[B, list_size, F]`` array: 137 of them for the
        reference's one-feature-per-column data -- move with one pinned copy of that storage and are re-cut on the
        device; staged one by one, every slice is a strided gather over the whole array on the host (137 passes over it)
        and a copy of its own."""
        leaves = []
        self._map(batch, lambda t: (leaves.append(t), t)[1])
        groups = {}
        for t in leaves:
            if not t.is_cuda and t.numel() > 0:
                groups.setdefault((t.untyped_storage().data_ptr(), t.dtype), []).append(t)
        moved = {}
        for (_, dtype), views in groups.items():
            if len(views) < 2:
                continue
            base = torch.empty(0, dtype=dtype).set_(views[0].untyped_storage())      # the whole storage, 1-D
            if 2 * sum(v.numel() for v in views) < base.numel():                     # a few small views of a big array
                continue
            on_device = self._transfer(base)
            for v in views:
                moved[id(v)] = on_device.as_strided(v.size(), v.stride(), v.storage_offset())
        return self._map(batch, lambda t: moved[id(t)] if id(t) in moved else self._transfer(t))

    def _put(self, item) -> bool:
        while not self._stop.is_set():
            try:
                self._queue.put(item, timeout=0.1)
                return True
            except queue.Full:
                continue
        return False

    def _run(self):
        try:
            if self._cuda:
                torch.cuda.set_device(self._device)
            for batch in self._it:
                event = None
                if self._cuda:
                    with torch.cuda.stream(self._stream):
                        batch = self._stage(batch)
                        event = torch.cuda.Event()
                        event.record(self._stream)
                elif self._device is not None:
                    batch = self._stage(batch)
                if not self._put((batch, event)):
                    return
            self._put(self._END)
        except BaseException as e:                              # handed to the consumer, raised there
            self._put(e)

    def __iter__(self):
        return self

    def __next__(self):
        if self._done:
            raise StopIteration
        item = self._queue.get()
        if item is self._END:
            self._done = True
            raise StopIteration
        if isinstance(item, BaseException):
            self._done = True
            raise item
        batch, event = item
        if event is not None:
            current = torch.cuda.current_stream(self._device)
            current.wait_event(event)
            self._map(batch, lambda t: (t.record_stream(current), t)[1])     # allocator: in use on `current` now
        return batch

    def close(self):
        """Stops the background thread (needed for endless iterators: ``num_epochs=None``)."""
        self._stop.set()
        self._done = True
        try:
            while True:
                self._queue.get_nowait()
        except queue.Empty:
            pass
        self._thread.join(timeout=5.0)

    def __del__(self):
        self._stop.set()


def build_ranking_dataset_with_parsing_fn(file_pattern, parsing_fn, batch_size, reader=None, reader_args=None,
                                          num_epochs=None, shuffle=True, shuffle_buffer_size=10000,
                                          shuffle_seed=None, prefetch_buffer_size=None, reader_num_threads=None,
                                          sloppy_ordering=False, drop_final_batch=False,
                                          num_parser_threads=None, shard=None) -> Iterator[Dict[str, torch.Tensor]]:
    """data.py:914-1017 as an iterator of parsed batches (file and record shuffling with a ``torch.Generator``).
    ``prefetch_buffer_size`` > 0 reads and parses that many batches ahead on a background thread (``Prefetcher``;
    the reference's AUTOTUNE default is None here = no thread); the interleave knobs are accepted and ignored.
    ``shard=(rank, world)``: data-parallel input -- every rank keeps its ``1 / world`` of the identically shuffled
    record stream (what ``tf.distribute`` does to a dataset under ``AutoShardPolicy.DATA``); ``batch_size`` is then
    the per-rank batch.  All ranks must pass the same ``shuffle_seed``."""
    if shard is not None and not (0 <= int(shard[0]) < int(shard[1])):
        raise ValueError('shard must be (rank, world) with 0 <= rank < world, got %r' % (shard,))
    gen = _ranking_batches(file_pattern, parsing_fn, batch_size, num_epochs, shuffle, shuffle_seed, drop_final_batch,
                           shuffle_buffer_size, shard)
    if prefetch_buffer_size is not None and int(prefetch_buffer_size) > 0:
        return Prefetcher(gen, buffer_size=int(prefetch_buffer_size))
    return gen


def _shuffle_stream(records: Iterable[bytes], buffer_size: int, g: torch.Generator) -> Iterator[bytes]:
    """``dataset.shuffle(buffer_size)``: keep ``buffer_size`` records, emit a uniformly chosen one for every record
    read and refill its slot; drain in random order at the end.  A buffer at least as large as the data gives a
    uniform permutation; memory is the buffer, not the dataset."""
    buf: List[bytes] = []
    for r in records:
        if len(buf) < buffer_size:
            buf.append(r)
            continue
        k = int(torch.randint(len(buf), (1,), generator=g))
        out, buf[k] = buf[k], r
        yield out
    for k in torch.randperm(len(buf), generator=g).tolist():
        yield buf[k]


def _shard_stream(records: Iterable[bytes], rank: int, world: int) -> Iterator[bytes]:
    """tf.data's AutoShardPolicy.DATA: every rank walks the same (identically seeded) stream and keeps record
    ``k * world + rank`` of it.  Only complete groups of ``world`` records are handed out, so all ranks see the same
    number of records per epoch (a collective never waits for a rank that ran out of data)."""
    group: List[bytes] = []
    for r in records:
        group.append(r)
        if len(group) == world:
            yield group[rank]
            group = []


def _ranking_batches(file_pattern, parsing_fn, batch_size, num_epochs, shuffle, shuffle_seed, drop_final_batch,
                     shuffle_buffer_size=10000, shard=None):
    """Streams: one file in memory at a time (files in a fresh random order per epoch when ``shuffle``), records
    through a ``shuffle_buffer_size`` shuffle buffer, ``batch_size`` records per parsed batch (data.py:975-1013)."""
    files = sorted(sum((_glob.glob(p) for p in ([file_pattern] if isinstance(file_pattern, str) else file_pattern)), []))
    if not files:
        raise ValueError('no files match %r' % (file_pattern,))
    g = torch.Generator().manual_seed(0 if shuffle_seed is None else int(shuffle_seed))
    epoch = 0
    while num_epochs is None or epoch < num_epochs:
        order = torch.randperm(len(files), generator=g).tolist() if shuffle else range(len(files))
        stream: Iterable[bytes] = (r for fi in order for r in read_tfrecord(files[fi]))
        if shuffle:
            stream = _shuffle_stream(stream, max(1, int(shuffle_buffer_size or 1)), g)
        if shard is not None and shard[1] > 1:
            stream = _shard_stream(stream, int(shard[0]), int(shard[1]))
        chunk: List[bytes] = []
        for r in stream:
            chunk.append(r)
            if len(chunk) == batch_size:
                yield parsing_fn(chunk)
                chunk = []
        if chunk and not drop_final_batch:
            yield parsing_fn(chunk)
        epoch += 1


def build_ranking_dataset(file_pattern, data_format, batch_size, context_feature_spec, example_feature_spec,
                          list_size=None, size_feature_name=None, mask_feature_name=None, shuffle_examples=False,
                          seed=None, example_dtype=torch.float32, float32_features=(), **kwargs):
    """data.py:1020-1068.  ``example_dtype``: see ``parse_from_example_list``."""
    parsing_fn = make_parsing_fn(data_format, list_size, context_feature_spec, example_feature_spec,
                                 size_feature_name=size_feature_name, mask_feature_name=mask_feature_name,
                                 shuffle_examples=shuffle_examples, seed=seed, example_dtype=example_dtype,
                                 float32_features=float32_features)
    return build_ranking_dataset_with_parsing_fn(file_pattern, parsing_fn, batch_size, **kwargs)


def read_batched_sequence_example_dataset(file_pattern, batch_size, list_size, context_feature_spec, example_feature_spec,
                                          reader=None, reader_args=None, num_epochs=None, shuffle=True,
                                          shuffle_buffer_size=1000, shuffle_seed=None, prefetch_buffer_size=32,
                                          reader_num_threads=10, sloppy_ordering=True, drop_final_batch=False):
    """data.py:1149-1311: ``build_ranking_dataset`` on ``tf.SequenceExample`` records."""
    return build_ranking_dataset(
        file_pattern, SEQ, batch_size, context_feature_spec, example_feature_spec, list_size=list_size, reader=reader,
        reader_args=reader_args, num_epochs=num_epochs, shuffle=shuffle, shuffle_buffer_size=shuffle_buffer_size,
        shuffle_seed=shuffle_seed, prefetch_buffer_size=prefetch_buffer_size, reader_num_threads=reader_num_threads,
        sloppy_ordering=sloppy_ordering, drop_final_batch=drop_final_batch)


def load_libsvm_data(path: str, list_size: int, num_features: int = 136) -> Tuple[torch.Tensor, torch.Tensor]:
    """examples/tf_ranking_libsvm.py:137-195: (features [Q, list_size, num_features] fp32,
    labels [Q, list_size] fp32 with -1 padding); feature k of the reference's per-name map is
    ``features[:, :, k - 1:k]``."""
    lib = _io_lib.load()
    with open(path, 'rb') as f:
        text = f.read()
    ctext = ctypes.c_char_p(text)
    q = _io_lib.check(lib.tfr_io_libsvm_load(ctext, len(text), list_size, num_features, None, None, None),
                      'tfr_io_libsvm_load')
    feats = np.zeros((q, list_size, num_features), dtype=np.float32)
    labels = np.full((q, list_size), _PADDING_LABEL, dtype=np.float32)
    stats = np.zeros(2, dtype=np.int64)
    _io_lib.check(lib.tfr_io_libsvm_load(ctext, len(text), list_size, num_features, feats.ctypes.data,
                                         labels.ctypes.data, stats.ctypes.data), 'tfr_io_libsvm_load')
    return torch.from_numpy(feats), torch.from_numpy(labels)
