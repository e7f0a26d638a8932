"""TEST INFRASTRUCTURE ONLY -- pure-Python restatement of the reference's input path, used to
check ``libtfr_io.so`` (ranking_amd/csrc/tfr_io.cpp).  Never imported by the product path.

Follows (paths under /root/reference/tensorflow_ranking/):
  * python/data.py:59-96   ExampleListWithContext wire format (repeated bytes examples = 1;
                           bytes context = 2), each a serialized tf.Example;
  * python/data.py:133-208 truncate / pad to list_size, sizes, mask; padded examples parse to the
                           spec defaults (an empty serialized Example);
  * examples/tf_ranking_libsvm.py:137-195 load_libsvm_data;
  * the TFRecord container of TensorFlow (length, masked crc32c, data, masked crc32c), a
    third-party dependency (tensorflow < 2.16, tools/pip_package/setup.py:38) whose published
    framing is restated here.
Pinned against the reference's own data files (examples/data/*.tfrecord, *.txt) by
tests/test_data_cpu.py when /root/reference is present, and through tests/golden/elwc_golden.json.
"""
import struct


# ------------------------------------------------------------------ crc32c / TFRecord
def _crc_table():
    tab = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        tab.append(c)
    return tab


_TAB = _crc_table()


def crc32c(data: bytes) -> int:
    c = 0xFFFFFFFF
    for b in data:
        c = _TAB[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc32c(data: bytes) -> int:
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def read_tfrecord(buf: bytes, verify=True):
    out, pos = [], 0
    while pos < len(buf):
        (n,) = struct.unpack_from('<Q', buf, pos)
        (lcrc,) = struct.unpack_from('<I', buf, pos + 8)
        if verify:
            assert masked_crc32c(buf[pos:pos + 8]) == lcrc, 'length crc'
        data = buf[pos + 12:pos + 12 + n]
        (dcrc,) = struct.unpack_from('<I', buf, pos + 12 + n)
        if verify:
            assert masked_crc32c(data) == dcrc, 'data crc'
        out.append(data)
        pos += 12 + n + 4
    return out


def write_tfrecord(records):
    out = bytearray()
    for r in records:
        hdr = struct.pack('<Q', len(r))
        out += hdr + struct.pack('<I', masked_crc32c(hdr)) + r + struct.pack('<I', masked_crc32c(r))
    return bytes(out)


# ------------------------------------------------------------------ protobuf wire format
def _varint(buf, pos):
    v, shift = 0, 0
    while True:
        b = buf[pos]; pos += 1
        v |= (b & 0x7F) << shift
        if not b & 0x80:
            return v, pos
        shift += 7


def _fields(buf):
    pos = 0
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        field, wire = tag >> 3, tag & 7
        if wire == 0:
            v, pos = _varint(buf, pos)
        elif wire == 1:
            v = buf[pos:pos + 8]; pos += 8
        elif wire == 2:
            n, pos = _varint(buf, pos)
            v = buf[pos:pos + n]; pos += n
        elif wire == 5:
            v = buf[pos:pos + 4]; pos += 4
        else:
            raise ValueError('wire type %d' % wire)
        yield field, wire, v


def decode_feature(buf):
    """Feature -> ('float' | 'int64' | 'bytes', list)."""
    for field, wire, v in _fields(buf):
        if wire != 2:
            continue
        vals = []
        if field == 1:
            return 'bytes', [x for f, w, x in _fields(v) if f == 1]
        if field == 2:
            for f, w, x in _fields(v):
                if f != 1:
                    continue
                if w == 2:
                    vals += list(struct.unpack('<%df' % (len(x) // 4), x))
                else:
                    vals.append(struct.unpack('<f', x)[0])
            return 'float', vals
        if field == 3:
            for f, w, x in _fields(v):
                if f != 1:
                    continue
                if w == 2:
                    p = 0
                    while p < len(x):
                        iv, p = _varint(x, p)
                        vals.append(iv - (1 << 64) if iv >= 1 << 63 else iv)
                else:
                    vals.append(x - (1 << 64) if x >= 1 << 63 else x)
            return 'int64', vals
    return 'none', []


def decode_example(buf):
    """tf.Example -> {name: (kind, values)}."""
    out = {}
    for field, wire, v in _fields(buf):
        if field != 1 or wire != 2:
            continue
        for f2, w2, entry in _fields(v):
            if f2 != 1 or w2 != 2:
                continue
            key, val = None, None
            for f3, w3, x in _fields(entry):
                if f3 == 1:
                    key = bytes(x).decode('utf-8')
                elif f3 == 2:
                    val = x
            if key is not None and val is not None:
                out[key] = decode_feature(val)
    return out


def decode_elwc(buf):
    """ExampleListWithContext -> (context dict, [example dicts])."""
    ctx, examples = {}, []
    for field, wire, v in _fields(buf):
        if wire != 2:
            continue
        if field == 1:
            examples.append(decode_example(v))
        elif field == 2:
            ctx = decode_example(v)
    return ctx, examples


def parse_from_example_list(serialized, list_size, example_spec, context_spec=None):
    """data.py:133-208 for numeric FixedLenFeature specs {name: (width, default)}: returns
    (features {name: nested lists [B][L][width]}, context {name: [B][width]}, sizes, mask)."""
    decoded = [decode_elwc(s) for s in serialized]
    if not list_size:
        list_size = max([len(e) for _, e in decoded] + [1])
    return _pad_parse(decoded, list_size, example_spec, context_spec)


def decode_eie(buf):
    """ExampleInExample (data.py:136-151): a tf.Example whose bytes feature `serialized_context` holds one serialized
    tf.Example and whose bytes feature `serialized_examples` holds one per item -> (context dict, [example dicts]).
    `serialized_context` is a FixedLenFeature([1], string) there: a record without it is an error."""
    outer = decode_example(buf)
    ctx = outer.get('serialized_context', ('none', []))
    if ctx[0] != 'bytes' or len(ctx[1]) != 1:
        raise ValueError('serialized_context must hold exactly one serialized tf.Example')
    exs = outer.get('serialized_examples', ('bytes', []))
    if exs[0] not in ('bytes', 'none'):
        raise ValueError('serialized_examples must be a bytes_list')
    return decode_example(bytes(ctx[1][0])), [decode_example(bytes(e)) for e in exs[1]]


def decode_seq(buf):
    """tf.SequenceExample { Features context = 1; FeatureLists feature_lists = 2 } with FeatureLists { map<string,
    FeatureList> feature_list = 1 }, FeatureList { repeated Feature feature = 1 } -> (context dict, {name: [frames]}),
    a frame = (kind, values)."""
    ctx, lists = {}, {}
    for field, wire, v in _fields(buf):
        if wire != 2:
            continue
        if field == 1:
            ctx.update(decode_example(_ld(1, bytes(v))))          # Features == the payload of Example.features
        elif field == 2:
            for f2, w2, entry in _fields(v):
                if f2 != 1 or w2 != 2:
                    continue
                key, val = None, b''
                for f3, w3, x in _fields(entry):
                    if f3 == 1:
                        key = bytes(x).decode('utf-8')
                    elif f3 == 2:
                        val = x
                if key is None:
                    continue
                lists[key] = [decode_feature(x) for f4, w4, x in _fields(val) if f4 == 1 and w4 == 2]
    return ctx, lists


def _pad_parse(decoded, list_size, example_spec, context_spec):
    feats = {k: [] for k in example_spec}
    ctxs = {k: [] for k in (context_spec or {})}
    sizes, mask = [], []
    for ctx, examples in decoded:
        sizes.append(len(examples))
        mask.append([i < len(examples) for i in range(list_size)])
        for k, (w, d) in example_spec.items():
            rows = []
            for i in range(list_size):
                vals = examples[i].get(k, ('none', []))[1] if i < len(examples) else []
                if len(vals) == 0:
                    vals = [d] * w
                assert len(vals) == w, (k, len(vals), w)
                rows.append([float(x) for x in vals])
            feats[k].append(rows)
        for k, (w, d) in (context_spec or {}).items():
            vals = ctx.get(k, ('none', []))[1]
            if len(vals) == 0:
                vals = [d] * w
            ctxs[k].append([float(x) for x in vals])
    return feats, ctxs, sizes, mask


def parse_from_example_in_example(serialized, list_size, example_spec, context_spec=None):
    """data.py:133-208, 211-380 for numeric FixedLenFeature specs: same outputs as parse_from_example_list."""
    decoded = [decode_eie(s) for s in serialized]
    if not list_size:
        list_size = max([len(e) for _, e in decoded] + [1])
    return _pad_parse(decoded, list_size, example_spec, context_spec)


def parse_from_sequence_example(serialized, list_size, example_spec, context_spec=None):
    """data.py:572-710 for numeric FixedLenFeature specs {name: (width, default)}: every named example feature is
    parsed as a FixedLenSequenceFeature(allow_missing=True) -- a missing feature_list has no frames, a frame must carry
    exactly `width` values (an empty frame is an error, data_test.py:793-819) -- frames beyond a feature's own count
    take its default (:620-633, pad_fn :680-684), the list is truncated / padded to list_size (None: the longest
    feature list of the batch, :636-642), sizes = max over the named features (:701-702)."""
    decoded = [decode_seq(s) for s in serialized]
    counts = [[len(lists.get(k, [])) for k in example_spec] for _, lists in decoded]
    if not list_size:
        list_size = max([max(c + [0]) for c in counts] + [1])
    feats = {k: [] for k in example_spec}
    ctxs = {k: [] for k in (context_spec or {})}
    sizes, mask = [], []
    for (ctx, lists), cnt in zip(decoded, counts):
        n = max(cnt + [0])
        sizes.append(n)
        mask.append([i < n for i in range(list_size)])
        for k, (w, d) in example_spec.items():
            frames = lists.get(k, [])
            rows = []
            for i in range(list_size):
                if i < len(frames):
                    kind, vals = frames[i]
                    if kind == 'bytes' and len(vals):
                        raise TypeError('feature %s: bytes_list where a numeric frame is expected' % k)
                    if len(vals) != w:
                        raise ValueError('feature %s, frame %d: %d values, expected %d' % (k, i, len(vals), w))
                    rows.append([float(x) for x in vals])
                else:
                    rows.append([float(d)] * w)
            for i in range(list_size, len(frames)):               # (TF validates the frames it truncates as well)
                if len(frames[i][1]) != w:
                    raise ValueError('feature %s, frame %d: %d values, expected %d' % (k, i, len(frames[i][1]), w))
            feats[k].append(rows)
        for k, (w, d) in (context_spec or {}).items():
            vals = ctx.get(k, ('none', []))[1]
            if len(vals) == 0:
                vals = [d] * w
            ctxs[k].append([float(x) for x in vals])
    return feats, ctxs, sizes, mask


# ------------------------------------------------------------------ protobuf writer (fixtures)
def _enc_varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _ld(field, payload):
    return _enc_varint((field << 3) | 2) + _enc_varint(len(payload)) + payload


def encode_feature(kind, values, packed=True):
    if kind == 'float':
        if packed:
            inner = _ld(1, struct.pack('<%df' % len(values), *values))
        else:
            inner = b''.join(_enc_varint((1 << 3) | 5) + struct.pack('<f', v) for v in values)
        return _ld(2, inner)
    if kind == 'int64':
        if packed:
            inner = _ld(1, b''.join(_enc_varint(v) for v in values))
        else:
            inner = b''.join(_enc_varint(1 << 3) + _enc_varint(v) for v in values)
        return _ld(3, inner)
    if kind == 'bytes':
        return _ld(1, b''.join(_ld(1, v) for v in values))
    raise ValueError(kind)


def encode_example(features, packed=True):
    """{name: (kind, values)} -> serialized tf.Example."""
    entries = b''.join(_ld(1, _ld(1, k.encode('utf-8')) + _ld(2, encode_feature(kind, vals, packed)))
                       for k, (kind, vals) in features.items())
    return _ld(1, entries)


def encode_elwc(context, examples, packed=True):
    out = b''.join(_ld(1, encode_example(e, packed)) for e in examples)
    if context is not None:
        out += _ld(2, encode_example(context, packed))
    return out


def encode_eie(context, examples, packed=True):
    """data_test.py:568-577: the outer tf.Example with the two bytes features."""
    feats = {'serialized_context': ('bytes', [encode_example(context or {}, packed)]),
             'serialized_examples': ('bytes', [encode_example(e, packed) for e in examples])}
    return encode_example(feats, packed)


def encode_seq(context, feature_lists, packed=True):
    """{name: (kind, values)} context + {name: [(kind, values) per frame]} -> serialized tf.SequenceExample.
    A frame given as None is an empty Feature (no kind set)."""
    out = b''
    if context is not None:
        entries = b''.join(_ld(1, _ld(1, k.encode('utf-8')) + _ld(2, encode_feature(kind, vals, packed)))
                           for k, (kind, vals) in context.items())
        out += _ld(1, entries)
    lists = b''
    for k, frames in feature_lists.items():
        fl = b''.join(_ld(1, b'' if fr is None else encode_feature(fr[0], fr[1], packed)) for fr in frames)
        lists += _ld(1, _ld(1, k.encode('utf-8')) + _ld(2, fl))
    out += _ld(2, lists)
    return out


# ------------------------------------------------------------------ LibSVM (tf_ranking_libsvm.py:137-195)
def load_libsvm_data(text, list_size, num_features):
    qid_to_index, qid_to_ndoc = {}, {}
    feats, labels = [], []
    total = discarded = 0
    for line in text.splitlines():
        tokens = line.split('#')[0].split()
        if not tokens:
            continue
        assert len(tokens) >= 2
        label, qid = float(tokens[0]), tokens[1]
        kv = [t.split(':') for t in tokens[2:]]
        if qid not in qid_to_index:
            qid_to_index[qid] = len(qid_to_index)
            qid_to_ndoc[qid] = 0
            feats.append([[0.0] * num_features for _ in range(list_size)])
            labels.append([-1.0] * list_size)
        total += 1
        b, d = qid_to_index[qid], qid_to_ndoc[qid]
        qid_to_ndoc[qid] += 1
        if d >= list_size:
            discarded += 1
            continue
        for k, v in kv:
            feats[b][d][int(k) - 1] = float(v)
        labels[b][d] = label
    return feats, labels, total, discarded
