"""Input path: libtfr_io.so / ranking_amd.data against the pure-Python
restatement (oracle/data_ref.py), against the golden fixture cut from the reference's own data
files (tests/golden/elwc_golden.json) and, when /root/reference is present (build container
only), against those files in full.  Integer / byte work: bit-exact."""
import base64
import json
import os
import struct

import numpy as np
import pytest
import torch

from oracle import data_ref as D
from ranking_amd import _io_lib, data

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'elwc_golden.json')))
REF = '/root/reference/tensorflow_ranking/examples/data'
F32 = torch.float32


def test_io_header_and_library_agree():
    import re
    text = open(os.path.join(ROOT, 'include', 'tfr_io.h')).read()
    names = sorted(set(re.findall(r'\b(tfr_io_[a-z0-9_]+)\s*\(', text)))
    lib = _io_lib.load()
    for n in names:
        assert hasattr(lib, n), n
    assert names == sorted(_io_lib.EXPORTED_SYMBOLS)
    assert lib.tfr_io_abi_version() == _io_lib.ABI_VERSION == 2


def test_crc32c_known_answers():
    assert data.crc32c(b'123456789') == 0xE3069283            # the CRC-32C check value
    assert data.crc32c(b'') == 0
    assert data.crc32c(bytes(32)) == 0x8A9136AA               # RFC 3720 B.4: 32 zero bytes
    assert data.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43      # RFC 3720 B.4: 32 0xff bytes
    rng = np.random.RandomState(0)
    lib = _io_lib.load()
    for n in (1, 7, 8, 9, 31, 32, 33, 63, 64, 1000, 4099):
        b = rng.bytes(n)
        assert data.crc32c(b) == D.crc32c(b)
        assert lib.tfr_io_crc32c_portable(b, n) == D.crc32c(b)      # the table path behind the SSE4.2 one
        assert data.masked_crc32c(b) == D.masked_crc32c(b)


def test_tfrecord_round_trip_and_corruption(tmp_path):
    recs = [b'', b'a', bytes(range(256)) * 3, b'x' * 100000]
    p = str(tmp_path / 'a.tfrecord')
    data.write_tfrecord(p, recs)
    assert open(p, 'rb').read() == D.write_tfrecord(recs)
    assert data.read_tfrecord(p) == recs
    raw = bytearray(open(p, 'rb').read())
    raw[16 + 17 + 12 + 5] ^= 0x01                            # flip a payload bit of the third record
    open(p, 'wb').write(bytes(raw))
    with pytest.raises(_io_lib.TfrIoError):
        data.read_tfrecord(p)
    assert len(data.read_tfrecord(p, verify_crc=False)) == 4
    open(p, 'wb').write(bytes(raw[:-3]))                       # truncated tail
    with pytest.raises(_io_lib.TfrIoError):
        data.read_tfrecord(p, verify_crc=False)
    open(p, 'wb').write(b'')
    assert data.read_tfrecord(p) == []


def _spec(names, default=0.0, width=1):
    return {n: data.FixedLenFeature([width], F32, default) for n in names}


def test_golden_records_decode_like_the_oracle():
    recs = [base64.b64decode(s) for s in GOLDEN['records_b64']]
    names = sorted({k for d in GOLDEN['decoded'] for e in d['examples'] for k in e})
    assert 'utility' in names and len(names) > 100           # sparse: only the non-zero features are stored
    spec = _spec([n for n in names if n != 'utility'])
    spec['utility'] = data.FixedLenFeature([1], F32, -1.0)
    for list_size in (None, 3, 9, 12):
        got = data.parse_from_example_list(recs, list_size=list_size, example_feature_spec=spec,
                                           size_feature_name='n', mask_feature_name='mask')
        L = got['utility'].shape[1]
        assert L == (9 if list_size is None else list_size)
        assert got['n'].tolist() == [4, 4, 9]
        for b, d in enumerate(GOLDEN['decoded']):
            for i in range(L):
                assert bool(got['mask'][b, i]) == (i < len(d['examples']))
                for k in names:
                    want = (d['examples'][i].get(k, [None, []])[1] if i < len(d['examples']) else [])
                    want = np.float32(want[0]) if want else np.float32(-1.0 if k == 'utility' else 0.0)
                    assert got[k][b, i, 0].item() == want, (b, i, k)
    # and the oracle's own batch parser agrees wholesale
    ospec = {k: (1, -1.0 if k == 'utility' else 0.0) for k in names}
    feats, _, sizes, mask = D.parse_from_example_list(recs, 6, ospec)
    got = data.parse_from_example_list(recs, list_size=6, example_feature_spec=spec, mask_feature_name='m')
    for k in names:
        assert torch.equal(got[k], torch.tensor(feats[k], dtype=F32))
    assert got['m'].tolist() == mask


def test_synthetic_elwc_all_encodings(tmp_path):
    rng = np.random.RandomState(1)
    records, truth = [], []
    for b in range(7):
        n = int(rng.randint(0, 6))
        exs = []
        for i in range(n):
            e = {'f': ('float', [float(np.float32(rng.randn())) for _ in range(3)]),
                 'label': ('float', [float(rng.randint(0, 5))]),
                 'id': ('int64', [int(rng.randint(-5, 1 << 40))]),
                 'tok': ('bytes', [b'abc', b'd'])}
            if rng.rand() < 0.3:
                del e['f']                                   # absent -> default
            exs.append(e)
        ctx = {'q': ('float', [1.5, -2.5]), 'qlen': ('int64', [int(b)])} if b % 2 == 0 else None
        records.append(D.encode_elwc(ctx, exs, packed=(b % 3 != 0)))
        truth.append((ctx, exs))
    ex_spec = {'f': data.FixedLenFeature([3], F32, 0.25), 'label': data.FixedLenFeature([1], F32, -1.0),
               'id': data.FixedLenFeature([1], torch.int64, 0)}
    cx_spec = {'q': data.FixedLenFeature([2], F32, 9.0), 'qlen': data.FixedLenFeature([1], torch.int64, -7)}
    p = str(tmp_path / 'syn.tfrecord')
    data.write_tfrecord(p, records)
    recs = data.read_tfrecord(p)
    assert recs == records
    got = data.parse_from_example_list(recs, list_size=4, context_feature_spec=cx_spec, example_feature_spec=ex_spec,
                                       size_feature_name='size', mask_feature_name='mask', num_threads=3)
    assert got['f'].shape == (7, 4, 3) and got['id'].dtype == torch.int64 and got['q'].shape == (7, 2)
    for b, (ctx, exs) in enumerate(truth):
        assert got['size'][b].item() == len(exs)
        for i in range(4):
            assert bool(got['mask'][b, i]) == (i < len(exs))
            if i < len(exs):
                f = exs[i].get('f', ('float', [0.25] * 3))[1]
                assert got['f'][b, i].tolist() == [np.float32(x) for x in f]
                assert got['label'][b, i, 0].item() == exs[i]['label'][1][0]
                assert got['id'][b, i, 0].item() == int(np.float32(exs[i]['id'][1][0]))   # via fp32 (documented)
            else:
                assert got['f'][b, i].tolist() == [0.25] * 3 and got['label'][b, i, 0].item() == -1.0
        if ctx is None:
            assert got['q'][b].tolist() == [9.0, 9.0] and got['qlen'][b, 0].item() == -7
        else:
            assert got['q'][b].tolist() == [1.5, -2.5] and got['qlen'][b, 0].item() == b
    # error behaviour
    with pytest.raises(ValueError):                           # wrong width (tf.io.parse_example raises too)
        data.parse_from_example_list(records, list_size=4, example_feature_spec={'f': data.FixedLenFeature([2], F32, 0.)})
    with pytest.raises(ValueError):                           # numeric spec on a bytes feature
        data.parse_from_example_list(records, list_size=4, example_feature_spec={'tok': data.FixedLenFeature([1], F32, 0.)})
    with pytest.raises(ValueError):
        data.parse_from_example_list(records, example_feature_spec={})
    with pytest.raises(_io_lib.TfrIoError):                   # truncated protobuf
        data.parse_from_example_list([records[1][:-2]], list_size=4, example_feature_spec=ex_spec)
    with pytest.raises(ValueError):
        data.make_parsing_fn('tf_example_in_a_trenchcoat', example_feature_spec=ex_spec)      # data.py:911


def test_shuffle_examples_permutes_only_valid_items():
    exs = [{'x': ('float', [float(i)])} for i in range(5)]
    rec = D.encode_elwc(None, exs)
    spec = {'x': data.FixedLenFeature([1], F32, -1.0)}
    got = data.parse_from_example_list([rec] * 4, list_size=8, example_feature_spec=spec, shuffle_examples=True, seed=3)
    x = got['x'][:, :, 0]
    assert (x[:, 5:] == -1).all()
    assert all(sorted(row[:5].tolist()) == [0., 1., 2., 3., 4.] for row in x)


def test_dataset_builder_batches(tmp_path):
    recs = [D.encode_elwc(None, [{'x': ('float', [float(q * 10 + i)])} for i in range(q % 3 + 1)]) for q in range(10)]
    data.write_tfrecord(str(tmp_path / 'p0.tfrecord'), recs[:6])
    data.write_tfrecord(str(tmp_path / 'p1.tfrecord'), recs[6:])
    spec = {'x': data.FixedLenFeature([1], F32, -1.0)}
    ds = data.build_ranking_dataset(str(tmp_path / 'p*.tfrecord'), data.ELWC, 4, None, spec, list_size=3,
                                    mask_feature_name='mask', num_epochs=1, shuffle=False)
    batches = list(ds)
    assert [b['x'].shape[0] for b in batches] == [4, 4, 2]
    allx = torch.cat([b['x'] for b in batches])[:, 0, 0].tolist()
    assert allx == [float(q * 10) for q in range(10)]
    ds = data.build_ranking_dataset(str(tmp_path / 'p*.tfrecord'), data.ELWC, 4, None, spec, list_size=3,
                                    num_epochs=2, shuffle=True, shuffle_seed=5, drop_final_batch=True)
    assert sum(1 for _ in ds) == 4


def test_libsvm_golden_and_synthetic(tmp_path):
    g = GOLDEN['libsvm']
    p = str(tmp_path / 'head.txt')
    open(p, 'w').write(GOLDEN['libsvm_head'])
    feats, labels = data.load_libsvm_data(p, g['list_size'], g['num_features'])
    assert labels.tolist() == g['labels']
    want = torch.zeros_like(feats)
    for b, d, k, v in g['nonzero']:
        want[b, d, k] = v
    assert torch.equal(feats, want)
    text = '2 qid:7 1:0.5 3:-1e-3 # c\n0 qid:8 2:1\n\n1 qid:7 136:2.5\n3 qid:7 1:1\n'
    open(p, 'w').write(text)
    f2, l2 = data.load_libsvm_data(p, 2, 136)
    of, ol, total, disc = D.load_libsvm_data(text, 2, 136)
    assert l2.tolist() == ol and torch.equal(f2, torch.tensor(of, dtype=F32)) and (total, disc) == (4, 1)
    open(p, 'w').write('1 qid:1 137:1\n')
    with pytest.raises(ValueError):
        data.load_libsvm_data(p, 2, 136)


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference data files only exist in the build container')
def test_against_the_reference_data_files_in_full():
    for name in ('train_numerical_elwc.tfrecord', 'vali_numerical_elwc.tfrecord', 'test_numerical_elwc.tfrecord'):
        path = os.path.join(REF, name)
        buf = open(path, 'rb').read()
        want_records = D.read_tfrecord(buf)
        recs = data.read_tfrecord(path)
        assert recs == want_records
        if name.startswith('train'):
            assert len(recs) == GOLDEN['n_records_in_file'] and data.crc32c(buf) == GOLDEN['file_crc32c']
        names = sorted({k for r in recs for e in D.decode_elwc(r)[1] for k in e})
        ospec = {k: (1, -1.0 if k == 'utility' else 0.0) for k in names}
        spec = {k: data.FixedLenFeature([1], F32, d) for k, (w, d) in ospec.items()}
        feats, _, sizes, mask = D.parse_from_example_list(recs, None, ospec)
        got = data.parse_from_example_list(recs, example_feature_spec=spec, size_feature_name='n', mask_feature_name='m')
        assert got['n'].tolist() == sizes and got['m'].tolist() == mask
        for k in names:
            assert torch.equal(got[k], torch.tensor(feats[k], dtype=F32)), k
    for name in ('train.txt', 'vali.txt', 'test.txt'):
        text = open(os.path.join(REF, name)).read()
        of, ol, _, _ = D.load_libsvm_data(text, 10, 136)
        f, l = data.load_libsvm_data(os.path.join(REF, name), 10, 136)
        assert l.tolist() == ol and torch.equal(f, torch.tensor(of, dtype=F32))


def test_synthetic_bench_batch_survives_the_elwc_round_trip(tmp_path):
    """The bench / parity inputs (tests.common.make_batch + U(-1,1) features) written as ELWC
    TFRecords and read back through the native parser are bit-identical: "identical synthetic
    ELWC inputs" (BASELINE.json north_star) is literal."""
    from tests.common import make_batch
    B, L, F = 6, 40, 136
    labels, _ = make_batch(B, L, seed=12)
    g = torch.Generator().manual_seed(12)
    feats = torch.rand((B, L, F), generator=g) * 2 - 1
    records = []
    for b in range(B):
        n = int((labels[b] >= 0).sum())
        exs = [{'x': ('float', feats[b, i].tolist()), 'utility': ('float', [labels[b, i].item()])} for i in range(n)]
        records.append(D.encode_elwc(None, exs))
    p = str(tmp_path / 'syn.tfrecord')
    data.write_tfrecord(p, records)
    spec = {'x': data.FixedLenFeature([F], F32, 0.0), 'utility': data.FixedLenFeature([1], F32, -1.0)}
    got = data.parse_from_example_list(data.read_tfrecord(p), list_size=L, example_feature_spec=spec,
                                       mask_feature_name='mask')
    assert torch.equal(got['utility'][:, :, 0], labels)
    assert torch.equal(got['mask'], labels >= 0)
    valid = (labels >= 0).unsqueeze(-1)
    assert torch.equal(torch.where(valid, got['x'], torch.zeros(())), torch.where(valid, feats, torch.zeros(())))


def test_feature_order_hints_and_fast_paths_never_change_the_result():
    """The parser predicts the spec of the i-th map entry from the previous example and pattern-matches the usual
    byte images (tfr_io.cpp: SpecTable::lookup, decode_feature / map-entry fast paths).  Examples whose features are
    permuted, missing, unknown to the spec, duplicated (last one wins) or written unpacked must decode exactly like
    the pure-Python oracle."""
    rng = np.random.RandomState(7)
    names = ['a', 'bb', 'ccc', '17', '18', 'wide']
    widths = {'a': 1, 'bb': 1, 'ccc': 2, '17': 1, '18': 1, 'wide': 40}        # 'wide': 160-byte payload, 2-byte varints
    records = []
    for b in range(24):
        exs = []
        for i in range(int(rng.randint(1, 7))):
            order = list(names) + ['unknown%d' % rng.randint(3)]
            if rng.rand() < 0.6:
                rng.shuffle(order)
            e = {}
            for k in order:
                if rng.rand() < 0.2:
                    continue                                  # absent -> default
                w = widths.get(k, 1)
                e[k] = ('float', [float(np.float32(rng.randn())) for _ in range(w)])
            exs.append(e)
        rec = D.encode_elwc(None, exs, packed=(b % 2 == 0))
        if b % 5 == 0 and exs and 'a' in exs[0]:              # a duplicated key: the later entry wins
            dup = D._ld(1, D._ld(1, b'a') + D._ld(2, D.encode_feature('float', [123.5])))
            inner = b''.join(D._ld(1, D._ld(1, k.encode()) + D._ld(2, D.encode_feature(kind, vals, b % 2 == 0)))
                             for k, (kind, vals) in exs[0].items()) + dup
            rec = D._ld(1, D._ld(1, inner)) + b''.join(D._ld(1, D.encode_example(e, b % 2 == 0)) for e in exs[1:])
        records.append(rec)
    spec = {k: data.FixedLenFeature([widths[k]], F32, -3.0 + i) for i, k in enumerate(names)}
    ospec = {k: (widths[k], -3.0 + i) for i, k in enumerate(names)}
    feats, _, sizes, mask = D.parse_from_example_list(records, 5, ospec)
    for threads in (1, 3):
        got = data.parse_from_example_list(records, list_size=5, example_feature_spec=spec, size_feature_name='n',
                                           mask_feature_name='m', num_threads=threads)
        for k in names:
            assert torch.equal(got[k], torch.tensor(feats[k], dtype=F32)), k
        assert got['n'].tolist() == sizes and got['m'].tolist() == mask


def test_prefetcher_semantics_on_the_host(tmp_path):
    """data.Prefetcher = dataset.prefetch(buffer_size) (data.py:1015): same batches in the same order, exceptions of
    the producer surface in the consumer, endless iterators stop on close(); build_ranking_dataset honours
    prefetch_buffer_size."""
    import time

    def gen(n):
        for i in range(n):
            time.sleep(0.002)
            yield ({'x': torch.full((2, 3), float(i)), 'meta': 'kept'}, [torch.tensor([i])])
    got = list(data.Prefetcher(gen(7), buffer_size=3))
    assert [b[1][0].item() for b in got] == list(range(7)) and got[3][0]['meta'] == 'kept'
    assert torch.equal(got[5][0]['x'], torch.full((2, 3), 5.0))

    def bad():
        yield torch.zeros(1)
        raise RuntimeError('producer failed')
    p = data.Prefetcher(bad(), 2)
    next(p)
    with pytest.raises(RuntimeError, match='producer failed'):
        next(p)
    with pytest.raises(StopIteration):
        next(p)

    def endless():
        i = 0
        while True:
            yield torch.tensor([i])
            i += 1
    p = data.Prefetcher(endless(), 4)
    assert [next(p).item() for _ in range(3)] == [0, 1, 2]
    p.close()
    assert not p._thread.is_alive()
    # the dataset builder: identical batches with and without the background thread
    rng = np.random.RandomState(3)
    recs = [D.encode_elwc(None, [{'f': ('float', [float(np.float32(rng.randn()))]), 'label': ('float', [1.0])}
                                 for _ in range(int(rng.randint(1, 4)))]) for _ in range(10)]
    path = str(tmp_path / 'p.tfrecord')
    data.write_tfrecord(path, recs)
    spec = {'f': data.FixedLenFeature([1], F32, 0.0), 'label': data.FixedLenFeature([1], F32, -1.0)}
    kw = dict(data_format=data.ELWC, batch_size=4, context_feature_spec=None, example_feature_spec=spec, list_size=3,
              num_epochs=1, shuffle=True, shuffle_seed=5)
    plain = list(data.build_ranking_dataset(path, **kw))
    ahead = data.build_ranking_dataset(path, prefetch_buffer_size=2, **kw)
    assert isinstance(ahead, data.Prefetcher)
    ahead = list(ahead)
    assert len(plain) == len(ahead) == 3
    for a, b in zip(plain, ahead):
        assert a.keys() == b.keys() and all(torch.equal(a[k], b[k]) for k in a)
    with pytest.raises(ValueError):
        next(iter(data.build_ranking_dataset(str(tmp_path / 'none*.tfrecord'), prefetch_buffer_size=2, **kw)))


def test_libsvm_decimal_conversion_is_pythons_float_then_float32(tmp_path):
    """tf_ranking_libsvm.py:160-181 stores float(token) (a double) into a float32 array: decimal -> double -> float.
    The loader reproduces that double rounding (std::from_chars<double> + cast), '+1.5' / '1E5' / '.5' / '5.' forms,
    overflow to inf and denormals; a token float() would reject is an error, as in the reference."""
    pairs = [('0.1', '1e-45'), ('16777217', '0.30000001192092896'), ('+1.5', '1E5'),
             ('7.038531e-26', '3.4028235677973366e38'), ('.5', '5.'), ('-0.0', '1.00000005960464477539')]
    text = '\n'.join('1 qid:1 1:%s 2:%s' % p for p in pairs) + '\n'
    path = str(tmp_path / 't.txt')
    open(path, 'w').write(text)
    f, l = data.load_libsvm_data(path, 8, 2)
    of, ol = D.load_libsvm_data(text, 8, 2)[:2]
    with np.errstate(over='ignore'):
        want = np.asarray(of, dtype=np.float32)
    assert np.array_equal(f.numpy().view(np.uint32), want.view(np.uint32))       # bit for bit, signed zero included
    assert np.array_equal(l.numpy(), np.asarray(ol, dtype=np.float32))
    open(path, 'w').write('1 qid:1 1:abc\n')
    with pytest.raises(_io_lib.TfrIoError):
        data.load_libsvm_data(path, 8, 2)


def test_dataset_streams_through_a_shuffle_buffer(tmp_path):
    """data.py:975-1013: files are read one at a time and records pass a `shuffle_buffer_size` buffer -- every record
    appears exactly once per epoch, a buffer of 1 keeps the order inside a file, a large one mixes files."""
    recs = [D.encode_elwc(None, [{'x': ('float', [float(q)])}]) for q in range(40)]
    for k in range(4):
        data.write_tfrecord(str(tmp_path / ('s%d.tfrecord' % k)), recs[k * 10:(k + 1) * 10])
    spec = {'x': data.FixedLenFeature([1], F32, -1.0)}

    def epoch(**kw):
        ds = data.build_ranking_dataset(str(tmp_path / 's*.tfrecord'), data.ELWC, 8, None, spec, list_size=1,
                                        num_epochs=1, **kw)
        return torch.cat([b['x'] for b in ds])[:, 0, 0].tolist()
    assert epoch(shuffle=False) == [float(q) for q in range(40)]
    big = epoch(shuffle=True, shuffle_seed=1, shuffle_buffer_size=1000)
    assert sorted(big) == [float(q) for q in range(40)] and big != sorted(big)
    assert big == epoch(shuffle=True, shuffle_seed=1, shuffle_buffer_size=1000)          # seeded: reproducible
    assert big != epoch(shuffle=True, shuffle_seed=2, shuffle_buffer_size=1000)
    one = epoch(shuffle=True, shuffle_seed=3, shuffle_buffer_size=1)                        # only the FILE order moves
    assert sorted(one) == [float(q) for q in range(40)]
    for k in range(4):
        block = one[k * 10:(k + 1) * 10]
        assert block == sorted(block) and block[-1] - block[0] == 9.0
    small = epoch(shuffle=True, shuffle_seed=4, shuffle_buffer_size=5)
    assert sorted(small) == [float(q) for q in range(40)]
    # a small buffer cannot move a record far: the i-th output was read no later than position i + buffer
    files_first = [int(v) // 10 for v in small]
    assert len(set(files_first[:10])) <= 2


def test_data_parallel_sharding_of_the_record_stream(tmp_path):
    """shard=(rank, world): the ranks' epochs are disjoint, together they cover the (group-complete part of the)
    data, and every rank gets the same number of records -- tf.distribute's AutoShardPolicy.DATA."""
    recs = [D.encode_elwc(None, [{'x': ('float', [float(q)])}]) for q in range(43)]       # 43 = 3 * 14 + 1
    for k in range(3):
        data.write_tfrecord(str(tmp_path / ('d%d.tfrecord' % k)), recs[k::3])
    spec = {'x': data.FixedLenFeature([1], F32, -1.0)}

    def epoch(shard, **kw):
        ds = data.build_ranking_dataset(str(tmp_path / 'd*.tfrecord'), data.ELWC, 4, None, spec, list_size=1,
                                        num_epochs=1, shard=shard, **kw)
        return torch.cat([b['x'] for b in ds])[:, 0, 0].tolist()
    for kw in (dict(shuffle=False), dict(shuffle=True, shuffle_seed=9, shuffle_buffer_size=16)):
        whole = epoch(None, **kw)
        parts = [epoch((r, 3), **kw) for r in range(3)]
        assert [len(p) for p in parts] == [14, 14, 14]
        assert sorted(sum(parts, [])) == sorted(whole[:42])              # the incomplete last group is dropped
        assert all(parts[r] == whole[r:42:3] for r in range(3))
    with pytest.raises(ValueError):
        epoch((3, 3), shuffle=False)


def _template_batch(seed):
    """Lists whose examples repeat one byte structure (what the example template of tfr_io.cpp replays), salted with
    every way an example can leave it."""
    rng = np.random.RandomState(seed)
    names = ['1', '2', '3', 'lab', 'wide', 'cnt']
    widths = {'1': 1, '2': 1, '3': 1, 'lab': 1, 'wide': 6, 'cnt': 1}
    tricky = [1.0, -0.0, float(np.frombuffer(b'\x0a\x12\x0d\x05', dtype=np.float32)[0]),
              float(np.frombuffer(b'\x12\x05\x0d\x0a', dtype=np.float32)[0]), 3.5e-39]     # payloads that look like tags
    records = []
    for b in range(20):
        exs = []
        for i in range(int(rng.randint(3, 14))):
            e = {}
            for k in ['1', '2', 'skip', '3', 'lab', 'wide']:
                w = widths.get(k, 1)
                vals = [float(np.float32(rng.randn())) if rng.rand() < 0.8 else tricky[rng.randint(len(tricky))]
                        for _ in range(w)]
                e[k] = ('float', vals)
            roll = rng.rand()
            if roll < 0.08:
                del e['2']                                                   # a feature missing: shorter example
            elif roll < 0.16:
                e = {('2' if k == '1' else '1' if k == '2' else k): v for k, v in e.items()}   # same length, keys swapped
            elif roll < 0.24:
                e['skip'] = ('float', [1.0, 2.0])                            # an unnamed feature of another length
            elif roll < 0.32:
                e['cnt'] = ('int64', [int(rng.randint(0, 300))])             # a varint list: not replayable
            elif roll < 0.40:
                e['1'] = ('float', [])                                       # present but empty -> default
            exs.append(e)
        ctx = {'q': ('float', [float(np.float32(rng.randn()))]), 'other': ('float', [float(rng.randn())])}
        rec = D.encode_elwc(ctx, exs, packed=(b % 3 != 0))
        if b % 4 == 1:                                                       # a repeated key inside the repeated structure
            def with_dup(e):
                inner = b''.join(D._ld(1, D._ld(1, k.encode()) + D._ld(2, D.encode_feature(kind, vals, True)))
                                 for k, (kind, vals) in e.items())
                inner += D._ld(1, D._ld(1, b'3') + D._ld(2, D.encode_feature('float', [e['lab'][1][0] + 7.0], True)))
                return D._ld(1, D._ld(1, inner))
            rec = b''.join(with_dup(e) for e in exs) + D._ld(2, D.encode_example(ctx, True))
            for e in exs:
                e['3'] = ('float', [e['lab'][1][0] + 7.0])                   # what the oracle must see: the later entry
        records.append((rec, exs, ctx))
    return names, widths, records


def test_example_template_replay_equals_the_generic_parse():
    """tfr_io.cpp replays the byte structure of the previous example when the unmasked bytes are identical; whatever
    leaves the structure (missing / swapped / unnamed-of-another-length / int64 / empty / unpacked features) must fall
    back.  Checked against the pure-Python oracle with the template on (this process) and off (TFR_IO_TEMPLATE=0 in a
    child process: the switch is read once)."""
    import subprocess
    import sys
    names, widths, records = _template_batch(11)
    spec = {k: data.FixedLenFeature([widths[k]], F32, -2.0 - i) for i, k in enumerate(names)}
    cspec = {'q': data.FixedLenFeature([1], F32, 9.0)}
    ospec = {k: (widths[k], -2.0 - i) for i, k in enumerate(names)}
    recs = [r for r, _, _ in records]
    feats, ctxs, sizes, mask = D.parse_from_example_list(recs, 12, ospec, {'q': (1, 9.0)})

    def counters():
        import ctypes
        a, b = ctypes.c_uint64(0), ctypes.c_uint64(0)
        _io_lib.load().tfr_io_parse_counters(ctypes.byref(a), ctypes.byref(b))
        return a.value, b.value
    before = counters()
    for threads in (1, 2):
        got = data.parse_from_example_list(recs, list_size=12, example_feature_spec=spec, context_feature_spec=cspec,
                                           size_feature_name='n', mask_feature_name='m', num_threads=threads)
        for k in names:
            assert torch.equal(got[k], torch.tensor(feats[k], dtype=F32)), k
        assert torch.equal(got['q'], torch.tensor(ctxs['q'], dtype=F32))
        assert got['n'].tolist() == sizes and got['m'].tolist() == mask
    replayed, walked = (x - y for x, y in zip(counters(), before))
    # both paths ran: a good part of the tf.Example messages replayed the previous structure (every salted example costs
    # two walks: its own and the re-arming one after it), and the salted
    # ones (and every first example) went through the generic walk
    assert replayed > 0.1 * (replayed + walked) and walked > 0.2 * (replayed + walked), (replayed, walked)
    code = ("import os, sys, torch; sys.path.insert(0, %r); os.environ['TFR_IO_TEMPLATE'] = '0'\n"
            "from tests.test_data_cpu import _template_batch, data, F32\n"
            "names, widths, records = _template_batch(11)\n"
            "spec = {k: data.FixedLenFeature([widths[k]], F32, -2.0 - i) for i, k in enumerate(names)}\n"
            "got = data.parse_from_example_list([r for r, _, _ in records], list_size=12, example_feature_spec=spec)\n"
            "torch.save({k: got[k] for k in names}, sys.argv[1])\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for seed in range(20, 32):                                           # more structures, template on
        names, widths, records = _template_batch(seed)
        recs = [r for r, _, _ in records]
        feats, ctxs, sizes, mask = D.parse_from_example_list(recs, 12, ospec, {'q': (1, 9.0)})
        got = data.parse_from_example_list(recs, list_size=12, example_feature_spec=spec, context_feature_spec=cspec,
                                           size_feature_name='n', mask_feature_name='m', num_threads=1 + seed % 3)
        for k in names:
            assert torch.equal(got[k], torch.tensor(feats[k], dtype=F32)), (seed, k)
        assert torch.equal(got['q'], torch.tensor(ctxs['q'], dtype=F32)) and got['n'].tolist() == sizes
    names, widths, records = _template_batch(11)
    feats, ctxs, sizes, mask = D.parse_from_example_list([r for r, _, _ in records], 12, ospec, {'q': (1, 9.0)})
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, 'off.pt')
        subprocess.run([sys.executable, '-c', code, out], check=True, timeout=300)
        off = torch.load(out)
    for k in names:
        assert torch.equal(off[k], torch.tensor(feats[k], dtype=F32)), k


def test_prefetcher_moves_the_slices_of_one_array_with_one_copy(monkeypatch):
    """Per-feature views of the parser's dense array are staged as ONE transfer of their storage and re-cut on the
    other side; lone tensors, small views of a big array and non-tensors take the plain path.  (Host-only: the
    transfer is replaced by a counting clone.)"""
    base = torch.arange(2 * 3 * 5, dtype=torch.float32).reshape(2, 3, 5)
    feats = {'a': base[:, :, 0:1], 'b': base[:, :, 1:4], 'c': base[:, :, 4:5]}
    lone = torch.arange(6, dtype=torch.int32).reshape(2, 3)
    big = torch.zeros(1000)
    batch = ({**feats, 'n': lone, 'tiny': big[3:5], 'tiny2': big[7:9], 'name': 'x'}, lone.clone())
    calls = []

    def counting_transfer(self, t):
        calls.append(tuple(t.shape))
        return t.clone()
    monkeypatch.setattr(data.Prefetcher, '_transfer', counting_transfer)
    pf = data.Prefetcher(iter([batch]), buffer_size=1, device='cpu')
    got = list(pf)[0]
    assert (30,) in calls and calls.count((30,)) == 1                      # the whole [2, 3, 5] storage, once
    assert (2, 3, 1) not in calls and (2, 3, 3) not in calls                # no per-slice transfer
    assert calls.count((2,)) == 2                                           # the two small views of `big`: plain path
    for k in feats:
        assert torch.equal(got[0][k], feats[k]) and got[0][k].data_ptr() != feats[k].data_ptr()
    assert got[0]['a'].untyped_storage().data_ptr() == got[0]['b'].untyped_storage().data_ptr()
    assert torch.equal(got[0]['n'], lone) and torch.equal(got[1], lone) and got[0]['name'] == 'x'
    assert torch.equal(got[0]['tiny'], big[3:5])


def test_bf16_rounding_of_the_parser_is_torchs_and_the_devices():
    """tfr_io_f32_to_bf16: round to nearest even with NaN kept -- bit for bit torch's .to(bfloat16) (and
    v_cvt_pk_bf16_f32, which the GPU test of the ingest path checks) on ties, subnormals, the overflow to infinity,
    signed zeros, infinities and NaNs."""
    lib = _io_lib.load()
    rng = np.random.RandomState(5)
    bits = rng.randint(0, 2 ** 32, size=200000, dtype=np.uint64).astype(np.uint32)
    special = np.asarray([0x00000000, 0x80000000, 0x3f800000, 0x3f808000, 0x3f818000, 0x3f807fff, 0x3f808001,
                          0x7f7fffff, 0x7f7f8000, 0x7f7f7fff, 0xff7fffff, 0x7f800000, 0xff800000, 0x7fc00000,
                          0x7f800001, 0xffc12345, 0x00000001, 0x00008000, 0x00018000, 0x007fffff, 0x807fffff],
                         dtype=np.uint32)
    src = np.concatenate([bits, special]).view(np.float32)
    dst = np.empty(src.shape, dtype=np.uint16)
    lib.tfr_io_f32_to_bf16(src.ctypes.data, dst.ctypes.data, src.size)
    want = torch.from_numpy(src.copy()).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    nan = np.isnan(src)
    assert np.array_equal(dst[~nan], want[~nan])
    got_nan = dst[nan]
    assert np.all((got_nan & 0x7f80) == 0x7f80) and np.all((got_nan & 0x007f) != 0)       # still NaN
    assert np.array_equal(got_nan & 0x8000, (src.view(np.uint32)[nan] >> 16).astype(np.uint16) & 0x8000)


def test_bf16_parse_equals_the_rounded_fp32_parse():
    """example_dtype=bfloat16 (tfr_io_parse_elwc_batch_bf16) == the fp32 parse rounded afterwards, for every path of
    the parser (template replay and generic walk, packed and unpacked lists, defaults, padding, truncation, threads);
    float32_features come back unrounded; sizes / mask / context features are untouched."""
    names, widths, records = _template_batch(3)
    spec = {n: data.FixedLenFeature([widths[n]], F32, -2.5 if n == 'lab' else 0.1) for n in names if n != 'cnt'}
    cspec = {'q': data.FixedLenFeature([1], F32, 0.0)}
    recs = [r for r, _, _ in records]
    for list_size in (None, 5, 16):
        for threads in (1, 3):
            ref = data.parse_from_example_list(recs, list_size=list_size, context_feature_spec=cspec,
                                               example_feature_spec=spec, size_feature_name='n',
                                               mask_feature_name='m', num_threads=threads)
            got = data.parse_from_example_list(recs, list_size=list_size, context_feature_spec=cspec,
                                               example_feature_spec=spec, size_feature_name='n',
                                               mask_feature_name='m', num_threads=threads,
                                               example_dtype=torch.bfloat16, float32_features=('lab',))
            assert set(got) == set(ref)
            for k in spec:
                if k == 'lab':
                    assert got[k].dtype == torch.float32 and torch.equal(got[k].view(torch.int32), ref[k].view(torch.int32))
                else:
                    assert got[k].dtype == torch.bfloat16 and got[k].shape == ref[k].shape
                    assert torch.equal(got[k].view(torch.int16), ref[k].to(torch.bfloat16).view(torch.int16)), k
            for k in ('q', 'n', 'm'):
                assert torch.equal(got[k], ref[k])
    # all features as bf16 (no side columns), through make_parsing_fn
    fn = data.make_parsing_fn(data.ELWC, list_size=7, example_feature_spec=spec, example_dtype=torch.bfloat16)
    ref = data.make_parsing_fn(data.ELWC, list_size=7, example_feature_spec=spec)(recs)
    got = fn(recs)
    for k in spec:
        assert torch.equal(got[k].view(torch.int16), ref[k].to(torch.bfloat16).view(torch.int16))


def test_bf16_parse_argument_errors_and_shuffle():
    names, widths, records = _template_batch(4)
    recs = [r for r, _, _ in records]
    spec = {'1': data.FixedLenFeature([1], F32, 0.0), 'cnt': data.FixedLenFeature([1], torch.int64, 0)}
    with pytest.raises(ValueError, match='only float features'):
        data.parse_from_example_list(recs, example_feature_spec=spec, example_dtype=torch.bfloat16)
    with pytest.raises(ValueError, match='not an example feature'):
        data.parse_from_example_list(recs, example_feature_spec=spec, float32_features=('nope',))
    with pytest.raises(ValueError, match='example_dtype'):
        data.parse_from_example_list(recs, example_feature_spec=spec, example_dtype=torch.float16)
    # an int64 feature may ride along as a float32 side feature
    got = data.parse_from_example_list(recs, example_feature_spec=spec, example_dtype=torch.bfloat16,
                                       float32_features=('cnt',))
    ref = data.parse_from_example_list(recs, example_feature_spec=spec)
    assert got['cnt'].dtype == torch.int64 and torch.equal(got['cnt'], ref['cnt'])
    # shuffle_examples permutes the bf16 features and the side features with the same permutation
    spec2 = {'1': data.FixedLenFeature([1], F32, 0.0), 'lab': data.FixedLenFeature([1], F32, -1.0)}
    # (the op seed's stream advances from call to call like TF's: compare the PAIRS against the unshuffled parse)
    a = data.parse_from_example_list(recs, example_feature_spec=spec2, mask_feature_name='m')
    b = data.parse_from_example_list(recs, list_size=6, example_feature_spec=spec2, shuffle_examples=True, seed=9,
                                     example_dtype=torch.bfloat16, float32_features=('lab',), mask_feature_name='m')
    assert b['1'].shape == (len(recs), 6, 1) and b['lab'].shape == (len(recs), 6, 1)
    moved = 0
    for r in range(len(recs)):
        n = int(a['m'][r].sum())
        pairs = [(a['1'][r, j, 0].to(torch.bfloat16).view(torch.int16).item(), a['lab'][r, j, 0].view(torch.int32).item())
                 for j in range(n)]
        for i in range(min(n, 6)):
            assert b['m'][r, i]
            pair = (b['1'][r, i, 0].view(torch.int16).item(), b['lab'][r, i, 0].view(torch.int32).item())
            assert pair in pairs
            moved += pair != pairs[i]
    assert moved > 0
    # the C entry refuses inconsistent side-column arguments
    lib = _io_lib.load()
    ex_names, ex_arr, _keep = data._spec_array(spec2)
    ptrs, lens = data._record_arrays(recs)
    out = np.empty((len(recs), 4, 2), dtype=np.uint16)
    cols = np.asarray([5], dtype=np.int32)
    side = np.empty((len(recs), 4, 1), dtype=np.float32)
    rc = lib.tfr_io_parse_elwc_batch_bf16(ptrs, lens.ctypes.data, len(recs), 4, ex_arr, 2, None, 0, out.ctypes.data,
                                          None, None, None, 1, cols.ctypes.data, 1, side.ctypes.data)
    assert rc == -1
    rc = lib.tfr_io_parse_elwc_batch_bf16(ptrs, lens.ctypes.data, len(recs), 4, ex_arr, 2, None, 0, None,
                                          None, None, None, 1, None, 0, None)
    assert rc == -1


def _random_lists(seed, n_lists=24):
    """Per list: a context and 0..9 examples over features a[1] float, b[3] float, c[1] int64, lab[1] float -- with
    missing features, features the spec does not name and bytes features to skip."""
    rng = np.random.RandomState(seed)
    out = []
    for _ in range(n_lists):
        exs = []
        for _ in range(int(rng.randint(0, 10))):
            e = {}
            if rng.rand() < 0.9:
                e['a'] = ('float', [float(np.float32(rng.randn()))])
            if rng.rand() < 0.9:
                e['b'] = ('float', [float(np.float32(v)) for v in rng.randn(3)])
            if rng.rand() < 0.8:
                e['c'] = ('int64', [int(rng.randint(-5, 1000))])
            e['lab'] = ('float', [float(rng.randint(0, 5))])
            if rng.rand() < 0.3:
                e['tok'] = ('bytes', [b'x' * int(rng.randint(0, 5)), b'yz'])
            if rng.rand() < 0.3:
                e['other'] = ('float', [1.0, 2.0])
            exs.append(e)
        ctx = {'q': ('float', [float(np.float32(rng.randn()))])} if rng.rand() < 0.8 else {}
        if rng.rand() < 0.5:
            ctx['n'] = ('int64', [int(rng.randint(0, 50))])
        out.append((ctx, exs))
    return out


_RSPEC = {'a': (1, 0.5), 'b': (3, -2.0), 'c': (1, 7.0), 'lab': (1, -1.0)}
_RCTX = {'q': (1, 0.25), 'n': (1, -3.0)}


def _product_spec(spec):
    return {k: data.FixedLenFeature([w], torch.int64 if k in ('c', 'n') else F32, d) for k, (w, d) in spec.items()}


@pytest.mark.parametrize('fmt', ['eie', 'seq'])
def test_eie_and_seq_parsers_equal_the_oracle_on_random_lists(fmt):
    """tfr_io_parse_batch(EIE / SEQ) == the pure-Python restatement (oracle/data_ref.py) on random lists: packed and
    unpacked encodings, missing / unnamed / bytes features, int64 values, padding, truncation, dynamic list size, one
    and several threads, fp32 and bf16 example features; sizes, mask and context features included."""
    lists = _random_lists(17 if fmt == 'eie' else 18)
    ex_spec, cx_spec = _product_spec(_RSPEC), _product_spec(_RCTX)
    for packed in (True, False):
        if fmt == 'eie':
            recs = [D.encode_eie(ctx, exs, packed) for ctx, exs in lists]
            oracle_fn, product_fn = D.parse_from_example_in_example, data.parse_from_example_in_example
        else:
            recs = []
            for ctx, exs in lists:                                # a SequenceExample: every feature_list as long as ITS values go
                fl = {}
                for name in ('a', 'b', 'c', 'lab', 'tok', 'other'):
                    frames = []
                    for e in exs:
                        if name not in e:
                            break                                 # (a feature_list has no holes: stop at the first gap)
                        frames.append(e[name])
                    if frames or name == 'lab':
                        fl[name] = frames
                recs.append(D.encode_seq(ctx if ctx else None, fl, packed))
            oracle_fn, product_fn = D.parse_from_sequence_example, data.parse_from_sequence_example
        for list_size in (None, 4, 12):
            feats, ctxs, sizes, mask = oracle_fn(recs, list_size, _RSPEC, _RCTX)
            for threads in (1, 3):
                got = product_fn(recs, list_size=list_size, context_feature_spec=cx_spec, example_feature_spec=ex_spec,
                                 size_feature_name='n_items', mask_feature_name='m', num_threads=threads)
                for k in _RSPEC:
                    assert got[k].tolist() == feats[k], (fmt, packed, list_size, k)
                for k in _RCTX:
                    assert got[k].tolist() == ctxs[k], (fmt, packed, list_size, k)
                assert got['n_items'].tolist() == sizes and got['m'].tolist() == mask
            fspec = {k: v for k, v in ex_spec.items() if k != 'c'}
            ref = product_fn(recs, list_size=list_size, example_feature_spec=fspec)
            b16 = product_fn(recs, list_size=list_size, example_feature_spec=fspec, example_dtype=torch.bfloat16,
                             float32_features=('lab',))
            assert torch.equal(b16['lab'], ref['lab'])
            for k in ('a', 'b'):
                assert torch.equal(b16[k].view(torch.int16), ref[k].to(torch.bfloat16).view(torch.int16))


def test_eie_and_seq_error_behaviour():
    lists = _random_lists(19, n_lists=6)
    ex_spec = _product_spec(_RSPEC)
    eie = [D.encode_eie(ctx, exs) for ctx, exs in lists]
    # truncated protobufs -> TfrIoError (corrupt), never a crash
    for cut in (1, 3, 7, 20):
        for rec in eie:
            if len(rec) > cut:
                try:
                    data.parse_from_example_in_example([rec[:-cut]], list_size=4, example_feature_spec=ex_spec)
                except (_io_lib.TfrIoError, ValueError):
                    pass
    # an ExampleInExample without its context feature (FixedLenFeature([1], string) in the reference: required)
    no_ctx = D.encode_example({'serialized_examples': ('bytes', [D.encode_example({'a': ('float', [1.0])})])})
    with pytest.raises(ValueError, match='serialized_context'):
        data.parse_from_example_in_example([no_ctx], list_size=2, example_feature_spec=ex_spec)
    with pytest.raises(ValueError):
        D.parse_from_example_in_example([no_ctx], 2, _RSPEC)
    two_ctx = D.encode_example({'serialized_context': ('bytes', [b'', b'']), 'serialized_examples': ('bytes', [])})
    with pytest.raises(ValueError):
        data.parse_from_example_in_example([two_ctx], list_size=2, example_feature_spec=ex_spec)
    # an empty serialized example is an example of defaults and counts towards the list size
    rec = D.encode_eie({}, [{}, {'a': ('float', [2.0])}])
    got = data.parse_from_example_in_example([rec], example_feature_spec=ex_spec, size_feature_name='n')
    assert got['n'].tolist() == [2] and got['a'].tolist() == [[[0.5], [2.0]]] and got['lab'].tolist() == [[[-1.0], [-1.0]]]
    # SequenceExample: a frame of the wrong width, a bytes frame under a numeric spec, a repeated key (later entry wins)
    bad = D.encode_seq(None, {'b': [('float', [1.0, 2.0, 3.0]), ('float', [1.0])]})
    with pytest.raises(ValueError, match='length different'):
        data.parse_from_sequence_example([bad], example_feature_spec=ex_spec)
    bad = D.encode_seq(None, {'a': [('bytes', [b'zz'])]})
    with pytest.raises(ValueError, match='bytes_list'):
        data.parse_from_sequence_example([bad], example_feature_spec=ex_spec)
    first = D.encode_seq(None, {'a': [('float', [1.0]), ('float', [2.0]), ('float', [3.0])]})
    second = D.encode_seq(None, {'a': [('float', [9.0])]})
    dup = first + second                                          # concatenated messages merge: feature_lists twice
    got = data.parse_from_sequence_example([dup], list_size=3, example_feature_spec={'a': ex_spec['a']},
                                           size_feature_name='n')
    feats, _, sizes, _ = D.parse_from_sequence_example([dup], 3, {'a': _RSPEC['a']})
    assert got['a'].tolist() == feats['a'] == [[[9.0], [0.5], [0.5]]] and got['n'].tolist() == sizes == [1]
    for cut in (1, 2, 5, 9):
        try:
            data.parse_from_sequence_example([first[:-cut]], list_size=3, example_feature_spec={'a': ex_spec['a']})
        except (_io_lib.TfrIoError, ValueError):
            pass
    # the C entry validates its arguments
    lib = _io_lib.load()
    ex_names, ex_arr, _keep = data._spec_array(ex_spec)
    ptrs, lens = data._record_arrays(eie)
    out = np.empty((len(eie), 2, 6), dtype=np.float32)
    args = (ptrs, lens.ctypes.data, len(eie), 2, ex_arr, len(ex_names), None, 0)
    assert lib.tfr_io_parse_batch(7, *args, out.ctypes.data, None, None, None, None, 1, None, 0, None) == -1
    assert lib.tfr_io_parse_batch(1, *args, None, None, None, None, None, 1, None, 0, None) == -1
    assert lib.tfr_io_parse_batch(1, *args, out.ctypes.data, out.ctypes.data, None, None, None, 1, None, 0, None) == -1
    assert lib.tfr_io_parse_batch(1, *args, out.ctypes.data, None, None, None, None, 1, None, 0, None) == 0
    assert lib.tfr_io_max_list_size(9, ptrs, lens.ctypes.data, len(eie), None, 0) == -1
    assert lib.tfr_io_max_list_size(2, ptrs, lens.ctypes.data, len(eie), None, 0) == -1
    assert lib.tfr_io_max_list_size(1, ptrs, lens.ctypes.data, len(eie), None, 0) == max(len(e) for _, e in lists)


def test_parsers_survive_the_sanitizer_fuzzer(tmp_path):
    """tools/io_fuzz.cpp: libtfr_io's source built under -fsanitize=address,undefined and fed valid records of the four
    formats (and LibSVM text, TFRecord framing) with random truncations, bit flips, splices and deletions.  Any error code
    is fine; an out-of-bounds access or undefined behaviour aborts the fuzzer."""
    import shutil
    import subprocess
    cxx = shutil.which('g++')
    if cxx is None:
        pytest.skip('no g++')
    exe = str(tmp_path / 'io_fuzz')
    build = subprocess.run([cxx, '-O1', '-g', '-std=c++17', '-fsanitize=address,undefined', '-fno-sanitize-recover=all',
                            '-pthread', '-I', os.path.join(ROOT, 'include'), os.path.join(ROOT, 'tools', 'io_fuzz.cpp'),
                            os.path.join(ROOT, 'ranking_amd', 'csrc', 'tfr_io.cpp'), '-o', exe],
                           capture_output=True, text=True)
    if build.returncode != 0 and ('asan' in build.stderr or 'ubsan' in build.stderr or 'sanitize' in build.stderr):
        pytest.skip('this toolchain has no sanitizer runtime: %s' % build.stderr[-200:])
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe, '15000', '11'], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, (run.stdout[-500:], run.stderr[-3000:])
    assert 'io_fuzz: 15000 rounds' in run.stdout
    parsed = int(run.stdout.split('rounds,')[1].split('batches parsed')[0])
    assert parsed > 1000                                         # the fuzzer does reach the accepting paths


def test_simple_dataset_builder_formats_and_bf16_features(tmp_path):
    """keras.pipeline.SimpleDatasetBuilder(data_format=..., example_dtype=bfloat16): the example features arrive as
    bfloat16 (rounded like the fp32 parse rounds afterwards), label and sample weight stay float32; the three list
    formats give the same batches for the same lists (host only)."""
    import ranking_amd as tfr
    P = tfr.keras.pipeline
    rng = np.random.RandomState(3)
    lists = []
    for _ in range(12):
        n = int(rng.randint(2, 7))
        lists.append([{'x': ('float', [float(np.float32(v)) for v in rng.randn(4)]),
                       'utility': ('float', [float(rng.randint(0, 5))]), 'w': ('float', [float(np.float32(rng.rand()))])}
                      for _ in range(n)])
    files = {}
    for fmt, enc in ((data.ELWC, lambda exs: D.encode_elwc(None, exs)), (data.EIE, lambda exs: D.encode_eie({}, exs)),
                     (data.SEQ, lambda exs: D.encode_seq(None, {k: [e[k] for e in exs] for k in ('x', 'utility', 'w')}))):
        path = str(tmp_path / ('%s.tfrecord' % fmt))
        data.write_tfrecord(path, [enc(exs) for exs in lists])
        files[fmt] = path
    ex_spec = {'x': data.FixedLenFeature([4], F32, 0.0)}
    label_spec = ('utility', data.FixedLenFeature([1], F32, -1.0))
    weight_spec = ('w', data.FixedLenFeature([1], F32, 1.0))
    batches = {}
    for fmt in files:
        for dt in (torch.float32, torch.bfloat16):
            h = P.DatasetHparams(train_input_pattern=files[fmt], valid_input_pattern=files[fmt], train_batch_size=4,
                                 valid_batch_size=4, list_size=6)
            db = P.SimpleDatasetBuilder({}, ex_spec, 'mask', label_spec, h, sample_weight_spec=weight_spec,
                                        data_format=fmt, example_dtype=dt)
            import itertools
            out = list(itertools.islice(db.build_valid_dataset(), 3))          # (the dataset repeats: num_epochs=None)
            assert len(out) == 3
            for feats, label, weight in out:
                assert feats['x'].dtype == dt and label.dtype == torch.float32 and weight.dtype == torch.float32
                assert tuple(feats['x'].shape) == (4, 6, 4) and tuple(label.shape) == (4, 6)
            batches[(fmt, dt)] = out
    ref = batches[(data.ELWC, torch.float32)]
    for key, out in batches.items():
        for (fa, la, wa), (fb, lb, wb) in zip(ref, out):
            assert torch.equal(la, lb) and torch.equal(wa, wb) and torch.equal(fa['mask'], fb['mask']), key
            want = fa['x'] if key[1] == torch.float32 else fa['x'].to(torch.bfloat16)
            assert torch.equal(want.view(torch.int32 if key[1] == torch.float32 else torch.int16),
                               fb['x'].view(torch.int32 if key[1] == torch.float32 else torch.int16)), key
