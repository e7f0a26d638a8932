"""Input side pinned on the REFERENCE's own expectations, not on the restatement:

  * `python/data_test.py:35-93` -- EXAMPLE_LIST_PROTO_1 / _2, re-encoded here from their text format with the
    pure-Python protobuf writer of oracle/data_ref.py (the text is transcribed field by field below);
  * `python/data_test.py:222-230` -- CONTEXT_FEATURE_SPEC / EXAMPLE_FEATURE_SPEC;
  * `python/data_test.py:296-412, 484-535` -- the values, padding (-1), sizes [2, 1], mask, truncation, shuffle
    and static shapes that `parse_from_example_list` must produce;
  * `examples/tf_ranking_libsvm.py:137-195` -- the LibSVM loader, hand-evaluated on a 7-line file.

  * `python/data_test.py:568-577, 581-643` -- ExampleInExample: the same context / examples wrapped by
    `_example_in_example`, the same expectations;
  * `python/data_test.py:192-220, 716-848` -- SEQ_EXAMPLE_PROTO_1 / _2 and what `parse_from_sequence_example` makes of
    them (values, sizes, mask, large / small list_size, the missing-frame error, a missing feature_list).

Both the oracle (oracle/data_ref.py) and the product (libtfr_io.so through ranking_amd.data) are held to the same
literals.  The string feature "unigrams" (a VarLenFeature -> SparseTensor in the reference) is outside the numeric
subset this repository parses; it is present in the protos so that the parsers have to skip it correctly."""
import numpy as np
import torch

from oracle import data_ref as D
from ranking_amd import data, utils

F32, I64 = torch.float32, torch.int64

# data_test.py:35-68
EXAMPLE_LIST_PROTO_1 = D.encode_elwc(
    {'query_length': ('int64', [3])},
    [{'unigrams': ('bytes', [b'tensorflow']), 'utility': ('float', [0.0])},
     {'unigrams': ('bytes', [b'learning', b'to', b'rank']), 'utility': ('float', [1.0])}])
# data_test.py:70-93
EXAMPLE_LIST_PROTO_2 = D.encode_elwc(
    {'query_length': ('int64', [2])},
    [{'unigrams': ('bytes', [b'gbdt']), 'utility': ('float', [0.0])}])
SERIALIZED = [EXAMPLE_LIST_PROTO_1, EXAMPLE_LIST_PROTO_2]

# data_test.py:222-230 (numeric subset)
CONTEXT_FEATURE_SPEC = {'query_length': data.FixedLenFeature([1], I64, [0])}
EXAMPLE_FEATURE_SPEC = {'utility': data.FixedLenFeature([1], F32, [-1.])}
_SIZE, _MASK = 'example_list_size', 'mask'                       # data_test.py:32-33


def _both(list_size=None, **kw):
    """(oracle result, product result) as comparable nested lists."""
    feats, ctxs, sizes, mask = D.parse_from_example_list(SERIALIZED, list_size, {'utility': (1, -1.0)},
                                                         {'query_length': (1, 0)})
    oracle = {'utility': feats['utility'], 'query_length': ctxs['query_length'], _SIZE: sizes, _MASK: mask}
    got = data.parse_from_example_list(SERIALIZED, list_size=list_size, context_feature_spec=CONTEXT_FEATURE_SPEC,
                                       example_feature_spec=EXAMPLE_FEATURE_SPEC, size_feature_name=_SIZE,
                                       mask_feature_name=_MASK, **kw)
    product = {k: v.tolist() for k, v in got.items()}
    return oracle, product, got


def test_decode_as_serialized_example_list():
    """data_test.py:296-302: one context, two examples, sizes [2]."""
    ctx, examples = D.decode_elwc(EXAMPLE_LIST_PROTO_1)
    assert len(examples) == 2 and ctx['query_length'] == ('int64', [3])
    got = data.parse_from_example_list([EXAMPLE_LIST_PROTO_1], example_feature_spec=EXAMPLE_FEATURE_SPEC,
                                       context_feature_spec=CONTEXT_FEATURE_SPEC, size_feature_name=_SIZE)
    assert got[_SIZE].tolist() == [2]
    assert tuple(got['utility'].shape) == (1, 2, 1) and tuple(got['query_length'].shape) == (1, 1)


def test_parse_from_example_list():
    """data_test.py:304-323."""
    for res in _both()[:2]:
        assert res['query_length'] == [[3], [2]]
        assert res['utility'] == [[[0.], [1.0]], [[0.], [-1.]]]


def test_parse_from_example_list_padding():
    """data_test.py:325-346: list_size 3 > 2 pads with the spec default."""
    for res in _both(3)[:2]:
        assert res['query_length'] == [[3], [2]]
        assert res['utility'] == [[[0.], [1.0], [-1.]], [[0.], [-1.], [-1.]]]


def test_parse_example_list_with_sizes():
    """data_test.py:348-365."""
    for res in _both(3)[:2]:
        assert res[_SIZE] == [2, 1]
        assert res[_MASK] == [[True, True, False], [True, False, False]]


def test_parse_from_example_list_truncate():
    """data_test.py:367-385: list_size 1 keeps the first example; sizes stay the decoded sizes (:202-206)."""
    oracle, product, got = _both(1)
    for res in (oracle, product):
        assert res['query_length'] == [[3], [2]]
        assert res['utility'] == [[[0.]], [[0.]]]
        assert res[_SIZE] == [2, 1]
        assert res[_MASK] == [[True], [True]]
    assert got['query_length'].dtype == I64 and got['utility'].dtype == F32


def test_parse_from_example_list_shuffle():
    """data_test.py:387-412: shuffle, THEN truncate to list_size 1.  With TF's seed=1 the reference keeps
    `learning to rank` (utility 1.) of list 1; the TF stream is not reproducible here (parity unpinned),
    what is pinned: the survivor of list 1 is one of ITS examples -- over draws, both of them, which only a
    shuffle that runs before the truncation can produce -- list 2 keeps its only example, context untouched."""
    utils.set_random_seed(0)
    seen = set()
    for _ in range(64):
        got = data.parse_from_example_list(SERIALIZED, list_size=1, context_feature_spec=CONTEXT_FEATURE_SPEC,
                                           example_feature_spec=EXAMPLE_FEATURE_SPEC, shuffle_examples=True, seed=1,
                                           size_feature_name=_SIZE, mask_feature_name=_MASK)
        assert tuple(got['utility'].shape) == (2, 1, 1)
        assert got['query_length'].tolist() == [[3], [2]]
        assert got['utility'][1].tolist() == [[0.]]
        assert got[_SIZE].tolist() == [2, 1] and got[_MASK].tolist() == [[True], [True]]
        seen.add(got['utility'][0, 0, 0].item())
    assert seen == {0.0, 1.0}
    # the expectation of data_test.py:408-412 is one of the outcomes:
    assert 1.0 in seen
    # padding + shuffle: the valid set is kept, padding stays behind it
    got = data.parse_from_example_list(SERIALIZED, list_size=3, context_feature_spec=CONTEXT_FEATURE_SPEC,
                                       example_feature_spec=EXAMPLE_FEATURE_SPEC, shuffle_examples=True, seed=1,
                                       mask_feature_name=_MASK)
    assert sorted(got['utility'][0, :2, 0].tolist()) == [0., 1.] and got['utility'][0, 2, 0].item() == -1.
    assert got['utility'][1, :, 0].tolist() == [0., -1., -1.]
    assert got[_MASK].tolist() == [[True, True, False], [True, False, False]]


def test_seeded_shuffle_is_a_stream_not_a_constant():
    """`tf.random.uniform(seed=s)` (utils.py:101) advances on every call: same program => same sequence, but
    consecutive batches see different permutations (a re-seeded generator per call would freeze them)."""
    is_valid = torch.ones((1, 16), dtype=torch.bool)
    utils.set_random_seed(0)
    a = [utils.shuffle_valid_indices(is_valid, seed=7).tolist() for _ in range(4)]
    utils.set_random_seed(0)
    b = [utils.shuffle_valid_indices(is_valid, seed=7).tolist() for _ in range(4)]
    assert a == b                                             # reproducible run to run
    assert len({str(x) for x in a}) == 4                      # but not the same draw every call
    utils.set_random_seed(1)
    assert [utils.shuffle_valid_indices(is_valid, seed=7).tolist() for _ in range(4)] != a


def test_parse_from_example_list_static_shape():
    """data_test.py:414-439."""
    for list_size, shape in ((None, (2, 2, 1)), (100, (2, 100, 1)), (1, (2, 1, 1))):
        got = data.parse_from_example_list(SERIALIZED, list_size=list_size, context_feature_spec=CONTEXT_FEATURE_SPEC,
                                           example_feature_spec=EXAMPLE_FEATURE_SPEC)
        assert tuple(got['query_length'].shape) == (2, 1)
        assert tuple(got['utility'].shape) == shape


def test_unpacked_encoding_parses_the_same():
    """proto3 writers pack repeated scalars, proto2 writers do not; both are valid wire images of the same
    message (data.py:59-77 is schema only) and must parse to the data_test.py literals."""
    recs = [D.encode_elwc({'query_length': ('int64', [3])},
                          [{'unigrams': ('bytes', [b'tensorflow']), 'utility': ('float', [0.0])},
                           {'unigrams': ('bytes', [b'learning', b'to', b'rank']), 'utility': ('float', [1.0])}],
                          packed=False),
            D.encode_elwc({'query_length': ('int64', [2])},
                          [{'unigrams': ('bytes', [b'gbdt']), 'utility': ('float', [0.0])}], packed=False)]
    got = data.parse_from_example_list(recs, list_size=3, context_feature_spec=CONTEXT_FEATURE_SPEC,
                                       example_feature_spec=EXAMPLE_FEATURE_SPEC)
    assert got['utility'].tolist() == [[[0.], [1.0], [-1.]], [[0.], [-1.], [-1.]]]
    assert got['query_length'].tolist() == [[3], [2]]


LIBSVM_TEXT = """\
2 qid:10 1:0.5 3:-1.25 # doc a
0 qid:10 2:7
1 qid:20 1:1e-3 2:2 3:3
0 qid:10 3:0.125
3 qid:10 1:9
4 qid:30 2:-0.5
1 qid:20 3:4.5   # trailing comment
"""


def test_libsvm_loader_hand_evaluated(tmp_path):
    """examples/tf_ranking_libsvm.py:137-195 walked by hand on LIBSVM_TEXT with list_size=3, 3 features:
    queries are indexed in order of first appearance (10, 20, 30); a query keeps its first 3 documents (the 4th of
    qid 10 is discarded); absent features are 0; padded labels are -1; text after '#' is ignored."""
    want_feats = np.zeros((3, 3, 3), dtype=np.float32)
    want_labels = -np.ones((3, 3), dtype=np.float32)
    want_feats[0, 0] = [0.5, 0, -1.25]; want_labels[0, 0] = 2
    want_feats[0, 1] = [0, 7, 0];       want_labels[0, 1] = 0
    want_feats[0, 2] = [0, 0, 0.125];   want_labels[0, 2] = 0
    want_feats[1, 0] = [1e-3, 2, 3];    want_labels[1, 0] = 1
    want_feats[1, 1] = [0, 0, 4.5];     want_labels[1, 1] = 1
    want_feats[2, 0] = [0, -0.5, 0];    want_labels[2, 0] = 4
    feats, labels, total, discarded = D.load_libsvm_data(LIBSVM_TEXT, 3, 3)
    assert (np.asarray(feats, dtype=np.float32) == want_feats).all()
    assert (np.asarray(labels, dtype=np.float32) == want_labels).all()
    assert (total, discarded) == (7, 1)
    p = tmp_path / 'tiny.txt'
    p.write_text(LIBSVM_TEXT)
    got_f, got_l = data.load_libsvm_data(str(p), 3, num_features=3)
    assert (got_f.numpy() == want_feats).all() and (got_l.numpy() == want_labels).all()


# ---------------------------------------------------------------- ExampleInExample (data_test.py:568-643)
CONTEXT_1 = {'query_length': ('int64', [3])}                     # data_test.py:95-103
EXAMPLES_1 = [{'unigrams': ('bytes', [b'tensorflow']), 'utility': ('float', [0.0])},             # :105-132
              {'unigrams': ('bytes', [b'learning', b'to', b'rank']), 'utility': ('float', [1.0])}]
CONTEXT_2 = {'query_length': ('int64', [2])}                     # :134-141
EXAMPLES_2 = [{'unigrams': ('bytes', [b'gbdt']), 'utility': ('float', [0.0])}]                  # :143-154
EIE_SERIALIZED = [D.encode_eie(CONTEXT_1, EXAMPLES_1), D.encode_eie(CONTEXT_2, EXAMPLES_2)]


def _eie(list_size=None, **kw):
    feats, ctxs, sizes, mask = D.parse_from_example_in_example(EIE_SERIALIZED, list_size, {'utility': (1, -1.0)},
                                                               {'query_length': (1, 0)})
    oracle = {'utility': feats['utility'], 'query_length': ctxs['query_length'], _SIZE: sizes, _MASK: mask}
    got = data.parse_from_example_in_example(EIE_SERIALIZED, list_size=list_size,
                                             context_feature_spec=CONTEXT_FEATURE_SPEC,
                                             example_feature_spec=EXAMPLE_FEATURE_SPEC, size_feature_name=_SIZE,
                                             mask_feature_name=_MASK, **kw)
    return oracle, {k: v.tolist() for k, v in got.items()}, got


def test_parse_from_example_in_example():
    """data_test.py:581-599."""
    oracle, product, got = _eie()
    assert product['query_length'] == [[3], [2]] and got['query_length'].dtype == I64
    assert product['utility'] == [[[0.], [1.0]], [[0.], [-1.]]]
    assert oracle['utility'] == product['utility'] and oracle['query_length'] == [[3.0], [2.0]]


def test_parse_example_in_example_with_sizes():
    """data_test.py:628-643."""
    oracle, product, _ = _eie(list_size=3)
    assert product[_SIZE] == [2, 1] == oracle[_SIZE]
    assert product[_MASK] == [[True, True, False], [True, False, False]] == oracle[_MASK]
    assert product['utility'] == [[[0.], [1.], [-1.]], [[0.], [-1.], [-1.]]] == oracle['utility']


def test_parse_from_example_in_example_shuffle():
    """data_test.py:601-626: list_size=1 with shuffle_examples keeps ONE of the two examples of list 1 (which one is
    the TF random stream's business: SURVEY 8c) and the only example of list 2."""
    seen = set()
    for seed in range(8):
        got = data.parse_from_example_in_example(EIE_SERIALIZED, list_size=1, context_feature_spec=CONTEXT_FEATURE_SPEC,
                                                 example_feature_spec=EXAMPLE_FEATURE_SPEC, shuffle_examples=True,
                                                 seed=seed)
        assert tuple(got['utility'].shape) == (2, 1, 1) and got['query_length'].tolist() == [[3], [2]]
        assert got['utility'][1].tolist() == [[0.]]
        assert got['utility'][0].tolist() in ([[0.]], [[1.]])
        seen.add(got['utility'][0, 0, 0].item())
    assert seen == {0.0, 1.0}                                  # both outcomes occur over the seeds


def test_make_parsing_fn_eie_and_seq():
    """data_test.py:935-974."""
    fn = data.make_parsing_fn(data.EIE, context_feature_spec=CONTEXT_FEATURE_SPEC,
                              example_feature_spec=EXAMPLE_FEATURE_SPEC)
    got = fn(EIE_SERIALIZED)
    assert got['query_length'].tolist() == [[3], [2]] and got['utility'].tolist() == [[[0.], [1.0]], [[0.], [-1.]]]
    fn = data.make_parsing_fn(data.SEQ, context_feature_spec=CONTEXT_FEATURE_SPEC,
                              example_feature_spec=EXAMPLE_FEATURE_SPEC)
    got = fn(SEQ_SERIALIZED)
    assert got['query_length'].tolist() == [[3], [2]] and got['utility'].tolist() == [[[0.], [1.0]], [[0.], [-1.]]]
    import pytest
    with pytest.raises(ValueError, match='Format non_existing is not supported.'):             # data_test.py:976-980
        data.make_parsing_fn('non_existing', example_feature_spec=EXAMPLE_FEATURE_SPEC)


# ---------------------------------------------------------------- SequenceExample (data_test.py:192-220, 716-848)
SEQ_EXAMPLE_PROTO_1 = D.encode_seq(
    {'query_length': ('int64', [3])},
    {'unigrams': [('bytes', [b'tensorflow']), ('bytes', [b'learning', b'to', b'rank'])],
     'utility': [('float', [0.0]), ('float', [1.0])]})
SEQ_EXAMPLE_PROTO_2 = D.encode_seq(
    {'query_length': ('int64', [2])},
    {'unigrams': [('bytes', [b'gbdt'])], 'utility': [('float', [0.0])]})
SEQ_SERIALIZED = [SEQ_EXAMPLE_PROTO_1, SEQ_EXAMPLE_PROTO_2]


def _seq(serialized, list_size=None, ctx=True):
    feats, ctxs, sizes, mask = D.parse_from_sequence_example(serialized, list_size, {'utility': (1, -1.0)},
                                                             {'query_length': (1, 0)} if ctx else None)
    oracle = {'utility': feats['utility'], _SIZE: sizes, _MASK: mask}
    got = data.parse_from_sequence_example(serialized, list_size=list_size,
                                           context_feature_spec=CONTEXT_FEATURE_SPEC if ctx else None,
                                           example_feature_spec=EXAMPLE_FEATURE_SPEC, size_feature_name=_SIZE,
                                           mask_feature_name=_MASK)
    product = {k: v.tolist() for k, v in got.items()}
    for k in oracle:
        assert oracle[k] == product[k], k
    return product, got


def test_parse_from_sequence_example():
    """data_test.py:718-751: values, static shapes, sizes, mask."""
    product, got = _seq(SEQ_SERIALIZED)
    assert product['query_length'] == [[3], [2]]
    assert product['utility'] == [[[0.], [1.]], [[0.], [-1.]]]
    assert tuple(got['query_length'].shape) == (2, 1) and tuple(got['utility'].shape) == (2, 2, 1)
    assert product[_SIZE] == [2, 1] and product[_MASK] == [[True, True], [True, False]]


def test_parse_from_sequence_example_with_large_and_small_list_size():
    """data_test.py:753-791."""
    product, got = _seq([SEQ_EXAMPLE_PROTO_1], list_size=3)
    assert product['query_length'] == [[3]] and product['utility'] == [[[0.], [1.], [-1.]]]
    assert tuple(got['utility'].shape) == (1, 3, 1)
    product, got = _seq([SEQ_EXAMPLE_PROTO_1], list_size=1)
    assert product['query_length'] == [[3]] and product['utility'] == [[[0.]]]
    assert tuple(got['utility'].shape) == (1, 1, 1) and product[_SIZE] == [2]


def test_parse_from_sequence_example_missing_frame_exception():
    """data_test.py:793-819: `feature { }` inside a feature_list is an error ("values size: 0 but output shape: [1]")."""
    import pytest
    missing_frame = D.encode_seq(None, {'utility': [('float', [0.0]), None]})
    with pytest.raises(ValueError):
        D.parse_from_sequence_example([missing_frame], 2, {'utility': (1, -1.0)})
    with pytest.raises(ValueError, match='length different from its spec'):       # (tf.errors.InvalidArgumentError there)
        data.parse_from_sequence_example([missing_frame], list_size=2, example_feature_spec=EXAMPLE_FEATURE_SPEC)
    with pytest.raises(ValueError):                            # also when the bad frame is one the truncation drops
        data.parse_from_sequence_example([missing_frame], list_size=1, example_feature_spec=EXAMPLE_FEATURE_SPEC)


def test_parse_from_sequence_example_missing_feature_list():
    """data_test.py:821-848: a spec'd feature without a feature_list is all defaults; the dynamic list size is the
    longest list among the NAMED features."""
    proto = D.encode_seq(None, {'utility2': [('float', [0.0])]})
    got = data.parse_from_sequence_example([proto], list_size=2, example_feature_spec=EXAMPLE_FEATURE_SPEC)
    assert tuple(got['utility'].shape) == (1, 2, 1) and got['utility'].tolist() == [[[-1.], [-1.]]]
    spec2 = dict(EXAMPLE_FEATURE_SPEC, utility2=EXAMPLE_FEATURE_SPEC['utility'])
    got0 = data.parse_from_sequence_example([proto], example_feature_spec=spec2, size_feature_name=_SIZE)
    assert tuple(got0['utility'].shape) == (1, 1, 1) and got0['utility'].tolist() == [[[-1.]]]
    assert got0['utility2'].tolist() == [[[0.]]] and got0[_SIZE].tolist() == [1]
    # nothing named is present: the reference pads to max(bounding shapes) = 0 frames; here the list is one default row
    got1 = data.parse_from_sequence_example([proto], example_feature_spec=EXAMPLE_FEATURE_SPEC, size_feature_name=_SIZE)
    assert got1[_SIZE].tolist() == [0] and got1['utility'].tolist() == [[[-1.]]]


def test_sequence_example_refuses_shuffle():
    """data.py:577-579."""
    import pytest
    with pytest.raises(ValueError, match='Shuffling examples is not supported in SequenceExample format.'):
        data.parse_from_sequence_example(SEQ_SERIALIZED, example_feature_spec=EXAMPLE_FEATURE_SPEC,
                                         shuffle_examples=True)


def test_parse_from_tf_example():
    """data_test.py:156-190, 1266-1288: TF_EXAMPLE_PROTO_1 / _2, one item per list, context features in the same proto."""
    protos = [D.encode_example({'query_length': ('int64', [1]), 'unigrams': ('bytes', [b'tensorflow']),
                                'utility': ('float', [0.0])}),
              D.encode_example({'query_length': ('int64', [3]), 'unigrams': ('bytes', [b'learning', b'to', b'rank']),
                                'utility': ('float', [1.0])})]
    got = data.parse_from_tf_example(protos, context_feature_spec=CONTEXT_FEATURE_SPEC,
                                     example_feature_spec=EXAMPLE_FEATURE_SPEC, size_feature_name=_SIZE,
                                     mask_feature_name=_MASK)
    assert got[_SIZE].tolist() == [1, 1] and got[_SIZE].dtype == F32
    assert got[_MASK].tolist() == [[True], [True]]
    assert got['query_length'].tolist() == [[1], [3]] and got['utility'].tolist() == [[[0.]], [[1.]]]
    got = data.parse_from_tf_example([D.encode_example({})], context_feature_spec=CONTEXT_FEATURE_SPEC,
                                     example_feature_spec=EXAMPLE_FEATURE_SPEC)
    assert got['query_length'].tolist() == [[0]] and got['utility'].tolist() == [[[-1.]]]      # the specs' defaults


def test_read_batched_sequence_example_dataset(tmp_path):
    """data_test.py:1148-1186: the two SequenceExamples written to a TFRecord file and read back in one batch of two."""
    path = str(tmp_path / 'seq.tfrecord')
    data.write_tfrecord(path, SEQ_SERIALIZED)
    ds = data.read_batched_sequence_example_dataset(path, 2, 2, CONTEXT_FEATURE_SPEC, EXAMPLE_FEATURE_SPEC,
                                                    num_epochs=1, shuffle=False, prefetch_buffer_size=None)
    batches = list(ds)
    assert len(batches) == 1
    got = batches[0]
    assert got['query_length'].tolist() == [[3], [2]] and got['utility'].tolist() == [[[0.], [1.0]], [[0.], [-1.]]]
