"""CPU-side tests: the C-ABI library loads and exports every symbol declared in
include/tfr_hip.h, host-side argument / error conventions, serialisation, and
that the product path refuses to run without a HIP device (no CPU fallback)."""
import ctypes
import os
import re

import pytest
import torch

import ranking_amd as ra
from ranking_amd import _lib, _ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, 'include', 'tfr_hip.h')).read()
    return sorted(set(re.findall(r'^(?:int|long)\s+(tfr_\w+)\s*\(', src, flags=re.M)))


def test_header_and_library_agree():
    lib = _lib.load()
    names = declared_symbols()
    assert names, 'no entry points parsed from include/tfr_hip.h'
    for n in names:
        assert hasattr(lib, n), 'libtfr_hip.so does not export %s' % n
    assert sorted(_lib.EXPORTED_SYMBOLS) == names
    # one version number in three places: the header's constant, what the library was compiled with, what the binding expects
    hdr = open(os.path.join(ROOT, 'include', 'tfr_hip.h')).read()
    want = int(re.search(r'#define\s+TFR_HIP_ABI_VERSION\s+(\d+)', hdr).group(1))
    assert lib.tfr_hip_abi_version() == want == _lib.ABI_VERSION


def test_library_is_in_tree_and_for_gfx950():
    assert _lib.LIB_PATH.startswith(ROOT)
    assert '--offload-arch=gfx950' in _lib.HIPCC_FLAGS
    blob = open(_lib.LIB_PATH, 'rb').read()
    assert b'gfx950' in blob


def test_invalid_arguments_are_rejected_before_any_launch():
    lib = _lib.load()
    # null pointers / bad sizes -> TFR_EINVAL (-1); L > 8192 -> TFR_ETOOLARGE (-2).  No GPU needed.
    assert lib.tfr_sort_ranks_f32(None, None, None, None, 1, 4, None, None, None) == -1
    one = ctypes.c_void_p(16)
    assert lib.tfr_approx_ndcg_f32(one, one, None, one, None, 1, 9000, 0.1, 0, one, one, None, None, None) == -2
    assert lib.tfr_approx_ndcg_f32(one, one, None, one, None, 1, 8, -1.0, 0, one, one, None, None, None) == -1
    assert lib.tfr_approx_ndcg_f32(one, one, None, one, None, 1, 8, 0.1, 3, one, one, None, None, None) == -1
    assert lib.tfr_pairwise_logistic_f32(one, one, None, None, None, 2, 0, 1.5, 0, 0, None, one, 1, 8, 1.0,
                                         None, None, None, None, None) == -1       # smooth_fraction
    assert lib.tfr_pairwise_logistic_f32(one, one, None, None, None, 7, 0, 0.0, 0, 0, None, one, 1, 8, 1.0,
                                         None, None, None, None, None) == -1       # lambda kind
    assert lib.tfr_list_order_i32(None, None, 4, 8, one, one, None) == -1
    # beyond the LDS range a list needs a workspace slot (tfr_list_workspace_bytes): absent / short -> TFR_ETOOLARGE
    assert lib.tfr_list_mle_f32(one, one, None, None, None, 1, 5000, 1.0, one, None, 0, None, 0, None) == -2
    assert lib.tfr_list_mle_f32(one, one, None, None, None, 1, 5000, 1.0, one, None, 7, one, 16 * 8192 - 1, None) == -2
    assert lib.tfr_list_mle_f32(one, one, None, None, None, 1, 9000, 1.0, one, None, 0, one, 1 << 30, None) == -2
    assert lib.tfr_list_mle_f32(one, one, None, None, None, 0, 5000, 1.0, one, None, 0, one, 16 * 8192, None) == 0    # B == 0
    assert lib.tfr_circle_loss_f32(one, one, None, None, 1, 5000, 64.0, 0.25, 1, one, None, None, None, 0, None) == -2
    assert lib.tfr_unique_softmax_f32(one, one, None, None, 1, 5000, 1.0, one, None, None, 0, None) == -2
    assert lib.tfr_neural_sort_loss_f32(0, one, one, None, one, None, 1, 2049, 1.0, one, None, None, 0, None) == -2
    assert lib.tfr_neural_sort_loss_f32(1, one, one, None, None, None, 0, 2049, 1.0, one, None, one, 52 * 4096, None) == 0
    one_topn = (ctypes.c_int32 * 1)(1)
    assert lib.tfr_rank_metric_f32(3, one, one, None, 0, None, None, one, one_topn, 1, 1, 5000, one, one, 0, None, 0, None) == -2
    assert lib.tfr_div_metric_f32(1, one, one, None, 0, None, None, 0.5, one_topn, 1, 1, 5000, 2, one, one, 9, None, 0, None) == -2
    ws = lib.tfr_list_workspace_bytes
    assert [ws(op, 4096) for op in range(5)] == [0] * 5 and ws(5, 2048) == 0 and ws(6, 2048) == 0
    assert [ws(op, 4097) for op in range(7)] == [16 * 8192, 28 * 8192, 28 * 8192, 24 * 8192, 20 * 8192, 32 * 8192, 52 * 8192]
    assert ws(5, 2049) == 32 * 4096 and ws(6, 3000) == 52 * 4096 and ws(0, 8193) == 0 and ws(7, 5000) == 0
    assert lib.tfr_rank_metric_f32(12, one, one, None, 0, None, None, one, (ctypes.c_int32 * 1)(1), 1, 1, 8, one, one, 0, None, 0, None) == -1
    topn = (ctypes.c_int32 * 1)(10)
    assert lib.tfr_ndcg_metric_f32(one, one, None, 0, None, None, one, topn, 9, 1, 8, one, one, 0, None) == -1
    assert lib.tfr_gumbel_sample_f32(one, one, None, None, 0, 0, 1, 0, 8, 1.0, one, None) == -1
    # tower: the two-pass output-layer backward accepts exactly (dy, partial), (NULL, partial), (dy, NULL, pqr)
    ob = lambda dy, partial, pqr: lib.tfr_tower_out_bwd2(one, 8, 4, 8, 0, None, None, None, None, one, one, 1, dy, 8,
                                                         partial, 1, None, pqr, None)
    assert ob(None, None, None) == -1 and ob(one, None, None) == -1
    assert ob(None, one, one) == -1 and ob(one, one, one) == -1
    assert lib.tfr_flatten_row_index(None, 1, 8, one, None) == -1
    assert lib.tfr_flatten_row_index(one, 1, 9000, one, None) == -2         # L > 8192
    assert lib.tfr_flatten_row_index(one, 0, 8, one, None) == 0
    assert lib.tfr_tower_multi_add(None, None, None, 0, None) == 0
    assert lib.tfr_tower_multi_add(None, None, None, 2, None) == -1
    # B == 0 is a no-op
    assert lib.tfr_softmax_loss_f32(one, one, None, None, 0, 0, 0, 0, 0, None, None, 0, 8, 1.0, one, one,
                                    None, None) == 0
    # the reduced-scalar entry points (round 5) need their sum / ticket (/ scratch) buffers
    assert lib.tfr_softmax_loss_sum_f32(one, one, None, None, 0, 0, 0, 0, 0, None, None, 1, 8, 1.0, 0.0, one, one,
                                        None, None, one, one, None) == -1
    assert lib.tfr_pairwise_loss_sum_f32(0, one, one, None, None, None, 0, 0, 0.0, 0, 0, None, None, 1, 8, 1.0,
                                         None, None, None, None, None, None, one, one, 0, None) == -1  # no list_loss_out
    assert lib.tfr_list_mle_sum_f32(one, one, None, None, None, 1, 8, 1.0, one, None, one, None, 0, None, 0, None) == -1
    assert lib.tfr_unique_softmax_sum_f32(one, one, None, None, 1, 8, 1.0, one, None, None, one, None, 0, None) == -1
    assert lib.tfr_pointwise_loss_sum_f32(0, one, one, None, None, None, 1, 8, 1.0, one, None, None, None, None, one,
                                          None) == -1
    with pytest.raises(ValueError):
        _lib.check(-1, 'x')
    with pytest.raises(ValueError):
        _lib.check(-2, 'x')
    with pytest.raises(_lib.TfrHipError):
        _lib.check(719, 'x')


def test_no_cpu_fallback():
    y = torch.tensor([[1., 0.]])
    s = torch.tensor([[0.6, 0.8]])
    for loss in (ra.keras.losses.ApproxNDCGLoss(), ra.keras.losses.SoftmaxLoss(),
                 ra.keras.losses.PairwiseLogisticLoss(), ra.keras.losses.GumbelApproxNDCGLoss(seed=1)):
        with pytest.raises(_lib.TfrHipError):
            loss(y, s)
        with pytest.raises(_lib.TfrHipError):
            loss.loss_and_grad(y, s)
    with pytest.raises(_lib.TfrHipError):
        ra.keras.metrics.NDCGMetric()(y, s)
    with pytest.raises(_lib.TfrHipError):
        ra.utils.sort_by_scores(s, [s])
    with pytest.raises(_lib.TfrHipError):
        ra.losses.make_loss_fn('approx_ndcg_loss')(y, s, {})


def test_product_never_imports_the_oracle():
    import ast
    pkg = os.path.join(ROOT, 'ranking_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                tree = ast.parse(open(os.path.join(dirpath, f)).read())
                for node in ast.walk(tree):
                    names = []
                    if isinstance(node, ast.Import):
                        names = [a.name for a in node.names]
                    elif isinstance(node, ast.ImportFrom):
                        names = [node.module or '']
                    assert not any(n.split('.')[0] == 'oracle' for n in names), (f, names)


def test_keys_and_factories():
    K = ra.keras.losses
    assert K.RankingLossKey.APPROX_NDCG_LOSS == 'approx_ndcg_loss'
    assert set(['pairwise_logistic_loss', 'softmax_loss', 'gumbel_approx_ndcg_loss']) <= set(
        K.RankingLossKey.all_keys())
    assert isinstance(K.get('pairwise_logistic_loss', lambda_weight=K.NDCGLambdaWeight()),
                      K.PairwiseLogisticLoss)
    assert isinstance(K.get('gumbel_approx_ndcg_loss', sample_size=4), K.GumbelApproxNDCGLoss)
    with pytest.raises(ValueError):
        K.get('list_mle_loss_typo')
    with pytest.raises(ValueError):
        K.ApproxNDCGLoss(reduction='mean')
    with pytest.raises(ValueError):
        ra.keras.metrics.get('nope')
    with pytest.raises(ValueError):
        ra.keras.metrics.get(3)
    assert [m.name for m in ra.keras.metrics.default_keras_metrics()][:4] == [
        'metric/ndcg_1', 'metric/ndcg_3', 'metric/ndcg_5', 'metric/ndcg_10']
    # the messages python/losses_test.py:1066-1086,1168-1172 match on
    with pytest.raises(ValueError, match='loss_keys cannot be None or empty.'):
        ra.losses.make_loss_fn([], None)
    with pytest.raises(ValueError, match='loss_keys cannot be None or empty.'):
        ra.losses.make_loss_fn('')
    with pytest.raises(ValueError, match='loss_keys and loss_weights must have the same size.'):
        ra.losses.make_loss_fn(['softmax_loss'], [1.0, 2.0])
    with pytest.raises(ValueError, match='`loss_weights` has to be None when weights are encoded in `loss_keys`'):
        ra.losses.make_loss_fn('softmax_loss:0.5', [1.0])
    with pytest.raises(ValueError, match='`loss_weights` has to be None'):
        ra.losses.make_loss_fn('softmax_loss:0.5,mean_squared_loss:2', [])
    with pytest.raises(ValueError, match='Invalid reduction'):
        ra.losses.make_loss_fn('softmax_loss', reduction='none')
    with pytest.raises(ValueError, match='Invalid loss_key: invalid_key.'):
        ra.losses.make_loss_fn(['invalid_key'])(torch.zeros(1, 2), torch.zeros(1, 2), {})
    with pytest.raises(ValueError):
        ra.losses_impl.DCGLambdaWeight(smooth_fraction=-0.1)          # losses_impl.py:329-331
    with pytest.raises(ValueError):
        ra.losses_impl.ndcg(torch.zeros(1, 2), ranks=torch.ones(1, 2), perm_mat=torch.ones(1, 2, 2))
    assert ra.utils.parse_keys_and_weights('a:0.5, b:2,c') == {'a': 0.5, 'b': 2.0, 'c': 1.0}


def test_config_round_trips():
    K = ra.keras.losses
    lw = K.NDCGLambdaWeight(topn=5, smooth_fraction=0.25)
    for loss in (K.PairwiseLogisticLoss(lambda_weight=lw, temperature=0.5, name='p'),
                 K.SoftmaxLoss(lambda_weight=K.DCGLambdaWeight(normalized=True), ragged=True),
                 K.ApproxNDCGLoss(temperature=0.3),
                 K.GumbelApproxNDCGLoss(sample_size=4, gumbel_temperature=2.0, seed=7)):
        cfg = loss.get_config()
        clone = type(loss).from_config(cfg)
        assert clone.get_config() == cfg
    cfg = lw.get_config()       # keras/losses_test.py:1359-1459 style
    assert cfg['topn'] == 5 and cfg['smooth_fraction'] == 0.25 and cfg['normalized'] is True
    assert cfg['gain_fn'] is ra.keras.utils.pow_minus_1 and cfg['rank_discount_fn'] is ra.keras.utils.log2_inverse
    m = ra.keras.metrics.NDCGMetric(name='n', topn=10)
    assert type(m).from_config(m.get_config()).get_config() == m.get_config()
    s = ra.keras.utils.serialize_keras_object(lw)
    back = ra.keras.utils.deserialize_keras_object(s)
    assert isinstance(back, K.NDCGLambdaWeight) and back._topn == 5


def test_keras_utils_functions():   # keras/utils_test.py
    u = ra.keras.utils
    assert u.identity(3.0) == 3.0
    assert torch.allclose(u.inverse(torch.tensor([1., 2., 0.])), torch.tensor([1., .5, 0.]))
    assert torch.allclose(u.pow_minus_1(torch.tensor([0., 1., 3.])), torch.tensor([0., 1., 7.]))
    assert torch.allclose(u.log2_inverse(torch.tensor([1., 3.])), torch.tensor([1., .5]))
    assert u.is_greater_equal_1(torch.tensor([0.5, 1.0])).tolist() == [False, True]


def test_ragged_to_dense_and_indices():   # utils.py:421-443, 203-356
    l, p, w, m = ra.utils.ragged_to_dense([[1., 0.], [0., 1., 2.]], [[.1, .2], [.3, .4, .5]],
                                          [[1., 2.], [3., 4., 5.]], device=torch.device('cpu'))
    assert l.tolist() == [[1., 0., -1.], [0., 1., 2.]]
    assert p[0, 2].item() == -1e6 and w[0, 2].item() == 0.0
    assert m.tolist() == [[True, True, False], [True, True, True]]
    idx, mask = ra.utils.padded_nd_indices(torch.tensor([[True, True, False], [False, True, False]]))
    assert idx.tolist() == [[0, 1, 0], [1, 1, 1]] and mask.tolist() == [[True, True, False], [True, False, False]]
    org = ra.utils.organize_valid_indices(torch.tensor([[False, True, True]]), shuffle=False)
    assert org.tolist() == [[1, 2, 0]]


def test_flatten_restore_and_tower_shapes():   # keras/layers.py:87-108,195-216 = keras/layers_test.py:25-30,34-90,109-135,156-166
    import math
    L = ra.keras.layers
    ctx = {'c': torch.tensor([[1.], [0.]])}
    ex = {'e': torch.tensor([[[1.], [0.], [-1.]], [[0.], [1.], [0.]]])}
    mask = torch.tensor([[True, True, False], [True, False, False]])
    fc, fe = L.FlattenList()((ctx, ex, mask))
    assert fc['c'].reshape(-1).tolist() == [1., 1., 1., 0., 0., 0.]
    assert fe['e'].reshape(-1).tolist() == [1., 0., 1., 0., 0., 0.]
    out = L.RestoreList()((torch.tensor([1., 2., 3., 4., 5., 6.]), mask))
    e = math.log(1e-10)
    assert torch.allclose(out, torch.tensor([[1., 2., e], [4., e, e]]))
    out = L.RestoreList(by_scatter=True)((torch.tensor([1., 2., 3., 4., 5., 6.]), mask))
    assert torch.allclose(out, torch.tensor([[2., 2., e], [5., e, e]]))
    doc = torch.tensor([1., .5, 2., 0., -1., 0.])                            # layers.py:195-216
    assert torch.allclose(L.RestoreList()((doc, mask)), torch.tensor([[1., .5, e], [0., e, e]]))
    assert torch.allclose(L.RestoreList(by_scatter=True)((doc, mask)),
                          torch.tensor([[1.5, .5, e], [-1. / 3., e, e]]))
    fc, fe = L.FlattenList(circular_padding=False)((ctx, ex, mask))            # keras/layers_test.py:58-80
    assert fc['c'].reshape(-1).tolist() == [1., 1., 1., 0., 0., 0.]
    assert fe['e'].reshape(-1).tolist() == [1., 0., -1., 0., 1., 0.]
    doc2 = doc.reshape(6, 1)                                                    # keras/layers_test.py:116-120,129-135
    assert torch.allclose(L.RestoreList()((doc2, mask)), torch.tensor([[1., .5, e], [0., e, e]]))
    assert torch.allclose(L.RestoreList(by_scatter=True)((doc2, mask)), torch.tensor([[1.5, .5, e], [-1. / 3., e, e]]))
    with pytest.raises(ValueError):
        L.FlattenList()(({}, {}, mask))
    with pytest.raises(ValueError):
        L.FlattenList()((ctx, {}, mask))                                        # keras/layers_test.py:82-90
    with pytest.raises(ValueError):
        L.RestoreList()((torch.zeros(5), mask))
    with pytest.raises(ValueError):
        L.RestoreList()((torch.zeros(4, 2), mask))                              # keras/layers_test.py:156-166
    assert L.FlattenList().get_config() == {'circular_padding': True}           # :92-97, :137-141 (serialisation = config)
    assert L.RestoreList(by_scatter=True).get_config() == {'by_scatter': True}
    t3 = L.create_tower([3, 2, 1], 1, input_dim=1)                              # keras/layers_test.py:25-30
    x3 = torch.tensor([[[1.], [0.], [-1.]], [[0.], [1.], [0.]]])
    t3.eval()
    assert t3(x3).shape == (2, 3, 1)                                            # the last axis is the feature axis
    assert torch.equal(t3(x3).reshape(6, 1), t3(x3.reshape(6, 1)))
    tower = L.create_tower([8, 4], 1, activation=torch.relu, input_dim=3)
    assert tower(torch.zeros(6, 3)).shape == (6, 1)
    scorer = ra.keras.model.DNNScorer(input_dim=2, hidden_layer_dims=[4], output_units=1,
                                      activation=torch.relu, use_batch_norm=False, dropout=0.)
    logits = scorer(ctx, ex, mask)
    assert logits.shape == (2, 3) and logits[0, 2].item() == pytest.approx(e)


def test_integration_doc_names_every_exported_symbol():
    """INTEGRATION.md is the map from C entry points to the reference interfaces they replace: nothing exported
    may be missing from it."""
    from ranking_amd import _io_lib
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'INTEGRATION.md')).read()
    missing = [s for s in list(_lib.EXPORTED_SYMBOLS) + list(_io_lib.EXPORTED_SYMBOLS) if s not in text]
    assert not missing, missing


def integration_stub_namespace():
    """Executes the ctypes stub documented in INTEGRATION.md section 2 (the binding a reference maintainer would
    add) against the in-tree library and returns its namespace."""
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    blocks = re.findall(r'```python\n(.*?)```', text, flags=re.S)
    stub = [b for b in blocks if 'ctypes.CDLL(' in b]
    assert len(stub) == 1, 'INTEGRATION.md must hold exactly one ctypes stub'
    code = stub[0].replace("'ranking_amd/csrc/libtfr_hip.so'", repr(_lib.LIB_PATH))
    _lib.load()                                               # builds the library when it is stale
    ns = {}
    exec(compile(code, 'INTEGRATION.md', 'exec'), ns)
    return ns


def header_arity(name):
    src = open(os.path.join(ROOT, 'include', 'tfr_hip.h')).read()
    m = re.search(r'^(?:int|long)\s+%s\s*\((.*?)\)\s*;' % re.escape(name), src, flags=re.M | re.S)
    assert m, name
    return len([a for a in m.group(1).split(',') if a.strip() and a.strip() != 'void'])


def test_integration_stub_matches_the_header():
    """The documented stub must bind the entry point with the header's arity and the argtypes this repository's own
    binding uses (round 1 shipped a 13-argument stub for a 14-argument function)."""
    ns = integration_stub_namespace()
    fn = ns['lib'].tfr_approx_ndcg_f32
    assert len(fn.argtypes) == header_arity('tfr_approx_ndcg_f32')
    assert list(fn.argtypes) == list(_lib._SIGNATURES['tfr_approx_ndcg_f32'][1])
    # the call inside the stub passes exactly that many arguments
    doc = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    call_args = doc.split('rc = lib.tfr_approx_ndcg_f32(')[1].split(')\n')[0]
    assert len([a for a in call_args.split(',') if a.strip()]) == header_arity('tfr_approx_ndcg_f32')
    # every header entry point has the arity the in-repo binding declares
    for name, (_, argtypes) in _lib._SIGNATURES.items():
        assert len(argtypes) == header_arity(name), name


def test_every_environment_switch_is_documented():
    """DESIGN.md section 8 lists the A/B switches; a switch read by the sources but missing there is undocumented
    behaviour."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    used = set()
    for path in (glob.glob(os.path.join(root, 'ranking_amd', '**', '*.py'), recursive=True) +
                 glob.glob(os.path.join(root, 'ranking_amd', 'csrc', '*.hip')) +
                 glob.glob(os.path.join(root, 'ranking_amd', 'csrc', '*.cpp')) + [os.path.join(root, 'bench.py')]):
        text = open(path).read()
        used |= set(re.findall(r'(?:getenv|env_int\w*|environ\.get)\(\s*[\'"](TFR_[A-Z0-9_]+)[\'"]', text))
    design = open(os.path.join(root, 'DESIGN.md')).read()
    section = design[design.index('## 8. Developer switches'):]
    missing = sorted(v for v in used if v not in section)
    assert not missing, missing


def test_isa_floor_json_describes_the_current_kernels():
    """bench.py's `valu_frac` / `trans_frac` floors come from ranking_amd/csrc/isa_floor.json, derived from the compiled
    sweep loops by tools/isa_floor.py: its fingerprint covers every kernel source, the tool and the issue costs, so a
    changed loop with a stale floor fails here (run `python tools/isa_floor.py`, or __graft_entry__.build())."""
    import json
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import isa_floor
    with open(isa_floor.OUT) as f:
        d = json.load(f)
    assert d['fingerprint'] == isa_floor.fingerprint(), 'isa_floor.json is stale: python tools/isa_floor.py'
    for kind in ('approx_ndcg', 'pairwise'):
        assert 10.0 < d[kind]['trans_cycles_per_64_pairs'] < d[kind]['valu_cycles_per_64_pairs'] < 200.0
        for part in d[kind]['parts'].values():
            assert part['loop_body']['rcp'] >= 2 and part['loop_body']['lds'] >= 1


def test_list_size_limits_in_the_header_are_the_ones_the_launchers_test():
    """the public header said "L <= 1024" where the code took 4096.  The limit is a macro of
    include/tfr_hip.h (one for every entry point since round 5; the LDS ranges beyond which a workspace is needed are
    TFR_LDS_LIST_SIZE_*); every "list_size <= N" statement of the header must name a macro with that value, the
    launchers must compare against the macros (no literal next to TFR_ETOOLARGE), and common.h's TFR_MAX_LIST must
    equal TFR_MAX_LIST_SIZE."""
    import re
    hdr = open(os.path.join(ROOT, 'include', 'tfr_hip.h')).read()
    macros = {m.group(1): int(m.group(2)) for m in re.finditer(r'#define (TFR_MAX_LIST_SIZE\w*) (\d+)', hdr)}
    assert macros == {'TFR_MAX_LIST_SIZE': 8192}, macros
    lds = {m.group(1): int(m.group(2)) for m in re.finditer(r'#define (TFR_LDS_LIST_SIZE\w*) (\d+)', hdr)}
    assert lds == {'TFR_LDS_LIST_SIZE_METRIC': 4096, 'TFR_LDS_LIST_SIZE_LISTWISE': 4096, 'TFR_LDS_LIST_SIZE_NEURAL_SORT': 2048}
    # statements: "list_size <= 8192 (TFR_MAX_LIST_SIZE" -- the number and the macro must agree
    stated = re.findall(r'list_size <= (\d+) \((TFR_MAX_LIST_SIZE\w*)', hdr)
    assert len(stated) >= 3, stated
    for n, name in stated:
        assert macros[name] == int(n), (n, name, macros[name])
    # no other "L <= <number >= 1000>" / "list_size <= <number>" claims without a macro (256 = the fast-path range, not a limit)
    loose = [m.group(0) for m in re.finditer(r'(?:\bL|list_size) <= (\d{4,})(?! \(TFR_MAX_LIST_SIZE)', hdr)]
    assert not loose, loose
    csrc = os.path.join(ROOT, 'ranking_amd', 'csrc')
    for f in sorted(os.listdir(csrc)):
        if not f.endswith(('.hip', '.h', '.cpp')):
            continue
        for ln in open(os.path.join(csrc, f)):
            m = re.search(r'if \(L > (\w+)\) return TFR_ETOOLARGE', ln)
            if m:
                assert m.group(1) in macros or m.group(1) == 'TFR_MAX_LIST', (f, ln.strip())
    common = open(os.path.join(csrc, 'common.h')).read()
    assert int(re.search(r'#define TFR_MAX_LIST (\d+)', common).group(1)) == macros['TFR_MAX_LIST_SIZE']


def test_fp32_dense_argument_checks_and_slab_proposals():
    """tfr_tower_gemm_f32 / tfr_tower_colsum_f32 reject bad arguments before any launch; the slab-count proposals are
    host arithmetic: one resident round of workgroups (four per CU on 256 CUs), at least 16 k tiles per slab, the thin
    weight gradient (output_units <= 4) by runs of rows.  No GPU needed."""
    lib = _lib.load()
    one = ctypes.c_void_p(256)
    g = lib.tfr_tower_gemm_f32
    assert g(one, 8, 1, one, 8, 1, one, 8, -1, 8, 8, None, 1, None, None) == -1          # M < 0
    assert g(one, 8, 1, one, 8, 1, one, 8, 4, 8, 8, None, 0, None, None) == -1           # splits < 1
    assert g(one, 8, 1, one, 8, 1, one, 4, 4, 8, 8, None, 1, None, None) == -1           # ldc < N
    assert g(one, 4, 1, one, 8, 1, one, 8, 4, 8, 8, None, 1, None, None) == -1           # lda < K (k-contiguous A)
    assert g(one, 2, 0, one, 8, 1, one, 8, 4, 8, 8, None, 1, None, None) == -1           # lda < M (row-contiguous A)
    assert g(one, 8, 1, one, 4, 0, one, 8, 4, 8, 8, None, 1, None, None) == -1           # ldb < N (column-contiguous B)
    assert g(None, 8, 1, one, 8, 1, one, 8, 4, 8, 8, None, 1, None, None) == -1          # A missing with K > 0
    assert g(one, 8, 1, one, 8, 1, None, 8, 4, 8, 8, None, 1, None, None) == -1          # C missing
    assert g(one, 8, 1, one, 8, 1, one, 8, 0, 8, 8, None, 1, None, None) == 0            # M == 0: nothing to do
    assert g(one, 8192, 0, one, 512, 0, one, 512, 512, 512, 8192, None, 8, None, None) == -1   # slabs without a workspace
    c = lib.tfr_tower_colsum_f32
    assert c(one, 4, 8, 8, one, one, None) == -1                                          # ldx < N
    assert c(one, 8, 8, 8, None, one, None) == -1                                         # no scratch
    assert c(one, 8, 8, 0, one, one, None) == 0
    s = lib.tfr_tower_gemm_f32_splits
    assert s(512, 512, 409600) == 64             # 16 tiles x 64 slabs = 1 024 workgroups
    assert s(512, 136, 409600) == 128            # 4 x 2 tiles
    assert s(512, 512, 1000) == 4                # at least 16 k tiles (256 rows) per slab
    assert s(512, 512, 100) == 1
    assert s(4096, 4096, 4096) == 1              # already 1 024 tiles
    assert s(1, 512, 409600) == 1024 and s(4, 64, 3000) == 12     # thin weight gradient: rows / 256, capped
    assert s(0, 5, 5) == 1
    r = lib.tfr_tower_colsum_rows
    assert r(0) == 1 and r(1) == 1 and r(257) == 2 and r(409600) == 1024


def test_utils_reference_literals():
    """The deterministic literals of python/utils_test.py for the helpers the path keeps on the host (the shuffled
    variants draw from TF's random stream and are checked structurally elsewhere)."""
    U = ra.utils
    assert U.is_label_valid(torch.tensor([[1.0, 0.0, -1.0]])).tolist() == [[True, True, False]]        # utils_test.py:30-34
    feat = torch.tensor([[[1., 2., 3.], [4., 5., 6.], [7., 8., 9.]], [[10., 20., 30.], [40., 50., 60.], [70., 80., 90.]]])
    got = U.gather_per_row(feat, torch.tensor([[1, 2, 0], [2, 1, 0]]))                                 # utils_test.py:47-62
    assert got.tolist() == [[[4., 5., 6.], [7., 8., 9.], [1., 2., 3.]], [[70., 80., 90.], [40., 50., 60.], [10., 20., 30.]]]
    got = U.gather_per_row(feat, torch.tensor([[2, 0], [1, 0]]))
    assert got.tolist() == [[[7., 8., 9.], [1., 2., 3.]], [[40., 50., 60.], [10., 20., 30.]]]
    ids = torch.tensor([[1, 2, 3], [4, 5, 6]])                                                         # utils_test.py:36-45 (ids for names)
    assert U.gather_per_row(ids, torch.tensor([[1, 2, 0], [2, 1, 0]])).tolist() == [[2, 3, 1], [6, 5, 4]]
    assert U.gather_per_row(ids, torch.tensor([[2, 0], [1, 0]])).tolist() == [[3, 1], [5, 4]]
    is_valid = U.is_label_valid(torch.tensor([[1.0, 0.0, -1.0], [-1.0, 1.0, 2.0]]))                    # utils_test.py:156-165
    assert U.organize_valid_indices(is_valid, shuffle=False).tolist() == [[0, 1, 2], [1, 2, 0]]       # (column of the nd index)
    assert U.reshape_to_2d(torch.tensor([[[1], [2], [3]], [[4], [5], [6]]])).tolist() == [[1, 2, 3], [4, 5, 6]]   # :195-201
    assert U.reshape_to_2d(torch.tensor([1, 2, 3])).tolist() == [[1], [2], [3]]
    idx, mask = U.padded_nd_indices(torch.tensor([[True, True, True], [True, True, False]]), shuffle=False)   # :221-240
    assert idx.tolist() == [[0, 1, 2], [0, 1, 0]] and mask.tolist() == [[True, True, True], [True, True, False]]
    for n, want_i, want_m in ((3, [0, 1, 2], [True] * 3), (2, [0, 1, 0], [True, True, False]), (0, [0, 0, 0], [False] * 3)):
        valid = torch.arange(3).unsqueeze(0) < n                                                        # utils_test.py:203-219
        idx, mask = U.padded_nd_indices(valid, shuffle=False)
        assert idx.tolist() == [want_i] and mask.tolist() == [want_m]
    l, p, w, m = U.ragged_to_dense([[0., 1.], [2., 3., 4.]], [[5., 6.], [7., 8., 9.]], [[1., 1.], [2., 1., 2.]])   # :282-295
    assert l.tolist() == [[0., 1., -1.], [2., 3., 4.]] and p.tolist() == [[5., 6., -1e6], [7., 8., 9.]]
    assert w.tolist() == [[1., 1., 0.], [2., 1., 2.]] and m.tolist() == [[True, True, False], [True, True, True]]
    pk = U.parse_keys_and_weights                                                                       # utils_test.py:297-322
    assert pk('a') == {'a': 1.0} and pk('a :0.9') == {'a': 0.9} and pk('a,b') == {'a': 1., 'b': 1.}
    assert pk('a, b') == {'a': 1., 'b': 1.} and pk('a, b: 2.') == {'a': 1., 'b': 2.}
    assert pk('a:0.1,b:0.9') == {'a': 0.1, 'b': 0.9} and pk('a:0.1, b : 0.9') == {'a': 0.1, 'b': 0.9}


def test_bucket_rank_algorithm_equals_the_counting_ranks():
    """The algorithm of csrc/common.h: wave_rank_by_bucket restated in numpy (fp32 like the kernel): a 64-bucket partition
    of the score range is monotone, so rank = items in earlier buckets + higher scores inside the own bucket; the scan of
    `hmax` entries needs no bound check (what follows a bucket scores lower, then the -inf padding); ties keep equal
    counts and the occupancy fix-up orders them by index; the partition declines on an empty / non-finite range or a
    bucket above 32 items.  Held against the definition (score descending, ties by index) on the score sets that stress
    it; the HIP code itself is checked on the GPU (bit-exact NDCG, bit-identical LambdaRank outputs)."""
    import numpy as np
    rng = np.random.RandomState(0)
    T = 32

    def reference(x):
        n = len(x)
        return np.asarray([sum((x[j] > x[i]) or (x[j] == x[i] and j < i) for j in range(n)) for i in range(n)])

    def bucket_ranks(x):
        x = np.asarray(x, np.float32)
        n = len(x)
        if n == 0:
            return None
        mn, mx = x.min(), x.max()
        with np.errstate(all='ignore'):
            span = np.float32(mx - mn)
            scale = np.float32(64.0) / span
        if not (span > 0) or not np.isfinite(span) or not np.isfinite(scale):
            return None
        b = np.minimum(63, ((mx - x) * scale).astype(np.int32))
        assert all(b[i] <= b[j] for i in range(n) for j in range(n) if x[i] > x[j])      # monotone
        hist = np.bincount(b, minlength=64)
        if hist.max() > T:
            return None
        start = np.concatenate([[0], np.cumsum(hist)[:-1]])
        fill = np.zeros(64, int)
        bx = np.full(n + T, -np.inf, np.float32)
        for p in rng.permutation(n):                              # the order of the LDS atomics is arbitrary
            bx[start[b[p]] + fill[b[p]]] = x[p]
            fill[b[p]] += 1
        hmax = int(hist.max())
        cnt = np.asarray([start[b[p]] + sum(bx[start[b[p]] + j] > x[p] for j in range(-(-hmax // 4) * 4))
                          for p in range(n)])
        occ = np.bincount(cnt, minlength=n)
        return np.asarray([cnt[p] + (sum(x[j] == x[p] for j in range(p)) if occ[cnt[p]] > 1 else 0) for p in range(n)])

    declined = ranked = 0
    for trial in range(400):
        n = int(rng.randint(1, 200))
        kind = trial % 5
        if kind == 0:
            x = rng.randn(n)
        elif kind == 1:
            x = rng.rand(n)
        elif kind == 2:
            x = np.round(rng.randn(n) * 3) / 3                   # heavy ties
        elif kind == 3:
            x = np.concatenate([rng.randn(n - 1) * 1e-3, [1e6]]) if n > 1 else rng.randn(n)      # an outlier
        else:
            x = rng.randn(n) * rng.choice([1e-30, 1.0, 1e30])
        x = x.astype(np.float32)
        got = bucket_ranks(x)
        if got is None:
            declined += 1
            continue
        ranked += 1
        assert np.array_equal(got, reference(x)), (trial, n, kind)
    assert ranked > 150 and declined > 20                          # both outcomes are exercised


def test_tie_seeds_of_losses_and_metrics_host_side():
    """shuffle_ties / seed on the loss and metric objects (round 5): which tie seed a call hands to the kernels.  ListMLE
    shuffles by default like the reference (losses_impl.py:1558-1561), metrics and pairwise losses on request; a fixed
    seed repeats, a fresh one follows torch's host generator (torch.manual_seed restarts the sequence); 0 = index order."""
    L, M, K = ra.losses_impl, ra.metrics_impl, ra.keras.losses
    mle = L.ListMLELoss(None)
    assert mle.shuffle_ties is True and mle.seed is None
    # (round 6, ADVICE r5: a PRIVATE generator keyed on torch.initial_seed() -- the caller's global stream is not consumed,
    # and the sequence restarts when the initial seed changes)
    torch.manual_seed(5); before = torch.rand(1)
    torch.manual_seed(5); a, b = mle._tie_seed(), mle._tie_seed()
    assert torch.equal(torch.rand(1), before)                     # the two tie seeds did not advance the global generator
    torch.manual_seed(6); other = mle._tie_seed()
    torch.manual_seed(5); c = mle._tie_seed()
    assert a != 0 and b != 0 and a != b and a == c and other != a and 0 < a < 2 ** 31
    mle.seed = 9
    assert mle._tie_seed() == 9 == mle._tie_seed()
    mle.seed = 0
    assert mle._tie_seed() == 1                                   # a fixed seed is never the "no shuffle" value
    mle.shuffle_ties = False
    assert mle._tie_seed() == 0
    k = K.get('list_mle_loss', seed=3, shuffle_ties=True)
    cfg = k.get_config()
    assert cfg['seed'] == 3 and cfg['shuffle_ties'] is True and k._loss._tie_seed() == 3
    assert type(k).from_config(cfg)._loss.seed == 3
    assert K.get('list_mle_loss', shuffle_ties=False)._loss._tie_seed() == 0
    for m in (M.NDCGMetric(None, None), M.MRRMetric(None, None), M.PrecisionMetric(None, None), M.AlphaDCGMetric(None, None, seed=4)):
        assert m._tie_seed() == 0                                 # default: index order (the fast NDCG kernels)
        m.shuffle_ties = True
        s1 = m._tie_seed()
        assert s1 != 0 and (s1 == 4 if isinstance(m, M.AlphaDCGMetric) else True)
        m.seed = 12
        assert m._tie_seed() == 12
    pw = L.PairwiseLogisticLoss(None, lambda_weight=K.NDCGLambdaWeight())
    assert pw._tie_seed() == 0
    pw.shuffle_ties, pw.seed = True, 21
    assert pw._tie_seed() == 21
    plain = L.PairwiseLogisticLoss(None)
    plain.shuffle_ties = True
    assert plain._tie_seed() == 0                                 # no lambda weight: no ranks, nothing to shuffle
    from ranking_amd import _ops as ops
    keys = ops.tie_keys(12345, 3, 50)
    assert keys.shape == (3, 50) and int(keys.min()) >= 0 and int(keys.max()) < 2 ** 15
    assert torch.equal(keys, ops.tie_keys(12345, 3, 50)) and not torch.equal(keys, ops.tie_keys(12346, 3, 50))
    assert int(ops.tie_keys(0, 2, 4).abs().sum()) == 0
