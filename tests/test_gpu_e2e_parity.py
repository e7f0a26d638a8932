"""End-to-end "NDCG@10 parity" (the second half of BASELINE.json's metric; VERDICT r2 "Next round" #3).

The same synthetic ELWC file is parsed by the product (libtfr_io) and by the pure-Python oracle parser, the two copies
must be bit-identical, and then two models are trained from ONE initialisation on the SAME batches for N steps:
  (a) the product path: DNNScorer (bf16 fused MFMA tower, BatchNorm, FlattenList folded into the input cast) + the
      fused HIP loss (loss_and_grad) + plain SGD on the GPU;
  (b) the oracle path: keras/layers.py:26-77 restated op by op in fp32 torch on the CPU (oracle.create_tower_train,
      flatten_list / restore_list) + the oracle loss + autograd + the same SGD.
Asserted: the loss trajectories agree step by step and the validation NDCG@10 (inference mode: moving BatchNorm
statistics) agrees at the end, within bounds that are what bf16 operands cost (8 mantissa bits on activations and
weights, fp32 accumulation): relative loss difference <= 5e-3 at every step and |NDCG@10 difference| <= 5e-3 (measured:
7.6e-4 / 3.2e-4 for ApproxNDCG, 9.4e-5 / 7.1e-4 for the groupwise run); both are printed and recorded in the parity-margin
table of the pytest log.
The groupwise variant does the same for BASELINE config 5 (group_size 2, GumbelApproxNDCG with the uniform draws
injected, group indices without the TRAIN-mode shuffle so that both sides see the same groups).
"""
import math

import pytest
import torch

from oracle import data_ref as D
from oracle import tfr_ref as R
from tests.margins import record_margin

pytestmark = pytest.mark.gpu
DEV = 'cuda'
BN_EPS = 1e-3


def ra():
    import ranking_amd
    return ranking_amd


def _synthetic_elwc(path, n_lists, L, F, seed):
    """Lists of ceil(L/2)..L examples, features U(-1, 1), graded labels 0..4 from a noisy non-linear teacher."""
    from ranking_amd import data
    g = torch.Generator().manual_seed(seed)
    gt = torch.Generator().manual_seed(1234)
    w1 = torch.randn(F, 16, generator=gt) / math.sqrt(F)
    w2 = torch.randn(16, generator=gt)
    recs = []
    for _ in range(n_lists):
        n = int(torch.randint((L + 1) // 2, L + 1, (1,), generator=g))
        x = torch.rand(n, F, generator=g) * 2 - 1
        s = torch.tanh(3.0 * (x @ w1)) @ w2 + 0.3 * torch.randn(n, generator=g)
        lab = torch.clamp(torch.round(0.7 * s + 2.0), 0, 4)
        recs.append(D.encode_elwc(None, [{'x': ('float', x[i].tolist()), 'utility': ('float', [lab[i].item()])}
                                         for i in range(n)]))
    data.write_tfrecord(path, recs)
    return recs


def _parse_both(path, L, F):
    """The product parser's tensors, checked bit for bit against the oracle parser."""
    from ranking_amd import data
    spec = {'x': data.FixedLenFeature([F], torch.float32, 0.0), 'utility': data.FixedLenFeature([1], torch.float32, -1.0)}
    recs = data.read_tfrecord(path)
    got = data.parse_from_example_list(recs, list_size=L, example_feature_spec=spec, mask_feature_name='mask')
    feats, _, _, mask = D.parse_from_example_list(recs[:32], L, {'x': (F, 0.0), 'utility': (1, -1.0)})
    assert torch.equal(got['x'][:32], torch.tensor(feats['x'], dtype=torch.float32))
    assert torch.equal(got['utility'][:32, :, 0], torch.tensor(feats['utility'], dtype=torch.float32)[:, :, 0])
    assert torch.equal(got['mask'][:32], torch.tensor(mask))
    return got['x'], got['utility'][:, :, 0].contiguous(), got['mask']


class _OracleTower:
    """create_tower (keras/layers.py:26-77) in fp32 torch ops on the CPU: Dense -> BatchNormalization -> ReLU per hidden
    layer, Dense(output_units); training = batch statistics + moving-average update, inference = moving statistics."""

    def __init__(self, fused, momentum):
        self.W = [w.detach().cpu().t().clone().requires_grad_(True) for w in fused.weights] + \
                 [fused.out_weight.detach().cpu().t().clone().requires_grad_(True)]
        self.b = [b.detach().cpu().clone().requires_grad_(True) for b in fused.biases] + \
                 [fused.out_bias.detach().cpu().clone().requires_grad_(True)]
        self.gamma = [g.detach().cpu().clone().requires_grad_(True) for g in fused.gammas]
        self.beta = [b.detach().cpu().clone().requires_grad_(True) for b in fused.betas]
        self.mm = [torch.zeros(w.shape[1]) for w in self.W[:-1]]
        self.mv = [torch.ones(w.shape[1]) for w in self.W[:-1]]
        self.momentum = momentum

    def params(self):
        return self.W + self.b + self.gamma + self.beta

    def __call__(self, x, training):
        if training:
            with torch.no_grad():                           # moving statistics (Keras: biased batch variance)
                h = x
                for i in range(len(self.gamma)):
                    z = h @ self.W[i] + self.b[i]
                    mean, var = z.mean(0), z.var(0, unbiased=False)
                    self.mm[i] = self.mm[i] * self.momentum + mean * (1 - self.momentum)
                    self.mv[i] = self.mv[i] * self.momentum + var * (1 - self.momentum)
                    h = torch.relu((z - mean) * torch.rsqrt(var + BN_EPS) * self.gamma[i] + self.beta[i])
            return R.create_tower_train(x, self.W, self.b, self.gamma, self.beta, activation=torch.relu, epsilon=BN_EPS)
        h = x
        for i in range(len(self.gamma)):
            z = h @ self.W[i] + self.b[i]
            h = torch.relu((z - self.mm[i]) * torch.rsqrt(self.mv[i] + BN_EPS) * self.gamma[i] + self.beta[i])
        return h @ self.W[-1] + self.b[-1]


def _sgd(params, lr):
    with torch.no_grad():
        for p in params:
            if p.grad is not None:
                p -= lr * p.grad
                p.grad = None


def _report(name, la, lb, na, nb, loss_bar, ndcg_bar):
    la, lb = torch.tensor(la, dtype=torch.float64), torch.tensor(lb, dtype=torch.float64)
    rel = ((la - lb).abs() / torch.clamp(lb.abs(), min=1e-3))
    record_margin('%s: loss trajectory, max relative difference bf16 fused vs fp32 oracle' % name, rel.max().item(), loss_bar)
    record_margin('%s: validation NDCG@10, |bf16 fused - fp32 oracle|' % name, abs(na - nb), ndcg_bar)
    print('\n[%s] loss first %.5f / %.5f  last %.5f / %.5f  max rel diff %.2e (bar %.0e); val NDCG@10 %.5f / %.5f (bar %.0e)'
          % (name, la[0], lb[0], la[-1], lb[-1], rel.max().item(), loss_bar, na, nb, ndcg_bar))
    assert rel.max().item() <= loss_bar, (name, rel.max().item())
    assert abs(na - nb) <= ndcg_bar, (name, na, nb)
    assert lb[-8:].mean() < lb[:8].mean() - 0.02 * abs(lb[:8].mean()), 'the oracle run did not learn: %r' % lb      # a real training run


@pytest.mark.parametrize('loss_name', ['softmax', 'approx_ndcg'])
def test_dnn_scorer_training_matches_the_fp32_oracle(tmp_path, loss_name):
    """Config 2 / 4 shape scaled down: 136-512-512-512-1 scorer, list_size 100, 64 lists per step, 120 SGD steps."""
    tfr = ra()
    L, F, B, steps, momentum = 100, 136, 64, 120, 0.9
    lr = 0.05 if loss_name == 'approx_ndcg' else 5e-4        # the softmax loss of a list scales with the sum of its labels (~150)
    _synthetic_elwc(str(tmp_path / 'train.tfrecord'), 512, L, F, seed=1)
    _synthetic_elwc(str(tmp_path / 'valid.tfrecord'), 256, L, F, seed=2)
    x, y, m = _parse_both(str(tmp_path / 'train.tfrecord'), L, F)
    vx, vy, vm = _parse_both(str(tmp_path / 'valid.tfrecord'), L, F)
    torch.manual_seed(0)
    scorer = tfr.keras.model.DNNScorer(input_dim=F, hidden_layer_dims=[512, 512, 512], output_units=1,
                                       activation=torch.relu, use_batch_norm=True, batch_norm_moment=momentum, dropout=0.0,
                                       compute_dtype=torch.bfloat16).to(DEV)
    oracle = _OracleTower(scorer._tower, momentum)
    K = tfr.keras.losses
    mine = K.SoftmaxLoss() if loss_name == 'softmax' else K.ApproxNDCGLoss()
    theirs = R.SoftmaxLoss() if loss_name == 'softmax' else R.ApproxNDCGLoss(temperature=0.1)
    la, lb = [], []
    for s in range(steps):
        lo = (s * B) % (x.shape[0] - B + 1)
        xb, yb, mb = x[lo:lo + B], y[lo:lo + B], m[lo:lo + B]
        # (a) product
        scorer.train()
        logits = scorer({}, {'x': xb.to(DEV)}, mb.to(DEV))
        value, dlogits = mine.loss_and_grad(yb.to(DEV), logits.detach())
        logits.backward(dlogits)
        _sgd(list(scorer.parameters()), lr)
        la.append(value.item())
        # (b) oracle
        _, flat = R.flatten_list(None, xb, mb)
        ol = R.restore_list(oracle(flat, True), mb)
        ov = R.keras_loss_call(theirs, yb, ol)
        ov.backward()
        _sgd(oracle.params(), lr)
        lb.append(ov.item())
    scorer.eval()
    with torch.no_grad():
        vl = scorer({}, {'x': vx.to(DEV)}, vm.to(DEV))
        na, _ = tfr.metrics_impl.NDCGMetric(None, 10).compute(vy.to(DEV), vl)
        _, vflat = R.flatten_list(None, vx, vm)
        nb, _ = R.NDCGMetric(topn=10).compute(vy, R.restore_list(oracle(vflat, False), vm))
    _report('e2e DNNScorer + %s' % loss_name, la, lb, na.mean().item(), nb.mean().item(), loss_bar=5e-3, ndcg_bar=5e-3)


def test_groupwise_scorer_training_matches_the_fp32_oracle(tmp_path):
    """Config 5 shape: groupwise scorer (group_size 2) 2F-512-512-512-2 + GumbelApproxNDCG (S = 8, draws injected),
    list_size 50, 64 lists per step, 100 SGD steps."""
    tfr = ra()
    from ranking_amd import model as gmodel
    L, F, B, S, steps, lr, momentum = 50, 136, 64, 8, 100, 0.05, 0.9
    _synthetic_elwc(str(tmp_path / 'train.tfrecord'), 384, L, F, seed=3)
    _synthetic_elwc(str(tmp_path / 'valid.tfrecord'), 256, L, F, seed=4)
    x, y, m = _parse_both(str(tmp_path / 'train.tfrecord'), L, F)
    vx, vy, vm = _parse_both(str(tmp_path / 'valid.tfrecord'), L, F)
    torch.manual_seed(0)
    tower = tfr.keras.layers.create_tower([512, 512, 512], 2, activation=torch.relu, use_batch_norm=True,
                                          batch_norm_moment=momentum, dropout=0.0, input_dim=2 * F,
                                          compute_dtype=torch.bfloat16)
    gw = gmodel.GroupwiseScorer(gmodel.FusedGroupScoreFn(tower), group_size=2).to(DEV)
    oracle = _OracleTower(tower, momentum)
    mine = tfr.keras.losses.GumbelApproxNDCGLoss(sample_size=S, gumbel_temperature=1.0, temperature=0.1, seed=3)
    sampler = R.GumbelSampler(sample_size=S, temperature=1.0)
    gu = torch.Generator().manual_seed(77)
    la, lb = [], []
    for s in range(steps):
        lo = (s * B) % (x.shape[0] - B + 1)
        xb, yb, mb = x[lo:lo + B], y[lo:lo + B], m[lo:lo + B]
        u = torch.rand((B, S, L), generator=gu)
        gw.train()
        logits = gw({}, {'x': xb.to(DEV)}, mb.to(DEV), shuffle=False)
        value, dlogits = mine.loss_and_grad(yb.to(DEV), logits.detach(), None, uniform=u.to(DEV))
        logits.backward(dlogits)
        _sgd(list(gw.parameters()), lr)
        la.append(value.item())
        ol = R.groupwise_logits(lambda g_: oracle(g_.reshape(g_.shape[0], -1), True), xb, mb, 2)
        ov = R.keras_loss_call(R.ApproxNDCGLoss(temperature=0.1), yb, ol, gumbel_sampler=sampler, uniform=u)
        ov.backward()
        _sgd(oracle.params(), lr)
        lb.append(ov.item())
    gw.eval()
    with torch.no_grad():
        vl = gw({}, {'x': vx.to(DEV)}, vm.to(DEV), shuffle=False)
        na, _ = tfr.metrics_impl.NDCGMetric(None, 10).compute(vy.to(DEV), vl)
        nb, _ = R.NDCGMetric(topn=10).compute(
            vy, R.groupwise_logits(lambda g_: oracle(g_.reshape(g_.shape[0], -1), False), vx, vm, 2))
    _report('e2e groupwise scorer + GumbelApproxNDCG', la, lb, na.mean().item(), nb.mean().item(), loss_bar=5e-3, ndcg_bar=5e-3)
