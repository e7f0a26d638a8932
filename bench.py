#!/usr/bin/env python
"""bench.py -- lists/sec of the ranking-loss hot path on MI355X.

Headline (BASELINE.json `metric`): ranked lists/sec, loss forward+backward,
ApproxNDCG (temperature 0.1) at list_size=200, 16384 lists per GPU per step
(SURVEY.md 8d row H), synthetic inputs already resident in HBM.  A "step" is one
pass of the hot path over one batch: one fused kernel launch that produces the
per-list loss AND d loss / d logits, plus the [B]-vector dot that reduces the
loss to a scalar.

    python bench.py --gpus N --steps K --warmup W [--workload NAME]

For N > 1 the driver launches one rank per GPU with torch.distributed.run; lists
shard across ranks with no data-path collective (the loss path has no exchange
step: SURVEY.md 8e) -> "scaling": "weak".  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)

WORKLOADS = {
    # name: (B per GPU, L, description, algorithmic HBM bytes per list)
    'approx_ndcg': (16384, 200, 'ApproxNDCGLoss(T=0.1) loss fwd+bwd, B=16384/GPU, L=200 (SURVEY 8d row H)',
                    lambda L: 12 * L + 12),
    'approx_ndcg_l1000': (512, 1000, 'ApproxNDCGLoss(T=0.1) loss fwd+bwd, B=512/GPU, L=1000 (config 4 shard)',
                          lambda L: 12 * L + 12),
    'pairwise_lambda': (4096, 200, 'PairwiseLogisticLoss+NDCGLambdaWeight loss fwd+bwd, B=4096, L=200 (config 3)',
                        lambda L: 12 * L + 12),
    'softmax': (4096, 100, 'SoftmaxLoss loss fwd+bwd, B=4096, L=100 (config 2 loss part)',
                lambda L: 12 * L + 12),
    'gumbel_approx_ndcg': (512, 50, 'GumbelApproxNDCGLoss(S=8) loss fwd+bwd, B=512/GPU, L=50 (config 5 loss part)',
                           lambda L: 12 * L + 12),
    'ndcg_metric': (16384, 200, 'NDCG@{1,3,5,10,all} metric, B=16384, L=200',
                    lambda L: 8 * L + 6 * 4),
    # end-to-end data-parallel training steps: scorer fwd/bwd (bf16 MFMA GEMMs) + fused loss +
    # ONE all-reduce of the flat gradient bucket + optimizer.
    'e2e_softmax': (4096, 100, 'config 2: DNNScorer 136-512-512-512-1 bf16 + SoftmaxLoss, 4096 lists/GPU, '
                               'L=100, SGD, 1 all-reduce/step', lambda L: 0),
    'e2e_pairwise_lambda': (4096, 200, 'config 3 end-to-end: DNNScorer 136-512-512-512-1 bf16 + '
                                       'PairwiseLogisticLoss(NDCGLambdaWeight), 4096 lists/GPU, L=200, SGD, '
                                       '1 all-reduce/step', lambda L: 0),
    'e2e_approx_ndcg_l1000': (512, 1000, 'config 4: DNNScorer 136-512-512-512-1 bf16 + ApproxNDCGLoss, '
                                         '512 lists/GPU, L=1000, 1 all-reduce/step', lambda L: 0),
    'e2e_groupwise_gumbel': (512, 50, 'config 5: groupwise scorer (group_size=2) 272-512-512-512-2 bf16 + '
                                      'GumbelApproxNDCGLoss(S=8), 512 lists/GPU, L=50, 1 all-reduce/step',
                             lambda L: 0),
}

MFMA_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md: dense bf16 MFMA peak (2.5 PFLOP/s)


def e2e_flops_per_list(workload, L):
    """Algorithmic scorer flops per list, fwd + bwd (3 x 2 x MACs)."""
    if workload == 'e2e_groupwise_gumbel':
        macs = 272 * 512 + 512 * 512 * 2 + 512 * 2          # per group; L groups per list
    else:
        macs = 136 * 512 + 512 * 512 * 2 + 512              # per document
    return 6.0 * macs * L


def make_inputs(B, L, seed, device):
    from ranking_amd.synthetic import make_batch
    labels, logits = make_batch(B, L, seed)
    return labels.to(device), logits.to(device)


def build_step(workload, labels, logits, dropout=0.0, use_graph=False):
    """Returns (step_fn, kernel_fn) -- kernel_fn launches only the dominant kernel."""
    from ranking_amd import _ops
    from ranking_amd.keras import losses as K
    from ranking_amd import metrics_impl
    if workload.startswith('approx_ndcg'):
        loss = K.ApproxNDCGLoss()
        B = labels.shape[0]
        scale = torch.full((B,), 1.0 / B, dtype=torch.float32, device=labels.device)
        # the dominant kernel alone, in the configuration the step runs it: with the longest-first launch order
        # when the step computes one (two small launches that the step pays for and this timing leaves out)
        order = _ops._auto_order(labels, None, None, 192)
        return (lambda: loss.loss_and_grad(labels, logits),
                lambda: _ops.approx_ndcg(logits, labels, None, scale, 0.1, 0, True,
                                         balance=order if order is not None else False))
    if workload == 'pairwise_lambda':
        loss = K.PairwiseLogisticLoss(lambda_weight=K.NDCGLambdaWeight())
        return (lambda: loss.loss_and_grad(labels, logits)), None
    if workload == 'softmax':
        loss = K.SoftmaxLoss()
        return (lambda: loss.loss_and_grad(labels, logits)), None
    if workload == 'gumbel_approx_ndcg':
        loss = K.GumbelApproxNDCGLoss(seed=1)
        return (lambda: loss.loss_and_grad(labels, logits)), None
    if workload == 'ndcg_metric':
        m = metrics_impl.NDCGMetric(None, None)
        return (lambda: m.compute_multi(labels, logits, None, None, [1, 3, 5, 10, None])), None
    if workload.startswith('e2e_'):
        return build_e2e_step(workload, labels, dropout, use_graph), None
    raise ValueError(workload)


def build_e2e_step(workload, labels, dropout=0.0, use_graph=False):
    """One data-parallel training step: features [B, L, 136] ~ U(-1, 1) resident in HBM."""
    import ranking_amd as ra
    from ranking_amd import distributed as D
    dev = labels.device
    B, L = labels.shape
    g = torch.Generator(device=dev).manual_seed(1234 + int(os.environ.get('RANK', '0')))
    feats = torch.rand((B, L, 136), generator=g, device=dev) * 2 - 1
    mask = labels >= 0
    torch.manual_seed(0)                                  # identical replicas on every rank
    if workload == 'e2e_groupwise_gumbel':
        from ranking_amd import model as gmodel
        tower = ra.keras.layers.create_tower([512, 512, 512], 2, activation=torch.relu, use_batch_norm=True,
                                             dropout=dropout, input_dim=272, compute_dtype=torch.bfloat16)

        def group_score_fn(ctx, group_features):
            x = group_features['x']
            return tower(x.reshape(x.shape[0], -1))
        gw = gmodel.GroupwiseScorer(group_score_fn, group_size=2).to(dev)
        gw.add_module('tower', tower)
        gw.to(dev)
        scorer = gw
        gw.train()
        gidx = gw.group_indices(mask)                         # index plumbing: once per batch, outside the graph
        run_scorer = lambda: gw({}, {'x': feats}, mask, group_indices=gidx)
        loss = ra.keras.losses.GumbelApproxNDCGLoss(seed=1)
    else:
        scorer = ra.keras.model.DNNScorer(input_dim=136, hidden_layer_dims=[512, 512, 512], output_units=1,
                                          activation=torch.relu, use_batch_norm=True, dropout=dropout,
                                          compute_dtype=torch.bfloat16).to(dev)
        run_scorer = lambda: scorer({}, {'x': feats}, mask)
        if workload == 'e2e_softmax':
            loss = ra.keras.losses.SoftmaxLoss()
        elif workload == 'e2e_pairwise_lambda':
            loss = ra.keras.losses.PairwiseLogisticLoss(lambda_weight=ra.keras.losses.NDCGLambdaWeight())
        else:
            loss = ra.keras.losses.ApproxNDCGLoss()
    scorer.train()
    bucket = D.FlatGradBucket(scorer.parameters(), n_scalars=2)
    if not os.environ.get('TFR_NO_INPLACE_GRADS'):
        bucket.attach(scorer)
    lr = 0.01
    _, world = D.world()
    params = [p for p in scorer.parameters() if p.requires_grad]
    flat_params = None

    def fwd_bwd():
        bucket.zero()
        logits = run_scorer()
        value, dlogits = loss.loss_and_grad(labels, logits.detach())
        logits.backward(dlogits)                           # scorer backward, grads land in the flat bucket
        return value

    grad_views, off = [], 0
    for p in params:
        grad_views.append(bucket.flat[off:off + p.numel()].view_as(p))
        off += p.numel()

    def sgd():
        with torch.no_grad():                              # SGD on the fp32 master weights: one multi-tensor launch
            torch._foreach_add_(params, grad_views, alpha=-lr)

    def eager_step():
        value = fwd_bwd()
        s = bucket.all_reduce(torch.stack([value, value.new_tensor(1.0)]), average=True)
        sgd()
        return s[0] / max(world, 1)

    if not use_graph:
        return eager_step

    # hipGraph capture of the launch-bound parts: [zero, scorer fwd, loss, scorer bwd] and [SGD];
    # the ONE all-reduce of the flat bucket stays between the two replays (RCCL, eager).
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            eager_step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g_fb, g_sgd = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    one = torch.ones((), device=dev)
    with torch.cuda.graph(g_fb):
        static_value = fwd_bwd()
        static_scalars = torch.stack([static_value, one])
    with torch.cuda.graph(g_sgd):
        sgd()

    def graph_step():
        g_fb.replay()
        s = bucket.all_reduce(static_scalars, average=True)
        g_sgd.replay()
        return s[0] / max(world, 1)
    return graph_step


def measured_traffic(workload, B, L):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes
    (profiles/r01_traffic.json), when the workload and batch match what was profiled."""
    path = os.path.join(ROOT, 'profiles', 'r01_traffic.json')
    try:
        with open(path) as f:
            t = json.load(f).get(workload)
    except (OSError, ValueError):
        return None
    if not t or t.get('B') != B or t.get('L') != L:
        return None
    return t['traffic_bytes']


def cpu_baseline(workload, L, budget_s=12.0):
    """Times the torch-CPU restatement of the reference op graph (oracle/) on a
    bounded sample of the same workload: fwd + autograd bwd on the host cores.
    The thread count is calibrated (best of a few candidates up to the cores this
    process may run on) so that the baseline is not handicapped by oversubscription."""
    from oracle import tfr_ref as R
    from ranking_amd.synthetic import make_batch
    if not workload.startswith('approx_ndcg'):
        return None
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:   # pragma: no cover
        avail = os.cpu_count() or 1
    loss = R.ApproxNDCGLoss()

    def run(labels, logits):
        lg = logits.clone().requires_grad_(True)
        out = R.keras_loss_call(loss, labels, lg)
        out.backward()
        return lg.grad

    Bc = max(8, min(1024, int(2.0e7 // (L * L))))       # [Bc, L, L] fp32 tensors of <= 80 MB
    labels, logits = make_batch(Bc, L, seed=4)
    best_t, best_threads = None, 1
    for threads in sorted({t for t in (avail, 128, 64, 32, 16, 8) if 1 <= t <= avail}, reverse=True):
        torch.set_num_threads(threads)
        run(labels, logits)                              # warm-up at this thread count
        t0 = time.perf_counter()
        run(labels, logits)
        dt = time.perf_counter() - t0
        if best_t is None or dt < best_t:
            best_t, best_threads = dt, threads
    torch.set_num_threads(best_threads)
    iters = max(3, min(200, int(budget_s / max(best_t, 1e-4))))
    times = []
    for _ in range(iters):
        t0 = time.perf_counter()
        run(labels, logits)
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {'value': Bc / med, 'unit': 'lists/s', 'cores': best_threads, 'kind': 'port',
            'sample': 'torch-CPU restatement of the TF-Ranking op graph (oracle/tfr_ref.py), '
                      'ApproxNDCG fwd+autograd bwd, %d lists x L=%d per iteration, median of %d '
                      'iterations, fp32, %d threads (best of a sweep; %d cores available)'
                      % (Bc, L, iters, best_threads, avail)}


def cpu_fused_c_baseline(workload, B, L):
    """The stricter CPU number: the plain-C restatement of the same fwd+bwd (oracle/approx_ndcg_c.c, float,
    -Ofast -march=native, OpenMP over lists) on the WHOLE batch, all host cores -- what a fused, vectorised CPU loop
    does, next to the op-graph port above (which is how the reference itself executes).  None when gcc / the
    library is unavailable on this host."""
    if not workload.startswith('approx_ndcg'):
        return None
    try:
        from oracle import c_ref
        from ranking_amd.synthetic import make_batch
        labels, logits = make_batch(B, L, seed=4)
        lb, lg = labels.numpy(), logits.numpy()
        n = max(256, B // 16)
        c_ref.approx_ndcg(lg[:n], lb[:n], temperature=0.1, variant='f32_fast')       # warm-up (threads, pages)
        times = []
        for _ in range(5):
            t0 = time.perf_counter()
            c_ref.approx_ndcg(lg, lb, temperature=0.1, variant='f32_fast')
            times.append(time.perf_counter() - t0)
        times.sort()
        return {'value': B / times[len(times) // 2], 'unit': 'lists/s', 'cores': c_ref.threads(), 'kind': 'port',
                'sample': 'plain-C restatement (oracle/approx_ndcg_c.c, fp32, -Ofast -march=native, OpenMP), '
                          'ApproxNDCG fwd+bwd, the full %d x L=%d batch, median of 5' % (B, L)}
    except Exception as e:                                 # the checker's build must never take the bench down
        return {'value': None, 'unit': 'lists/s', 'error': '%s: %s' % (type(e).__name__, e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--workload', default='approx_ndcg', choices=sorted(WORKLOADS))
    ap.add_argument('--batch', type=int, default=0, help='lists per GPU per step (0 = workload default)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-graph', dest='graph', action='store_false', default=True,
                    help='launch eagerly instead of replaying the step from hipGraphs')
    ap.add_argument('--dropout', type=float, default=0.0, help='e2e workloads: Dropout rate of the scorer tower')
    ap.add_argument('--traffic-bytes', type=float, default=None,
                    help='HBM bytes per launch from a separate rocprofv3 --pmc pass (profiles/)')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (torch.cuda.is_available() is False)')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if dist is not None:
        dist.barrier()

    B, L, desc, bytes_per_list = WORKLOADS[args.workload]
    if args.batch > 0:
        B = args.batch
    labels, logits = make_inputs(B, L, seed=4 + rank, device=dev)
    step, kernel_only = build_step(args.workload, labels, logits, args.dropout, args.graph)
    if args.graph and not args.workload.startswith('e2e_'):
        # The loss step is a handful of short launches (order, loss kernel, reduction): replay it from a
        # hipGraph so that the measured rate is the GPU's, not the Python launch path's.
        eager = step
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            static_out = eager()

        def step():
            graph.replay()
            return static_out

    def barrier():
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # dominant-kernel duration: HIP events on the launch stream, kernel launches only.
    kernel_ms = None
    if kernel_only is not None:
        stream = torch.cuda.current_stream()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
               for _ in range(args.steps)]
        torch.cuda.synchronize()
        for a, b in evs:
            a.record(stream)
            kernel_only()
            b.record(stream)
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in evs)
        kernel_ms = sum(ts) / len(ts)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    lists = B * world * args.steps
    value = lists / elapsed
    result = {
        'metric': 'ranked lists/sec (fwd+bwd), ApproxNDCG list_size=200' if args.workload == 'approx_ndcg'
                  else 'ranked lists/sec, %s' % args.workload,
        'value': value, 'unit': 'lists/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': desc, 'lists_per_gpu_per_step': B, 'list_size': L,
                   'valid_length': 'U{ceil(L/2)..L}', 'labels': 'randint{0..4}, -1 padding',
                   'logits': 'N(0,1) tie-free', 'parallelism': 'dp%d (lists sharded, no collective)' % world},
    }
    if kernel_ms is not None:
        algo_bytes = bytes_per_list(L) * B
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        result['roofline'] = {
            'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': achieved / HBM_PEAK_GBS,
            'traffic': args.traffic_bytes if args.traffic_bytes is not None else measured_traffic(args.workload, B, L),
            'kernel': 'approx_ndcg_wave_kernel<4>' if L <= 256 else 'approx_ndcg_kernel', 'kernel_ms': kernel_ms,
            'algorithmic_bytes_per_launch': algo_bytes,
            'note': 'O(L^2) pair work is on-chip: the kernel is VALU/transcendental bound, the HBM '
                    'fraction is reported as the contract asks; see DESIGN.md for the VALU roofline',
            'pairs_per_s': None,
        }
        n_valid_sq = float(((labels >= 0).sum(dim=1).double() ** 2).sum().item())
        result['roofline']['pairs_per_s'] = 2.0 * n_valid_sq / (kernel_ms * 1e-3)   # fwd + bwd evaluations
        # VALU issue floor of the two pair sweeps alone (tools/ubench.hip on MI355X: 15.3 + 20.5 cycles per 64
        # pair evaluations per SIMD; 1024 SIMDs at the 2.4 GHz peak clock) -- the bound that actually applies.
        valu_floor_ms = n_valid_sq / 64.0 * (15.3 + 20.5) / (1024 * 2.4e9) * 1e3
        result['roofline']['valu_floor_ms'] = valu_floor_ms
        result['roofline']['valu_frac'] = valu_floor_ms / kernel_ms
    if args.workload.startswith('e2e_'):
        tflops = e2e_flops_per_list(args.workload, L) * B * world / (elapsed / args.steps) / 1e12
        result['dtype'] = 'bf16'
        result['roofline'] = {
            'bound': 'mfma', 'achieved': tflops / world, 'peak': MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
            'frac': tflops / world / MFMA_PEAK_TFLOPS, 'traffic': measured_traffic(args.workload, B, L),
            'kernel': 'whole training step (tower_gemm256_kernel / tower_wgrad_kernel dominate; profiles/)',
            'note': 'algorithmic scorer flops (fwd+bwd = 6 x MACs) / step time, per GPU; the [M,512] layers '
                    'are HBM-bound above ~55 % MFMA utilisation (DESIGN.md 4.3)'}
    if not args.no_cpu_baseline and world == 1:             # rank 0 at N = 1 only (bounded sample, ~15 s)
        cb = cpu_baseline(args.workload, L)
        if cb is not None:
            result['cpu_baseline'] = cb
            result['gpu_over_cpu'] = value / cb['value']
            fc = cpu_fused_c_baseline(args.workload, B, L)
            if fc is not None:
                cb['fused_c'] = fc                          # second, stricter CPU baseline (same unit)
                if fc.get('value'):
                    result['gpu_over_fused_c_cpu'] = value / fc['value']
    print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
