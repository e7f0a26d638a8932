"""bench.py pieces that need no GPU: the workload table, the committed PMC traffic it reports, the flop count of the
end-to-end workloads and the schema of the CPU-baseline legs (the JSON-line contract of the driver)."""
import json
import os
import subprocess
import sys

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_workload_table_and_default():
    assert 'approx_ndcg' in bench.WORKLOADS                          # the headline: BASELINE.json's metric
    B, L, desc, bytes_per_list = bench.WORKLOADS['approx_ndcg']
    assert (B, L) == (16384, 200) and bytes_per_list(200) == 12 * 200 + 12          # 2 412 B per list
    for name, (b, l, d, f) in bench.WORKLOADS.items():
        assert b > 0 and l > 0 and isinstance(d, str) and callable(f), name
        assert name.startswith('e2e_') == (f(l) == 0), name


def test_e2e_flops_are_six_times_the_macs():
    per_doc = 136 * 512 + 512 * 512 * 2 + 512
    assert bench.e2e_flops_per_list('e2e_softmax', 100) == 6.0 * per_doc * 100
    assert abs(6.0 * per_doc / 1e6 - 3.567) < 5e-3                                   # 3.567 MFLOP / doc
    per_group = 272 * 512 + 512 * 512 * 2 + 512 * 2
    assert bench.e2e_flops_per_list('e2e_groupwise_gumbel', 50) == 6.0 * per_group * 50
    assert per_group == 664576


def test_committed_traffic_is_reported_for_the_profiled_shapes_only():
    newest = next(f for f in bench.TRAFFIC_FILES if os.path.exists(os.path.join(ROOT, f)))
    t = json.load(open(os.path.join(ROOT, newest)))                              # the newest committed PMC passes
    got = bench.measured_traffic('approx_ndcg', 16384, 200)
    assert got == t['approx_ndcg']['traffic_bytes'] and got >= t['approx_ndcg']['algorithmic_bytes']
    assert got < 1.25 * t['approx_ndcg']['algorithmic_bytes']                         # no wasted re-reads
    assert bench.measured_traffic('approx_ndcg', 8192, 200) is None
    pw = bench.measured_traffic('pairwise_lambda', 4096, 200)                         # the kernel north_star names
    assert pw and t['pairwise_lambda']['algorithmic_bytes'] <= pw < 1.5 * t['pairwise_lambda']['algorithmic_bytes']
    assert bench.measured_traffic('no_such_workload', 1, 1) is None


def test_fused_c_baseline_schema():
    for w in ('approx_ndcg', 'pairwise_lambda', 'softmax', 'ndcg_metric'):
        fc = bench.cpu_fused_c_baseline(w, 512, 50)
        assert fc['unit'] == 'lists/s' and fc['kind'] == 'port' and fc['value'] > 0 and fc['cores'] >= 1, w
    assert bench.cpu_fused_c_baseline('e2e_softmax', 512, 50) is None


def test_every_workload_has_a_cpu_baseline():
    """the op-graph port is timed for every workload (bounded sample), not only the headline."""
    for w, (B, L, _, _) in bench.WORKLOADS.items():
        L = min(L, 60)                                                              # keep the CPU suite short
        w = bench.HBM_VARIANTS[w][0] if w in bench.HBM_VARIANTS else w              # (same kernels on a cycled working set)
        cb = bench.cpu_baseline(w, L, budget_s=0.05)
        assert cb and cb['value'] > 0 and cb['unit'] == 'lists/s' and cb['kind'] == 'port' and cb['cores'] >= 1, w
        assert 'lists x L=%d' % L in cb['sample'], w


def test_traffic_is_labelled_as_committed_not_measured():
    assert 'NOT measured in this run' in bench.traffic_source('approx_ndcg', 16384, 200)
    assert 'profiles/' in bench.traffic_source('approx_ndcg', 16384, 200)
    assert bench.traffic_source('approx_ndcg', 1, 1).startswith('not profiled')


def test_launch_command_is_the_drivers():
    cmd = bench.launch_command(8, ['--gpus', '8', '--steps', '5'], 29511)
    assert cmd[1:3] == ['-m', 'torch.distributed.run']
    assert '--nnodes=1' in cmd and cmd[cmd.index('--nproc-per-node') + 1] == '8'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and cmd[cmd.index('--master-port') + 1] == '29511'
    assert cmd[-5].endswith('bench.py') and cmd[-4:] == ['--gpus', '8', '--steps', '5']


def test_gpus_n_from_a_bare_shell_spawns_n_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE re-executes itself under torch.distributed.run: two ranks
    rendezvous on 127.0.0.1, run a real all-reduce (gloo here: no GPU) and rank 0 prints ONE line."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--plumbing-check'],
                         capture_output=True, text=True, env=env, timeout=240)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['rccl_ranks'] == 2 and out['all_reduce_sum'] == 3.0
    # without the check flag and without GPUs the same launch refuses loudly BEFORE spawning anything
    import torch
    if not torch.cuda.is_available():
        res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1'],
                             capture_output=True, text=True, env=env, timeout=120)
        assert res.returncode != 0 and 'needs 2 MI355X GPUs' in (res.stderr + res.stdout)


def test_bench_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        return
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '1', '--warmup', '0'],
                         capture_output=True, text=True)
    assert res.returncode != 0 and 'MI355X' in (res.stderr + res.stdout)              # no CPU fallback, loudly


def test_extra_workloads_run_in_child_processes():
    """a GPU fault aborts the whole process, so every `also` workload gets its own: the child command is
    this script with --workload X --also none (N ranks of it under torch.distributed.run at N > 1)."""
    import argparse
    args = argparse.Namespace(no_cpu_baseline=False, graph=True, dropout=None)
    cmd = bench.child_command('pairwise_lambda', args, 1, 20, 5)
    assert cmd[0] == sys.executable and cmd[1].endswith('bench.py')
    assert cmd[cmd.index('--workload') + 1] == 'pairwise_lambda' and cmd[cmd.index('--also') + 1] == 'none'
    assert cmd[cmd.index('--steps') + 1] == '20' and cmd[cmd.index('--warmup') + 1] == '5'
    assert '--no-cpu-baseline' not in cmd and '--no-graph' not in cmd and '--dropout' not in cmd
    args = argparse.Namespace(no_cpu_baseline=True, graph=False, dropout=0.25)
    cmd = bench.child_command('e2e_softmax', args, 4, 10, 3, port=29999)
    assert cmd[1:3] == ['-m', 'torch.distributed.run'] and cmd[cmd.index('--nproc-per-node') + 1] == '4'
    assert cmd[cmd.index('--master-port') + 1] == '29999' and cmd[cmd.index('--gpus') + 1] == '4'
    assert '--no-cpu-baseline' in cmd and '--no-graph' in cmd and cmd[cmd.index('--dropout') + 1] == '0.25'


def test_a_dead_or_silent_child_costs_one_entry(monkeypatch, tmp_path):
    """run_child returns an error dict (never raises) when the child aborts, prints no JSON line or hangs; a healthy
    child's last JSON line is the result."""
    import argparse
    args = argparse.Namespace(no_cpu_baseline=True, graph=True, dropout=None)

    def fake(code):
        script = tmp_path / 'child.py'
        script.write_text(code)
        monkeypatch.setattr(bench, 'child_command', lambda *a, **k: [sys.executable, str(script)])

    fake('import os, sys\nsys.stderr.write("Memory access fault by GPU node-2\\n")\nos.abort()\n')
    r = bench.run_child('pairwise_lambda', args, 1, 10, 3, 30)
    assert 'error' in r and 'rc' in r['error'] and 'Memory access fault' in r['stderr_tail']
    fake('print("no json here")\n')
    assert 'printed no JSON line' in bench.run_child('pairwise_lambda', args, 1, 10, 3, 30)['error']
    fake('import time\ntime.sleep(60)\n')
    assert 'exceeded' in bench.run_child('pairwise_lambda', args, 1, 10, 3, 1)['error']
    fake('print("noise")\nprint(\'{"metric": "m", "value": 2.5}\')\nprint("trailing noise")\n')
    r = bench.run_child('pairwise_lambda', args, 1, 10, 3, 30)
    assert r['value'] == 2.5 and r['wall_s'] > 0


def test_last_json_line_takes_the_last_parseable_one():
    assert bench.last_json_line('{"a": 1}\n{"a": 2}\nnot json\n{broken') == {'a': 2}
    assert bench.last_json_line('nothing') is None


def test_traffic_entries_are_keyed_by_kernel_symbol():
    """Round 6: counters are attached only to the kernel symbol they were measured on -- no fallback to
    older rounds' files, no counters of a replaced kernel under a new kernel's time."""
    assert all(f.startswith('profiles/r06') or f.startswith('profiles/r05') for f in bench.TRAFFIC_FILES)
    assert bench._symbol('tower_gemm256p_kernel<2, 1, 2, 0> (forward hidden layer)') == 'tower_gemm256p_kernel'
    assert bench._symbol('ndcg_lean_kernel (ranks from ...)') == 'ndcg_lean_kernel'
    assert bench.measured_traffic('approx_ndcg', 16384, 200, 'approx_ndcg_wave_kernel') is not None
    assert bench.measured_traffic('approx_ndcg', 16384, 200, 'some_new_kernel') is None
    assert bench.measured_traffic('ndcg_metric_hbm', 16384, 200, 'ndcg_count_wave_kernel') is None
    assert 'kernel' in bench.traffic_source('approx_ndcg', 16384, 200, 'some_new_kernel')


def test_digest_line_carries_the_contract_and_fits_a_short_tail():
    """Round 6 (VERDICT r5 next #5b): the LAST line bench.py prints is the contract line in compact form -- every contract key
    of the main workload + one short record per extra -- below 4 KB for eight workloads, so it survives an 8 KB tail."""
    roof = {'bound': 'hbm', 'achieved': 376.9, 'peak': 8000.0, 'unit': 'GB/s', 'frac': 0.047, 'traffic': 45140992,
            'traffic_source': 'x' * 300, 'kernel': 'approx_ndcg_wave_kernel', 'kernel_ms': 0.1048, 'note': 'y' * 500,
            'valu_frac': 0.65, 'trans_frac': 0.3, 'algorithmic_bytes_per_launch': 39518208}
    cb = {'value': 5429.4, 'unit': 'lists/s', 'cores': 16, 'kind': 'port', 'sample': 's' * 300, 'fused_c': {'value': 1.4e6, 'note': 'n' * 300}}
    main = {'metric': 'ranked lists/sec (fwd+bwd), ApproxNDCG list_size=200', 'value': 1.36e8, 'unit': 'lists/s', 'n_gpus': 1,
            'steps': 20, 'warmup': 5, 'ms_per_step': 0.12, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic', 'config': {'workload': 'w' * 200, 'lists_per_gpu_per_step': 16384, 'list_size': 200,
                                                            'parallelism': 'dp1', 'launch_order': 'o' * 300},
            'roofline': roof, 'cpu_baseline': cb, 'steady_state': {'ms_per_step': 0.12, 'value': 1.37e8, 'note': 'z' * 400},
            'order_cached': {'ms_per_step': 0.11, 'value': 1.5e8, 'note': 'z' * 400}}
    child = dict(main, roofline=dict(roof, step={'frac': 0.15}, kernel='tower_gemm256p_kernel<BN+ReLU prologue> (long text ' + 'k' * 300),
                 dropout_0={'ms_per_step': 3.3, 'value': 1.0}, all_reduce={'ms': 0.08, 'exposed_ms': 0.01})
    main['also'] = {w: child for w in bench.DEFAULT_ALSO}
    main['also']['broken'] = {'error': 'child exited with rc -6 ' + 'e' * 500}
    d = bench.digest_line(main)
    line = json.dumps(d)
    assert len(line) < 4096, len(line)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in d['roofline'], k
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in d['cpu_baseline'], k
    assert set(d['also']) == set(bench.DEFAULT_ALSO) | {'broken'}
    e = d['also']['e2e_softmax']
    assert e['kernel'] == 'tower_gemm256p_kernel<BN+ReLU' or e['kernel'].startswith('tower_gemm256p_kernel')
    assert e['step_frac'] == 0.15 and e['all_reduce_exposed_ms'] == 0.01 and 'error' in d['also']['broken']
