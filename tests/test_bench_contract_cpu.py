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
    t = json.load(open(os.path.join(ROOT, 'profiles', 'r01_traffic.json')))
    got = bench.measured_traffic('approx_ndcg', 16384, 200)
    assert got == t['approx_ndcg']['traffic_bytes'] and got >= t['approx_ndcg']['algorithmic_bytes']
    assert got < 1.25 * t['approx_ndcg']['algorithmic_bytes']                         # no wasted re-reads
    assert bench.measured_traffic('approx_ndcg', 8192, 200) is None
    assert bench.measured_traffic('no_such_workload', 1, 1) is None


def test_fused_c_baseline_schema():
    fc = bench.cpu_fused_c_baseline('approx_ndcg', 512, 50)
    assert fc['unit'] == 'lists/s' and fc['kind'] == 'port' and fc['value'] > 0 and fc['cores'] >= 1
    assert bench.cpu_fused_c_baseline('softmax', 512, 50) is None


def test_bench_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        return
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '1', '--warmup', '0'],
                         capture_output=True, text=True)
    assert res.returncode != 0 and 'MI355X' in (res.stderr + res.stdout)              # no CPU fallback, loudly
