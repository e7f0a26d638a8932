python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --workload ndcg_metric --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
rocprofv3 --kernel-trace --stats -d gpurun_out/p4 -o r -- python bench.py --workload ndcg_metric --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/p4.log 2>&1
python tools/rocpd_summary.py stats gpurun_out/p4/r_results.db | head -12
