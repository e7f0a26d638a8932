python -m pytest tests/test_gpu_tower.py -x -q 2>&1 | tail -2
python bench.py --workload e2e_softmax --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
rocprofv3 --kernel-trace --stats -d gpurun_out/p3 -o r -- python bench.py --workload e2e_softmax --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/p3.log 2>&1
python tools/rocpd_summary.py stats gpurun_out/p3/r_results.db | head -8
