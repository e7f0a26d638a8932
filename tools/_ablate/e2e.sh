python -m pytest tests/test_gpu_tower.py tests/test_gpu_groupwise.py -x -q -m gpu 2>&1 | tail -4
for p in 0 1; do echo "PERSIST=$p"; TFR_TOWER_PERSIST=$p TFR_GEMM_FLAGS=0 MASKS=0 python tools/gemm_ablate.py run 2>&1 | tail -1; 
for w in e2e_softmax e2e_approx_ndcg_l1000 e2e_groupwise_gumbel; do TFR_TOWER_PERSIST=$p python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline --also none 2>/dev/null | python tools/bench_brief.py; done; done
