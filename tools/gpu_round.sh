#!/bin/bash
# One GPU-box visit: parity tests, every bench workload, rocprofv3 stats + PMC of the headline.
# Usage (from the repo root, through gpurun):  bash tools/gpu_round.sh <tag> [quick]
set -u
TAG=${1:-r01}
QUICK=${2:-}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -n 3 $OUT/pytest_gpu.log
python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 1 $OUT/smoke.log
python bench.py > $OUT/bench_headline.json 2> $OUT/bench_headline.err; tail -n 1 $OUT/bench_headline.json
for w in pairwise_lambda softmax gumbel_approx_ndcg ndcg_metric approx_ndcg_l1000 e2e_softmax e2e_pairwise_lambda e2e_approx_ndcg_l1000 e2e_groupwise_gumbel; do
  timeout 300 python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench_$w.json 2> $OUT/bench_$w.err
  tail -n 1 $OUT/bench_$w.json | cut -c1-400
done
if [ -z "$QUICK" ]; then
  for w in approx_ndcg pairwise_lambda softmax ndcg_metric; do
    rocprofv3 --kernel-trace --stats -d $OUT/prof_$w -o r -- python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline > $OUT/prof_$w.log 2>&1
    python tools/rocpd_summary.py stats $OUT/prof_$w/r_results.db > $OUT/stats_$w.txt 2>&1
    head -n 6 $OUT/stats_$w.txt
  done
  rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o r -- python bench.py --steps 20 --warmup 2 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o r -- python bench.py --steps 20 --warmup 2 --no-cpu-baseline > $OUT/pmc_write.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY -d $OUT/pmc_sq -o r -- python bench.py --steps 20 --warmup 2 --no-cpu-baseline > $OUT/pmc_sq.log 2>&1
  for p in fetch write sq; do python tools/rocpd_summary.py pmc $OUT/pmc_$p/r_results.db approx > $OUT/pmc_$p.txt 2>&1; cat $OUT/pmc_$p.txt; done
  find $OUT -name '*.db' -size +8M -delete
fi
