#!/bin/bash
# Round-2 GPU-box visit.  Usage (through gpurun, from the repo root):  bash tools/gpu_r02.sh <tag> <what...>
#   what: tests bench all prof:<workload> pmc:<workload>
set -u
TAG=${1:-r02}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for what in "$@"; do
  case $what in
    tests)
      timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
      tail -n 15 $OUT/pytest_gpu.log
      timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 1 $OUT/smoke.log ;;
    bench)
      timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
      tail -n 1 $OUT/bench_default.json | cut -c1-1500; tail -n 5 $OUT/bench_default.err ;;
    all)
      for w in pairwise_lambda softmax gumbel_approx_ndcg ndcg_metric approx_ndcg_l1000 e2e_softmax e2e_pairwise_lambda e2e_approx_ndcg_l1000 e2e_groupwise_gumbel; do
        timeout 300 python bench.py --workload $w --steps 50 --warmup 5 > $OUT/bench_$w.json 2> $OUT/bench_$w.err; echo "$w rc=$?"
        tail -n 1 $OUT/bench_$w.json | cut -c1-300; tail -n 3 $OUT/bench_$w.err
      done ;;
    prof:*)
      w=${what#prof:}
      timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_$w -o r -- python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline --also none > $OUT/prof_$w.log 2>&1
      python tools/rocpd_summary.py stats $OUT/prof_$w/r_results.db > $OUT/stats_$w.txt 2>&1
      head -n 40 $OUT/stats_$w.txt ;;
    pmc:*)
      w=${what#pmc:}
      timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch_$w -o r -- python bench.py --workload $w --steps 20 --warmup 2 --no-cpu-baseline --also none > $OUT/pmc_fetch_$w.log 2>&1
      timeout 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write_$w -o r -- python bench.py --workload $w --steps 20 --warmup 2 --no-cpu-baseline --also none > $OUT/pmc_write_$w.log 2>&1
      timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY -d $OUT/pmc_sq_$w -o r -- python bench.py --workload $w --steps 20 --warmup 2 --no-cpu-baseline --also none > $OUT/pmc_sq_$w.log 2>&1
      for p in fetch write sq; do python tools/rocpd_summary.py pmc $OUT/pmc_${p}_$w/r_results.db > $OUT/pmc_${p}_$w.txt 2>&1; head -n 30 $OUT/pmc_${p}_$w.txt; done ;;
  esac
done
find $OUT -name '*.db' -size +8M -delete
