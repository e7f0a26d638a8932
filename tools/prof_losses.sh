#!/bin/bash
# rocprofv3 kernel stats of the loss / metric workloads + PMC passes of the headline (the non-quick half of
# tools/gpu_round.sh, without the test run).  usage (through gpurun): bash tools/prof_losses.sh <tag>
set -u
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
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
