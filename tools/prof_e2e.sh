#!/bin/bash
# rocprofv3 --kernel-trace --stats of the config-2 end-to-end step (eager launches) + the graph-captured bench line.
# usage (through gpurun): bash tools/prof_e2e.sh ; summary in gpurun_out/e2e_now/stats.txt
export TMPDIR=/tmp
WL=${1:-e2e_softmax}
OUT=gpurun_out/e2e_now; mkdir -p $OUT
python bench.py --workload $WL --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -n 1 | cut -c1-300
rocprofv3 --kernel-trace --stats -d $OUT/prof -o r -- python bench.py --workload $WL --steps 20 --warmup 3 --no-cpu-baseline --no-graph > $OUT/prof.log 2>&1
python tools/rocpd_summary.py stats $OUT/prof/r_results.db > $OUT/stats.txt 2>&1
head -n 32 $OUT/stats.txt
find $OUT -name '*.db' -size +8M -delete
