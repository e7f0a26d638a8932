#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the config-2 end-to-end step, eager launches.
export TMPDIR=/tmp
OUT=gpurun_out/e2e_pmc; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $OUT/$c -o r -- python bench.py --workload e2e_softmax --steps 6 --warmup 2 --no-cpu-baseline --no-graph > $OUT/$c.log 2>&1
  python tools/rocpd_summary.py pmc $OUT/$c/r_results.db tower > $OUT/$c.txt 2>&1
  cat $OUT/$c.txt
done
find $OUT -name '*.db' -size +8M -delete
