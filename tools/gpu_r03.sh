#!/bin/bash
# Round-3 GPU-box visits.  Usage (through gpurun, from the repo root):  bash tools/gpu_r03.sh <tag> <what...>
#   what: pw_tests pw_bench tests bench all gemm32 prof:<workload> traffic:<workload> pmc:<workload>
set -u
TAG=${1:-r03}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
PW='lambdarank or pairwise or lambda_weight'
for what in "$@"; do
  case $what in
    pw_tests)
      # the group kernel forced onto every small-batch edge-case test (8 lists per workgroup, empty slots), then the default dispatch
      TFR_LAMBDARANK_GROUP_MIN_B=1 TFR_LAMBDARANK_WAVES=8 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$PW" > $OUT/pw_forced8.log 2>&1; echo "forced W=8 rc=$?"; tail -n 12 $OUT/pw_forced8.log
      TFR_LAMBDARANK_GROUP_MIN_B=1 TFR_LAMBDARANK_WAVES=3 TFR_LAMBDARANK_REP=16 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$PW" > $OUT/pw_forced3.log 2>&1; echo "forced W=3 R=16 rc=$?"; tail -n 5 $OUT/pw_forced3.log
      timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$PW or headline_batch" > $OUT/pw_default.log 2>&1; echo "default rc=$?"; tail -n 25 $OUT/pw_default.log ;;
    pw_bench)
      for b in 4096 16384; do
        for g in 1 0; do
          TFR_LAMBDARANK_GROUP=$g timeout 300 python bench.py --workload pairwise_lambda --batch $b --steps 100 --warmup 10 --no-cpu-baseline --also none --busy-seconds 0 > $OUT/pw_B${b}_g$g.json 2> $OUT/pw_B${b}_g$g.err
          echo "B=$b group=$g rc=$?"; python - <<PY
import json
try:
    d = json.loads(open('$OUT/pw_B${b}_g$g.json').read().strip().splitlines()[-1])
    r = d['roofline']
    print('  ms_per_step %.4f  kernel_ms %.4f  lists/s %.3e  valu_frac %.3f' % (d['ms_per_step'], r['kernel_ms'], d['value'], r['valu_frac']))
except Exception as e:
    print('  parse error', e)
PY
        done
      done
      for w in 4 8; do for r in 16 32; do
        TFR_LAMBDARANK_WAVES=$w TFR_LAMBDARANK_REP=$r timeout 300 python bench.py --workload pairwise_lambda --batch 4096 --steps 100 --warmup 10 --no-cpu-baseline --also none 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  W=$w R=$r B=4096 kernel_ms %.4f step %.4f' % (d['roofline']['kernel_ms'], d['ms_per_step']))"
      done; done ;;
    pw_quick)
      TFR_LAMBDARANK_GROUP_MIN_B=1 TFR_LAMBDARANK_WAVES=8 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lambdarank" > $OUT/pw_forced8.log 2>&1; echo "forced W=8 rc=$?"; tail -n 4 $OUT/pw_forced8.log
      for b in 4096 16384; do
          timeout 300 python bench.py --workload pairwise_lambda --batch $b --steps 100 --warmup 10 --no-cpu-baseline --also none 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  B=$b kernel_ms %.4f step %.4f valu_frac %.3f' % (d['roofline']['kernel_ms'], d['ms_per_step'], d['roofline']['valu_frac']))"
      done ;;
    pw_prof)
      for b in 4096 16384; do B=$b timeout 200 python tools/phase_profile.py group > $OUT/phase_group_B$b.txt 2>&1; cat $OUT/phase_group_B$b.txt; done
      B=4096 TFR_LAMBDARANK_WAVES=4 TFR_LAMBDARANK_REP=16 timeout 200 python tools/phase_profile.py group > $OUT/phase_group_B4096_W4.txt 2>&1; cat $OUT/phase_group_B4096_W4.txt
      for b in 4096 16384; do
        timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY -d $OUT/pmc_sq_pw$b -o r -- python bench.py --workload pairwise_lambda --batch $b --steps 20 --warmup 2 --no-cpu-baseline --also none --busy-seconds 0 > $OUT/pmc_sq_pw$b.log 2>&1
        python tools/rocpd_summary.py pmc $OUT/pmc_sq_pw$b/r_results.db lambdarank > $OUT/pmc_sq_pw$b.txt 2>&1; cat $OUT/pmc_sq_pw$b.txt
        timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA -d $OUT/pmc_sq2_pw$b -o r -- python bench.py --workload pairwise_lambda --batch $b --steps 20 --warmup 2 --no-cpu-baseline --also none --busy-seconds 0 > $OUT/pmc_sq2_pw$b.log 2>&1
        python tools/rocpd_summary.py pmc $OUT/pmc_sq2_pw$b/r_results.db lambdarank > $OUT/pmc_sq2_pw$b.txt 2>&1; cat $OUT/pmc_sq2_pw$b.txt
      done ;;
    tests)
      timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
      tail -n 70 $OUT/pytest_gpu.log
      timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 1 $OUT/smoke.log ;;
    bench)
      ( time timeout 1200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_default.time; echo "bench rc=$?"
      tail -n 1 $OUT/bench_default.json | cut -c1-1500; tail -n 5 $OUT/bench_default.err ;;
    all)
      for w in pairwise_lambda softmax gumbel_approx_ndcg ndcg_metric approx_ndcg_l1000 e2e_softmax e2e_pairwise_lambda e2e_approx_ndcg_l1000 e2e_groupwise_gumbel; do
        timeout 300 python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench_$w.json 2> $OUT/bench_$w.err; echo "$w rc=$?"
        tail -n 1 $OUT/bench_$w.json | cut -c1-300; tail -n 3 $OUT/bench_$w.err
      done ;;
    prof:*)
      w=${what#prof:}
      timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_$w -o r -- python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline --also none --busy-seconds 0 > $OUT/prof_$w.log 2>&1
      python tools/rocpd_summary.py stats $OUT/prof_$w/r_results.db > $OUT/stats_$w.txt 2>&1
      head -n 40 $OUT/stats_$w.txt ;;
    gemm32)             # fp32 Dense kernels: timings, kernel-trace table, SQ counters (separate passes)
      timeout 100 python tools/time_gemm_f32.py > $OUT/gemm32_time.txt 2>&1; tail -n 16 $OUT/gemm32_time.txt
      GEMM_QUICK=1 timeout 100 rocprofv3 --kernel-trace --stats -d $OUT/gemm32_st -o r -- python tools/time_gemm_f32.py > $OUT/gemm32_st.log 2>&1
      python tools/rocpd_summary.py stats $OUT/gemm32_st/r_results.db > $OUT/gemm32_stats.txt 2>&1; head -n 8 $OUT/gemm32_stats.txt
      GEMM_QUICK=1 timeout 100 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT -d $OUT/gemm32_pmc -o r -- python tools/time_gemm_f32.py > $OUT/gemm32_pmc.log 2>&1
      python tools/rocpd_summary.py pmc $OUT/gemm32_pmc/r_results.db gemm_f32 > $OUT/gemm32_pmc.txt 2>&1; head -n 30 $OUT/gemm32_pmc.txt | cut -c1-150 ;;
    traffic:*)          # FETCH_SIZE / WRITE_SIZE only, few steps (the e2e workloads: thousands of dispatches per second of bench)
      w=${what#traffic:}
      timeout 200 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch_$w -o r -- python bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline --also none --busy-seconds 0 --dropout 0.5 > $OUT/pmc_fetch_$w.log 2>&1
      timeout 200 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write_$w -o r -- python bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline --also none --busy-seconds 0 --dropout 0.5 > $OUT/pmc_write_$w.log 2>&1
      for p in fetch write; do python tools/rocpd_summary.py pmc $OUT/pmc_${p}_$w/r_results.db tower_gemm256p > $OUT/pmc_${p}_$w.txt 2>&1; head -n 8 $OUT/pmc_${p}_$w.txt | cut -c1-200; done ;;
    pmc:*)
      w=${what#pmc:}
      timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch_$w -o r -- python bench.py --workload $w --steps 20 --warmup 2 --no-cpu-baseline --also none --busy-seconds 0 > $OUT/pmc_fetch_$w.log 2>&1
      timeout 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write_$w -o r -- python bench.py --workload $w --steps 20 --warmup 2 --no-cpu-baseline --also none --busy-seconds 0 > $OUT/pmc_write_$w.log 2>&1
      timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY -d $OUT/pmc_sq_$w -o r -- python bench.py --workload $w --steps 20 --warmup 2 --no-cpu-baseline --also none --busy-seconds 0 > $OUT/pmc_sq_$w.log 2>&1
      for p in fetch write sq; do python tools/rocpd_summary.py pmc $OUT/pmc_${p}_$w/r_results.db > $OUT/pmc_${p}_$w.txt 2>&1; head -n 30 $OUT/pmc_${p}_$w.txt; done ;;
  esac
done
find $OUT -name '*.db' -size +8M -delete
