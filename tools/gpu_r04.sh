#!/bin/bash
# Round-4 GPU-box visits.  Usage (through gpurun, from the repo root):  bash tools/gpu_r04.sh <tag> <what...>
#   what: final (the driver's command, twice) | final1 | tests (full -m gpu suite + smoke) | repro / hunt / driver1 (the
#         fault hunt of DESIGN 4.2) | profiles / profiles2 / profiles3 / hbm (what profiles/r04_* were made from) |
#         one:<workload> | prof:<workload> | pmc:<workload> | same-box A/B steps of the switches in DESIGN 8:
#         headline_ab, int_ab, metrics_ab, gemm_ab, aout_ab, aout2_ab, bt_ab, stream_ab, stream_groups | approx_tests,
#         ingest, softmax_final (targeted test subsets)
set -u
TAG=${1:-r04}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
DRV="--gpus 1 --steps 20 --warmup 5"
brief() { python tools/bench_summary.py "$1" 2>/dev/null || tail -c 600 "$1"; }
for what in "$@"; do
  case $what in
    repro)
      # the driver's exact command first, then every workload of it alone (driver's K / W), graph and eager
      ( time timeout 900 python3 bench.py $DRV > $OUT/driver.out 2> $OUT/driver.err ) 2> $OUT/driver.time; echo "driver rc=$?"
      tail -n 4 $OUT/driver.err; tail -n 1 $OUT/driver.out | cut -c1-400
      for w in approx_ndcg pairwise_lambda e2e_approx_ndcg_l1000 e2e_groupwise_gumbel; do
        for mode in graph eager; do
          extra=""; [ $mode = eager ] && extra="--no-graph"
          timeout 300 python3 bench.py --workload $w --also none --no-cpu-baseline --busy-seconds 0 --steps 20 --warmup 5 $extra > $OUT/one_${w}_$mode.out 2> $OUT/one_${w}_$mode.err
          echo "$w $mode rc=$?"; tail -n 2 $OUT/one_${w}_$mode.err | cut -c1-300
        done
      done ;;
    hunt)
      # localise the e2e graph-mode fault: tight allocations (one hipMalloc per tensor), a sync + name after every ABI call
      for w in e2e_groupwise_gumbel e2e_approx_ndcg_l1000; do
        for d in 0.5 0; do
          PYTORCH_NO_HIP_MEMORY_CACHING=1 PYTORCH_NO_CUDA_MEMORY_CACHING=1 TFR_SYNC_EVERY_CALL=1 AMD_SERIALIZE_KERNEL=3 timeout 300 python3 bench.py --workload $w --also none --no-cpu-baseline --busy-seconds 0 --steps 3 --warmup 1 --no-graph --dropout $d > $OUT/hunt_${w}_d$d.out 2> $OUT/hunt_${w}_d$d.err
          echo "hunt eager nocache $w dropout=$d rc=$?"; grep -v "^\[tfr\]" $OUT/hunt_${w}_d$d.err | tail -n 2 | cut -c1-300; grep "^\[tfr\]" $OUT/hunt_${w}_d$d.err | tail -n 3
          timeout 300 python3 bench.py --workload $w --also none --no-cpu-baseline --busy-seconds 0 --steps 20 --warmup 5 --dropout $d > $OUT/graph_${w}_d$d.out 2> $OUT/graph_${w}_d$d.err
          echo "graph $w dropout=$d rc=$?"; tail -n 2 $OUT/graph_${w}_d$d.err | cut -c1-300
        done
      done ;;
    driver1)
      ( time timeout 1200 python3 bench.py $DRV > $OUT/driver.out 2> $OUT/driver.err ) 2> $OUT/driver.time; echo "driver rc=$?"
      tail -n 3 $OUT/driver.err | cut -c1-300; python tools/bench_summary.py $OUT/driver.out; tail -n 3 $OUT/driver.time ;;
    headline_ab)
      H="--workload approx_ndcg --also none --no-cpu-baseline --busy-seconds 0 --steps 200 --warmup 20"
      for v in "" "TFR_LOSS_SUM_FUSED=0" "TFR_APPROX_PAIR_RCP=0 TFR_LOSS_SUM_FUSED=0"; do
        env $v timeout 200 python3 bench.py $H > $OUT/h_$(echo $v | tr ' =' '__').out 2> $OUT/h.err; echo "[$v] rc=$?"; python tools/bench_summary.py $OUT/h_$(echo $v | tr ' =' '__').out | tail -n 1; tail -n 1 $OUT/h.err | cut -c1-200
      done ;;
    approx_tests)
      timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -x -q -m gpu -k "approx or order or headline or smoke or gumbel or keras_loss" > $OUT/t_approx.log 2>&1; echo "approx parity tests rc=$?"; tail -n 12 $OUT/t_approx.log ;;
    metrics_ab)
      timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -x -q -m gpu -k "ndcg or mrr or metric or approx or sort or rank" > $OUT/t_metrics.log 2>&1; echo "metric/approx tests rc=$?"; tail -n 3 $OUT/t_metrics.log
      for v in "TFR_SOFTMAX_NT=0" "TFR_SOFTMAX_NT=1"; do
        for w in softmax_hbm softmax; do
          env $v timeout 300 python3 bench.py --workload $w --also none --no-cpu-baseline --busy-seconds 0 --steps 50 --warmup 5 > $OUT/nt_${w}_$(echo $v | tr ' =' '__').out 2> $OUT/nt.err; echo "[$v] $w rc=$?"; python tools/bench_summary.py $OUT/nt_${w}_$(echo $v | tr ' =' '__').out | tail -n 1
        done
      done
      for w in ndcg_metric ndcg_metric_hbm approx_ndcg approx_ndcg_l1000 gumbel_approx_ndcg; do
        timeout 300 python3 bench.py --workload $w --also none --no-cpu-baseline --busy-seconds 0 --steps 100 --warmup 10 > $OUT/m_$w.out 2> $OUT/m.err; echo "$w rc=$?"; python tools/bench_summary.py $OUT/m_$w.out | tail -n 1
      done ;;
    gemm_ab)
      timeout 900 python -m pytest tests/test_gpu_tower.py tests/test_gpu_full_size.py -x -q -m gpu -k "not every_bench" > $OUT/t_tower.log 2>&1; echo "tower tests rc=$?"; tail -n 3 $OUT/t_tower.log
      for v in "" "TFR_GEMM_FLAGS=4"; do
        for w in e2e_approx_ndcg_l1000 e2e_softmax; do
          env $v timeout 300 python3 bench.py --workload $w --also none --no-cpu-baseline --busy-seconds 0 --steps 50 --warmup 5 > $OUT/g_${w}_$(echo $v | tr ' =' '__').out 2> $OUT/g.err; echo "[$v] $w rc=$?"; python tools/bench_summary.py $OUT/g_${w}_$(echo $v | tr ' =' '__').out | tail -n 1
          python - <<PY
import json
d=json.loads([l for l in open('$OUT/g_${w}_$(echo $v | tr ' =' '__').out') if l.startswith('{')][-1])
print('    dropout_0 ms/step', d.get('dropout_0',{}).get('ms_per_step'))
PY
        done
      done ;;
    aout_ab)
      timeout 900 python -m pytest tests/test_gpu_tower.py tests/test_gpu_full_size.py tests/test_gpu_e2e_parity.py tests/test_gpu_groupwise.py -x -q -m gpu -k "not every_bench" > $OUT/t_tower.log 2>&1; echo "tower tests rc=$?"; tail -n 4 $OUT/t_tower.log
      for v in "TFR_TOWER_AOUT=0" "TFR_TOWER_AOUT=1" "TFR_TOWER_AOUT=2"; do
        for w in e2e_approx_ndcg_l1000 e2e_softmax e2e_groupwise_gumbel; do
          env $v timeout 300 python3 bench.py --workload $w --also none --no-cpu-baseline --busy-seconds 0 --steps 50 --warmup 5 > $OUT/a_${w}_$(echo $v | tr ' =' '__').out 2> $OUT/a.err; echo "[$v] $w rc=$?"; python tools/bench_summary.py $OUT/a_${w}_$(echo $v | tr ' =' '__').out | tail -n 1
          python - <<PY
import json
d=json.loads([l for l in open('$OUT/a_${w}_$(echo $v | tr ' =' '__').out') if l.startswith('{')][-1])
print('    dropout_0 ms/step', d.get('dropout_0',{}).get('ms_per_step'))
PY
        done
      done ;;
    bt_ab)
      timeout 900 python -m pytest tests/test_gpu_tower.py tests/test_gpu_full_size.py tests/test_gpu_groupwise.py -x -q -m gpu -k "not every_bench" > $OUT/t_tower.log 2>&1; echo "tower tests rc=$?"; tail -n 4 $OUT/t_tower.log
      for v in "TFR_GEMM_FLAGS=8" ""; do
        for w in e2e_approx_ndcg_l1000 e2e_softmax e2e_groupwise_gumbel; do
          env $v timeout 300 python3 bench.py --workload $w --also none --no-cpu-baseline --busy-seconds 0 --steps 50 --warmup 5 > $OUT/b_${w}_$(echo $v | tr ' =' '__').out 2> $OUT/b.err; echo "[$v] $w rc=$?"; python tools/bench_summary.py $OUT/b_${w}_$(echo $v | tr ' =' '__').out | tail -n 1
        done
      done ;;
    profiles3)
      # the tower after the written operand / keep-bit table: e2e lines, kernel stats, FETCH / WRITE of config 4; the new test
      timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "reduced_scalar" > $OUT/t_sum.log 2>&1; echo "reduced-scalar test rc=$?"; tail -n 2 $OUT/t_sum.log
      for w in e2e_softmax e2e_pairwise_lambda e2e_approx_ndcg_l1000 e2e_groupwise_gumbel; do
        timeout 300 python3 bench.py --workload $w --also none --no-cpu-baseline --busy-seconds 0 --steps 50 --warmup 5 > $OUT/one_$w.out 2> $OUT/one_$w.err; echo "$w rc=$?"; python tools/bench_summary.py $OUT/one_$w.out | tail -n 1
      done
      for w in e2e_approx_ndcg_l1000 e2e_groupwise_gumbel; do
        timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_$w -o r -- python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline --also none --busy-seconds 0 > $OUT/prof_$w.log 2>&1
        python tools/rocpd_summary.py stats $OUT/prof_$w/r_results.db > $OUT/stats_$w.txt 2>&1; head -n 12 $OUT/stats_$w.txt | cut -c1-130
      done
      w=e2e_approx_ndcg_l1000
      timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch_$w -o r -- python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline --also none --busy-seconds 0 > $OUT/pmc_fetch_$w.log 2>&1
      timeout 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write_$w -o r -- python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline --also none --busy-seconds 0 > $OUT/pmc_write_$w.log 2>&1
      for p in fetch write; do python tools/rocpd_summary.py pmc $OUT/pmc_${p}_$w/r_results.db > $OUT/pmc_${p}_$w.txt 2>&1; grep "tower_gemm256p\|wgrad" $OUT/pmc_${p}_$w.txt | head -n 8 | cut -c1-160; done
      find $OUT -name '*.db' -size +4M -delete ;;
    aout2_ab)
      timeout 900 python -m pytest tests/test_gpu_tower.py tests/test_gpu_full_size.py -x -q -m gpu -k "not every_bench" > $OUT/t_tower.log 2>&1; echo "tower tests rc=$?"; tail -n 3 $OUT/t_tower.log
      for v in "TFR_TOWER_AOUT=0" "TFR_TOWER_AOUT=1" "TFR_TOWER_AOUT=2"; do
        for w in e2e_approx_ndcg_l1000 e2e_softmax; do
          env $v timeout 300 python3 bench.py --workload $w --also none --no-cpu-baseline --busy-seconds 0 --steps 50 --warmup 5 > $OUT/a_${w}_$(echo $v | tr ' =' '__').out 2> $OUT/a.err; echo "[$v] $w rc=$?"; python tools/bench_summary.py $OUT/a_${w}_$(echo $v | tr ' =' '__').out | tail -n 1
          python - <<PY
import json
d=json.loads([l for l in open('$OUT/a_${w}_$(echo $v | tr ' =' '__').out') if l.startswith('{')][-1])
print('    dropout_0 ms/step', d.get('dropout_0',{}).get('ms_per_step'))
PY
        done
      done ;;
    int_ab)
      H="--workload approx_ndcg --also none --no-cpu-baseline --busy-seconds 0 --steps 200 --warmup 20"
      for v in "TFR_APPROX_INT_LABELS=0" ""; do
        env $v timeout 200 python3 bench.py $H > $OUT/i_$(echo $v | tr ' =' '__').out 2> $OUT/i.err; echo "[$v] rc=$?"; python tools/bench_summary.py $OUT/i_$(echo $v | tr ' =' '__').out | tail -n 1
      done ;;
    ingest)
      # bf16 feature ingest: the new tower tests (and the tower suite they sit in), then the host-fed step, fp32 vs bf16 features
      timeout 400 python -m pytest tests/test_gpu_tower.py -x -q -m gpu > $OUT/t_tower.log 2>&1; echo "tower tests rc=$?"; tail -n 6 $OUT/t_tower.log
      timeout 200 python tools/ingest_bench.py > $OUT/ingest.txt 2> $OUT/ingest.err; echo "ingest bench rc=$?"; cat $OUT/ingest.txt; tail -n 2 $OUT/ingest.err | cut -c1-300 ;;
    stream_ab)
      timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -x -q -m gpu -k "softmax" > $OUT/t_softmax.log 2>&1; echo "softmax tests rc=$?"; tail -n 3 $OUT/t_softmax.log
      for v in "TFR_SOFTMAX_STREAM=0" "TFR_SOFTMAX_STREAM_DEPTH=1" "TFR_SOFTMAX_STREAM_DEPTH=2" "TFR_SOFTMAX_STREAM_DEPTH=4" "TFR_SOFTMAX_STREAM_DEPTH=8"; do
        env $v timeout 300 python3 bench.py --workload softmax_hbm --also none --no-cpu-baseline --busy-seconds 0 --steps 50 --warmup 5 > $OUT/st_hbm_$(echo $v | tr ' =' '__').out 2> $OUT/st.err; echo "[$v] softmax_hbm rc=$?"; python tools/bench_summary.py $OUT/st_hbm_$(echo $v | tr ' =' '__').out | tail -n 1
        env $v timeout 300 python3 bench.py --workload softmax --batch 16384 --also none --no-cpu-baseline --busy-seconds 0 --steps 50 --warmup 5 > $OUT/st16k_$(echo $v | tr ' =' '__').out 2> $OUT/st.err; echo "[$v] softmax B=16384 rc=$?"; python tools/bench_summary.py $OUT/st16k_$(echo $v | tr ' =' '__').out | tail -n 1
      done ;;
    stream_groups)
      for v in "TFR_SOFTMAX_STREAM_GROUPS=512" "TFR_SOFTMAX_STREAM_GROUPS=1024" "TFR_SOFTMAX_STREAM_GROUPS=1536" "TFR_SOFTMAX_STREAM_GROUPS=2048" "TFR_SOFTMAX_STREAM_GROUPS=3072" "TFR_SOFTMAX_STREAM_GROUPS=4096"; do
        env $v timeout 300 python3 bench.py --workload softmax_hbm --also none --no-cpu-baseline --busy-seconds 0 --steps 50 --warmup 5 > $OUT/sg_$(echo $v | tr ' =' '__').out 2> $OUT/sg.err; echo "[$v] softmax_hbm rc=$?"; python tools/bench_summary.py $OUT/sg_$(echo $v | tr ' =' '__').out | tail -n 1
      done ;;
    softmax_final)
      timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_e2e_parity.py -x -q -m gpu -k "softmax" > $OUT/t_softmax.log 2>&1; echo "softmax tests rc=$?"; tail -n 3 $OUT/t_softmax.log
      timeout 300 python -m pytest tests/test_gpu_full_size.py -x -q -m gpu -k "every_bench_workload" > $OUT/t_workloads.log 2>&1; echo "bench workloads test rc=$?"; tail -n 2 $OUT/t_workloads.log
      timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $OUT/smoke.log ;;
    bucket_ab)
      TFR_NDCG_BUCKET=1 timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -x -q -m gpu -k "ndcg_mrr_bit_exact or tied_and_squeezed or metric_reference_goldens or ndcg_at_10_bit_exact" > $OUT/t_bucket.log 2>&1; echo "ndcg tests (bucket ranks) rc=$?"; tail -n 4 $OUT/t_bucket.log
      for v in "TFR_NDCG_BUCKET=0" "TFR_NDCG_BUCKET=1"; do
        env $v timeout 100 python3 bench.py --workload ndcg_metric --also none --no-cpu-baseline --busy-seconds 0 --steps 50 --warmup 5 > $OUT/nb_$(echo $v | tr ' =' '__').out 2> $OUT/nb.err; echo "[$v] ndcg_metric rc=$?"; python tools/bench_summary.py $OUT/nb_$(echo $v | tr ' =' '__').out | tail -n 1
      done ;;
    lbucket_quick)
      # bit-identity of the bucket-rank builder against the (validated) counting builder, then the A/B -- about 25 s
      TFR_LAMBDARANK_BUCKET=0 timeout 60 python tools/lbucket_check.py $OUT a > $OUT/lbq.log 2>&1
      TFR_LAMBDARANK_BUCKET=1 timeout 60 python tools/lbucket_check.py $OUT b >> $OUT/lbq.log 2>&1
      timeout 30 python tools/lbucket_check.py $OUT compare >> $OUT/lbq.log 2>&1; echo "lbucket_check rc=$?"; tail -n 8 $OUT/lbq.log | cut -c1-200
      for v in "TFR_LAMBDARANK_BUCKET=0" "TFR_LAMBDARANK_BUCKET=1"; do
        env $v timeout 60 python3 bench.py --workload pairwise_lambda --also none --no-cpu-baseline --busy-seconds 0 --steps 100 --warmup 10 > $OUT/lb_$(echo $v | tr ' =' '__').out 2> $OUT/lb.err; echo "[$v] pairwise_lambda rc=$?"; python tools/bench_summary.py $OUT/lb_$(echo $v | tr ' =' '__').out | tail -n 1
      done ;;
    lbucket_ab)
      # NOT YET RUN (written after the round-4 budget was spent): the LambdaRank builder's ranks from the bucket partition
      TFR_LAMBDARANK_BUCKET=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -x -q -m gpu -k "pairwise or lambda" > $OUT/t_lbucket.log 2>&1; echo "pairwise tests (bucket ranks) rc=$?"; tail -n 4 $OUT/t_lbucket.log
      for v in "TFR_LAMBDARANK_BUCKET=0" "TFR_LAMBDARANK_BUCKET=1"; do
        env $v timeout 100 python3 bench.py --workload pairwise_lambda --also none --no-cpu-baseline --busy-seconds 0 --steps 100 --warmup 10 > $OUT/lb_$(echo $v | tr ' =' '__').out 2> $OUT/lb.err; echo "[$v] pairwise_lambda rc=$?"; python tools/bench_summary.py $OUT/lb_$(echo $v | tr ' =' '__').out | tail -n 1
      done ;;
    final1)
      ( time timeout 1200 python3 bench.py $DRV > $OUT/final_1.out 2> $OUT/final_1.err ) 2> $OUT/final_1.time; echo "final rc=$?"
      tail -n 2 $OUT/final_1.err | cut -c1-300; python tools/bench_summary.py $OUT/final_1.out; tail -n 3 $OUT/final_1.time ;;
    hbm)
      for w in softmax_hbm ndcg_metric_hbm softmax ndcg_metric; do
        timeout 300 python3 bench.py --workload $w --also none --no-cpu-baseline --busy-seconds 0 --steps 50 --warmup 5 > $OUT/hbm_$w.out 2> $OUT/hbm_$w.err; echo "$w rc=$?"; python tools/bench_summary.py $OUT/hbm_$w.out | tail -n 1; tail -n 1 $OUT/hbm_$w.err | cut -c1-200
      done ;;
    profiles)
      # one consolidated visit for profiles/r04_*: bench lines, kernel-trace stats, FETCH / WRITE passes
      for w in approx_ndcg pairwise_lambda softmax ndcg_metric softmax_hbm ndcg_metric_hbm gumbel_approx_ndcg approx_ndcg_l1000 e2e_softmax e2e_pairwise_lambda e2e_approx_ndcg_l1000 e2e_groupwise_gumbel; do
        timeout 300 python3 bench.py --workload $w --also none --no-cpu-baseline --busy-seconds 0 --steps 50 --warmup 5 > $OUT/one_$w.out 2> $OUT/one_$w.err; echo "$w rc=$?"; python tools/bench_summary.py $OUT/one_$w.out | tail -n 1
      done
      for w in approx_ndcg pairwise_lambda softmax_hbm ndcg_metric_hbm e2e_approx_ndcg_l1000 e2e_groupwise_gumbel; do
        timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_$w -o r -- python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline --also none --busy-seconds 0 > $OUT/prof_$w.log 2>&1
        python tools/rocpd_summary.py stats $OUT/prof_$w/r_results.db > $OUT/stats_$w.txt 2>&1; head -n 5 $OUT/stats_$w.txt | cut -c1-130
      done
      for w in approx_ndcg pairwise_lambda softmax_hbm ndcg_metric_hbm e2e_approx_ndcg_l1000; do
        st=20; case $w in *_hbm) st=4;; e2e_*) st=4;; esac
        timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch_$w -o r -- python bench.py --workload $w --steps $st --warmup 2 --no-cpu-baseline --also none --busy-seconds 0 > $OUT/pmc_fetch_$w.log 2>&1
        timeout 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write_$w -o r -- python bench.py --workload $w --steps $st --warmup 2 --no-cpu-baseline --also none --busy-seconds 0 > $OUT/pmc_write_$w.log 2>&1
        for p in fetch write; do python tools/rocpd_summary.py pmc $OUT/pmc_${p}_$w/r_results.db > $OUT/pmc_${p}_$w.txt 2>&1; grep -v "at::\|rocclr" $OUT/pmc_${p}_$w.txt | head -n 6 | cut -c1-160; done
      done
      find $OUT -name '*.db' -size +4M -delete ;;
    profiles2)
      # after the cheaper dz dither: tower tests, the e2e lines + kernel stats again, and the FETCH / WRITE passes of the
      # *_hbm workloads launched eagerly (rocprofv3 --pmc died on their 128-launch graphs)
      timeout 900 python -m pytest tests/test_gpu_tower.py tests/test_gpu_full_size.py -x -q -m gpu -k "not every_bench" > $OUT/t_tower.log 2>&1; echo "tower tests rc=$?"; tail -n 3 $OUT/t_tower.log
      for w in e2e_softmax e2e_pairwise_lambda e2e_approx_ndcg_l1000 e2e_groupwise_gumbel; do
        timeout 300 python3 bench.py --workload $w --also none --no-cpu-baseline --busy-seconds 0 --steps 50 --warmup 5 > $OUT/one_$w.out 2> $OUT/one_$w.err; echo "$w rc=$?"; python tools/bench_summary.py $OUT/one_$w.out | tail -n 1
      done
      for w in e2e_approx_ndcg_l1000 e2e_groupwise_gumbel; do
        timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_$w -o r -- python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline --also none --busy-seconds 0 > $OUT/prof_$w.log 2>&1
        python tools/rocpd_summary.py stats $OUT/prof_$w/r_results.db > $OUT/stats_$w.txt 2>&1; head -n 8 $OUT/stats_$w.txt | cut -c1-130
      done
      for w in softmax_hbm ndcg_metric_hbm; do
        timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch_$w -o r -- python bench.py --workload $w --steps 2 --warmup 1 --no-graph --kernel-timing none --no-cpu-baseline --also none --busy-seconds 0 > $OUT/pmc_fetch_$w.log 2>&1
        timeout 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write_$w -o r -- python bench.py --workload $w --steps 2 --warmup 1 --no-graph --kernel-timing none --no-cpu-baseline --also none --busy-seconds 0 > $OUT/pmc_write_$w.log 2>&1
        for p in fetch write; do python tools/rocpd_summary.py pmc $OUT/pmc_${p}_$w/r_results.db > $OUT/pmc_${p}_$w.txt 2>&1; grep -v "at::\|rocclr" $OUT/pmc_${p}_$w.txt | head -n 4 | cut -c1-160; done
      done
      timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $OUT/pmc_sq_ndcg_metric -o r -- python bench.py --workload ndcg_metric --steps 20 --warmup 2 --no-cpu-baseline --also none --busy-seconds 0 > $OUT/pmc_sq_ndcg_metric.log 2>&1
      python tools/rocpd_summary.py pmc $OUT/pmc_sq_ndcg_metric/r_results.db ndcg > $OUT/pmc_sq_ndcg_metric.txt 2>&1; cat $OUT/pmc_sq_ndcg_metric.txt | cut -c1-150
      find $OUT -name '*.db' -size +4M -delete ;;
    final)
      # the LAST GPU action of the round: the driver command, three times, on the final tree
      for i in 1 2; do
        ( time timeout 1200 python3 bench.py $DRV > $OUT/final_$i.out 2> $OUT/final_$i.err ) 2> $OUT/final_$i.time; echo "final $i rc=$?"
        tail -n 2 $OUT/final_$i.err | cut -c1-300; python tools/bench_summary.py $OUT/final_$i.out; tail -n 3 $OUT/final_$i.time
      done ;;
    tests)
      timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
      tail -n 40 $OUT/pytest_gpu.log
      timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 1 $OUT/smoke.log ;;
    one:*)
      w=${what#one:}
      timeout 400 python3 bench.py --workload $w --also none --no-cpu-baseline --busy-seconds 0 --steps 50 --warmup 5 > $OUT/one_$w.out 2> $OUT/one_$w.err
      echo "$w rc=$?"; tail -n 2 $OUT/one_$w.err | cut -c1-300; tail -n 1 $OUT/one_$w.out | cut -c1-700 ;;
    prof:*)
      w=${what#prof:}
      timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_$w -o r -- python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline --also none --busy-seconds 0 > $OUT/prof_$w.log 2>&1
      python tools/rocpd_summary.py stats $OUT/prof_$w/r_results.db > $OUT/stats_$w.txt 2>&1
      head -n 40 $OUT/stats_$w.txt ;;
    pmc:*)
      w=${what#pmc:}
      timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch_$w -o r -- python bench.py --workload $w --steps 20 --warmup 2 --no-cpu-baseline --also none --busy-seconds 0 > $OUT/pmc_fetch_$w.log 2>&1
      timeout 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write_$w -o r -- python bench.py --workload $w --steps 20 --warmup 2 --no-cpu-baseline --also none --busy-seconds 0 > $OUT/pmc_write_$w.log 2>&1
      for p in fetch write; do python tools/rocpd_summary.py pmc $OUT/pmc_${p}_$w/r_results.db > $OUT/pmc_${p}_$w.txt 2>&1; head -n 12 $OUT/pmc_${p}_$w.txt | cut -c1-200; done ;;
    *) echo "unknown step $what" ;;
  esac
done
