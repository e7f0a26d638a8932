#!/bin/bash
# Round-6 GPU-box visits.  Usage (through gpurun, from the repo root):  bash tools/gpu_r06.sh <tag> <what...>
#   what: tests (full -m gpu suite + smoke) | sums_ab (TFR_LOSS_SUM_FUSED 0 / 1 on the softmax and LambdaRank steps) |
#         ndcg_ab (TFR_NDCG_LEAN 0 / 1 and the persistent-grid size) | gemm_ab (tower GEMM variants: tools/tower_bench.py +
#         the e2e steps) | lrank_ab (LambdaRank switches) | late (tests of the late round-5 changes) | one:<workload> | prof:<workload> | pmc:<workload> |
#         profiles (everything profiles/r06_* is made from: bench lines, rocprofv3 kernel-trace stats, FETCH / WRITE / SQ
#         passes of every dominant kernel, on the tree as it is) | final (the driver's command) | multi (N = 2 if two
#         devices are visible: bench.py --gpus 2 and the RCCL test)
set -u
TAG=${1:-r06}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
DRV="--gpus 1 --steps 20 --warmup 5"
ONE="--also none --no-cpu-baseline --busy-seconds 0"
brief() { python tools/bench_summary.py "$1" 2>/dev/null | tail -n 1 || tail -c 400 "$1"; }
ab() {   # ab <label> <workload> <steps> <env assignments...>: one bench line under an environment
  local label=$1 w=$2 st=$3; shift 3
  local f=$OUT/ab_${w}_$(echo "$label" | tr ' =/' '___').out
  env "$@" timeout 300 python3 bench.py --workload $w $ONE --steps $st --warmup 10 > $f 2> $OUT/ab.err
  echo "[$label] $w rc=$?"; brief $f; tail -n 1 $OUT/ab.err | cut -c1-200
}
for what in "$@"; do
  case $what in
    tests)
      timeout 2400 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
      tail -n 30 $OUT/pytest_gpu.log | cut -c1-200
      timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 1 $OUT/smoke.log ;;
    softmax_quick)
      timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -x -q -m gpu -k "softmax or reduced_scalar" > $OUT/t_softmax.log 2>&1; echo "softmax tests rc=$?"; tail -n 2 $OUT/t_softmax.log | cut -c1-200
      ab "pack" softmax_hbm 50 TFR_DUMMY=0
      ab "pack groups=1024" softmax_hbm 50 TFR_SOFTMAX_STREAM_GROUPS=1024
      ab "pack groups=1536" softmax_hbm 50 TFR_SOFTMAX_STREAM_GROUPS=1536 ;;
    approx_sum_ab)
      # (visit r06k ran this with the round-4 meaning of the switch: 1 = in-launch sum, 0 = reduction launch; now 2 / 1)
      ab "in-launch sum" approx_ndcg 200 TFR_LOSS_SUM_FUSED=2
      ab "reduction launch" approx_ndcg 200 TFR_LOSS_SUM_FUSED=1
      ab "in-launch sum again" approx_ndcg 200 TFR_LOSS_SUM_FUSED=2
      ab "reduction launch again" approx_ndcg 200 TFR_LOSS_SUM_FUSED=1 ;;
    approx_quick)
      timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -x -q -m gpu -k "approx or order or headline or smoke or gumbel or keras_loss or launch_order" > $OUT/t_approx.log 2>&1; echo "approx tests rc=$?"; tail -n 3 $OUT/t_approx.log | cut -c1-200
      grep "headline ApproxNDCG\|config 4 ApproxNDCG\|config 5 Gumbel" $OUT/t_approx.log | cut -c1-160
      ab "pair rcp fwd+bwd" approx_ndcg 200 TFR_APPROX_PAIR_RCP=2
      ab "pair rcp fwd only" approx_ndcg 200 TFR_APPROX_PAIR_RCP=1
      ab "pair rcp fwd+bwd again" approx_ndcg 200 TFR_APPROX_PAIR_RCP=2
      ab "pair rcp fwd only again" approx_ndcg 200 TFR_APPROX_PAIR_RCP=1
      ab "gumbel" gumbel_approx_ndcg 200 TFR_DUMMY=0 ;;
    late)
      # the late round-5 changes: 16-bit Dropout fields, list sizes beyond the LDS range (workspace forms), flatten at 8192
      timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tower.py -x -q -m gpu -k "5000 or 8192 or workgroup_form or grid_stride or 2500 or flatten or dropout or writes_its_transformed" > $OUT/t_late.log 2>&1; echo "late tests rc=$?"; tail -n 12 $OUT/t_late.log | cut -c1-220 ;;
    tests_changed)
      timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -x -q -m gpu -k "reduced_scalar or ndcg or softmax or pairwise or lambda or list_mle or unique or pointwise or keras or metric or sigmoid" > $OUT/t_changed.log 2>&1; echo "changed-area tests rc=$?"; tail -n 12 $OUT/t_changed.log | cut -c1-200 ;;
    sums_ab)
      for v in 0 1; do
        ab "fused=$v" softmax_hbm 50 TFR_LOSS_SUM_FUSED=$v
        ab "fused=$v" softmax 200 TFR_LOSS_SUM_FUSED=$v
        ab "fused=$v" pairwise_lambda 200 TFR_LOSS_SUM_FUSED=$v
      done ;;
    softmax_ab)
      ab "pack=0" softmax_hbm 50 TFR_SOFTMAX_PACK=0
      ab "pack lg=32" softmax_hbm 50 TFR_SOFTMAX_PACK_LG=32
      ab "pack lg=16" softmax_hbm 50 TFR_SOFTMAX_PACK_LG=16
      ab "pack lg=16 groups=1024" softmax_hbm 50 TFR_SOFTMAX_PACK_LG=16 TFR_SOFTMAX_STREAM_GROUPS=1024
      ab "pack lg=16 groups=4096" softmax_hbm 50 TFR_SOFTMAX_PACK_LG=16 TFR_SOFTMAX_STREAM_GROUPS=4096
      TFR_SOFTMAX_PACK_LG=16 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -x -q -m gpu -k "softmax or reduced_scalar" > $OUT/t_softmax16.log 2>&1; echo "softmax tests (lg=16) rc=$?"; tail -n 2 $OUT/t_softmax16.log | cut -c1-200 ;;
    ndcg_ab)
      for v in 0 1; do
        ab "lean=$v" ndcg_metric_hbm 20 TFR_NDCG_LEAN=$v
        ab "lean=$v" ndcg_metric 200 TFR_NDCG_LEAN=$v
      done
      for n in 2048 4096 8192 16384; do ab "lean waves=$n" ndcg_metric_hbm 20 TFR_NDCG_LEAN_WAVES=$n; done ;;
    gemm_ab)
      # the two k-loop forms of the persistent tower GEMM: bit-identity + timing (each form in its own process), then the
      # tower parity suites under the default form, then the end-to-end steps under both
      i=0
      for v in ${GEMM_PP_MODES:-0 1 2}; do
        tag=$(echo abcdef | cut -c$((i+1))); i=$((i+1))
        TFR_GEMM_PP=$v timeout 200 python tools/gemm_pp_check.py $OUT $tag > $OUT/gemm_pp_$tag.txt 2>&1; echo "gemm_pp_check PP=$v rc=$?"; tail -n 11 $OUT/gemm_pp_$tag.txt | cut -c1-160
      done
      timeout 60 python tools/gemm_pp_check.py $OUT compare > $OUT/gemm_pp_compare.txt 2>&1; echo "compare rc=$?"; cat $OUT/gemm_pp_compare.txt | cut -c1-160
      timeout 900 python -m pytest tests/test_gpu_tower.py tests/test_gpu_full_size.py tests/test_gpu_groupwise.py tests/test_gpu_e2e_parity.py -x -q -m gpu -k "not every_bench" > $OUT/t_tower.log 2>&1; echo "tower tests rc=$?"; tail -n 4 $OUT/t_tower.log | cut -c1-200
      for v in ${GEMM_VARIANTS:-"TFR_GEMM_PP=0" "TFR_GEMM_PP=2"}; do
        for w in e2e_approx_ndcg_l1000 e2e_softmax; do ab "$v" $w 50 $v; done
      done ;;
    gemm_quick)
      i=0
      for v in ${GEMM_PP_MODES:-0 2}; do
        tag=$(echo abcdef | cut -c$((i+1))); i=$((i+1))
        TFR_GEMM_PP=$v timeout 200 python tools/gemm_pp_check.py $OUT $tag > $OUT/gemm_pp_$tag.txt 2>&1; echo "gemm_pp_check PP=$v rc=$?"; tail -n 11 $OUT/gemm_pp_$tag.txt | cut -c1-160
      done
      timeout 60 python tools/gemm_pp_check.py $OUT compare > $OUT/gemm_pp_compare.txt 2>&1; echo "compare rc=$?"; cat $OUT/gemm_pp_compare.txt | cut -c1-160 ;;
    lrank_ab)
      ab "default" pairwise_lambda 200 TFR_DUMMY=0
      ab "G=4" pairwise_lambda 200 TFR_LAMBDARANK_WAVES=4
      ab "G=4 helpers=4" pairwise_lambda 200 TFR_LAMBDARANK_WAVES=4 TFR_LAMBDARANK_HELPERS=4
      ab "G=4 helpers=0" pairwise_lambda 200 TFR_LAMBDARANK_WAVES=4 TFR_LAMBDARANK_HELPERS=0
      ab "G=4 R=16 helpers=2" pairwise_lambda 200 TFR_LAMBDARANK_WAVES=4 TFR_LAMBDARANK_REP=16 TFR_LAMBDARANK_HELPERS=2
      ab "G=6 helpers=3" pairwise_lambda 200 TFR_LAMBDARANK_WAVES=6 TFR_LAMBDARANK_HELPERS=3
      ab "G=8 helpers=2" pairwise_lambda 200 TFR_LAMBDARANK_HELPERS=2
      ab "G=8 helpers=6" pairwise_lambda 200 TFR_LAMBDARANK_HELPERS=6
      ab "G=8 R=16" pairwise_lambda 200 TFR_LAMBDARANK_REP=16 ;;
    lrank16)
      ab "default" pairwise_lambda 200 TFR_DUMMY=0
      ab "G=16 helpers=0" pairwise_lambda 200 TFR_LAMBDARANK_WAVES=16 TFR_LAMBDARANK_HELPERS=0
      ab "G=12 helpers=4" pairwise_lambda 200 TFR_LAMBDARANK_WAVES=12 TFR_LAMBDARANK_HELPERS=4
      ab "G=16 helpers=0 R=16" pairwise_lambda 200 TFR_LAMBDARANK_WAVES=16 TFR_LAMBDARANK_HELPERS=0 TFR_LAMBDARANK_REP=16
      ab "default again" pairwise_lambda 200 TFR_DUMMY=0
      TFR_LAMBDARANK_WAVES=16 TFR_LAMBDARANK_HELPERS=0 timeout 200 python tools/phase_profile.py group > $OUT/lrank_phase_g16.txt 2>&1; head -n 12 $OUT/lrank_phase_g16.txt; tail -n 16 $OUT/lrank_phase_g16.txt | head -n 7 ;;
    lrank_counts)
      # per-phase instruction counts of the LambdaRank group kernel: the stamped build truncated after each build phase /
      # with one sweep only (tools/phase_profile.py, STOP=n), SQ counters of every variant
      for st in 1 2 3 4 5 6 7 0; do
        STOP=$st timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM -d $OUT/cnt_$st -o r -- python tools/phase_profile.py group > $OUT/cnt_$st.log 2>&1
        python tools/rocpd_summary.py pmc $OUT/cnt_$st/r_results.db > $OUT/cnt_$st.txt 2>&1
        echo "STOP=$st"; grep "lambdarank_group" $OUT/cnt_$st.txt | cut -c60-130
      done
      find $OUT -name '*.db' -size +4M -delete ;;
    order_quick)
      timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "order or ticket or reduced_scalar or sum or slot" > $OUT/t_order.log 2>&1; echo "order tests rc=$?"; tail -n 4 $OUT/t_order.log | cut -c1-200
      timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -x -q -m gpu -k "pairwise or lambda" > $OUT/t_pw.log 2>&1; echo "pairwise tests rc=$?"; tail -n 3 $OUT/t_pw.log | cut -c1-200
      ab "now" approx_ndcg 200 TFR_DUMMY=0
      ab "now" pairwise_lambda 200 TFR_DUMMY=0
      ab "now again" approx_ndcg 200 TFR_DUMMY=0
      ab "now again" pairwise_lambda 200 TFR_DUMMY=0
      for w in approx_ndcg pairwise_lambda; do
        timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_$w -o r -- python bench.py --workload $w --steps 50 --warmup 5 $ONE > $OUT/prof_$w.log 2>&1
        python tools/rocpd_summary.py stats $OUT/prof_$w/r_results.db > $OUT/stats_$w.txt 2>&1; head -n 6 $OUT/stats_$w.txt | cut -c1-130
      done
      find $OUT -name '*.db' -size +4M -delete ;;
    lgraded)
      timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "graded_builder" > $OUT/t_graded.log 2>&1; echo "graded test rc=$?"; tail -n 6 $OUT/t_graded.log | cut -c1-300
      timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -x -q -m gpu -k "pairwise or lambda" > $OUT/t_pw.log 2>&1; echo "pairwise tests rc=$?"; tail -n 3 $OUT/t_pw.log | cut -c1-200
      for v in 1 0 1 0; do ab "graded=$v" pairwise_lambda 200 TFR_LAMBDARANK_GRADED=$v; done
      ab "graded=1" e2e_pairwise_lambda 20 TFR_LAMBDARANK_GRADED=1 ;;
    order_lpb)
      timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "order" > $OUT/t_order.log 2>&1; echo "order tests rc=$?"; tail -n 2 $OUT/t_order.log | cut -c1-200
      for v in 256 128 256 128; do ab "lpb=$v" approx_ndcg 200 TFR_ORDER_LPB=$v; done
      ab "lpb=128 deep" approx_ndcg 200 TFR_ORDER_LPB=128 TFR_ORDER_DEEP=1
      ab "lpb=128 at 4096" pairwise_lambda 200 TFR_ORDER_LPB=128
      ab "lpb=128 deep at 4096" pairwise_lambda 200 TFR_ORDER_LPB=128 TFR_ORDER_DEEP=1
      ab "default at 4096" pairwise_lambda 200 TFR_DUMMY=0
      for v in 256 128; do
        TFR_ORDER_LPB=$v timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_lpb$v -o r -- python bench.py --workload approx_ndcg --steps 50 --warmup 5 $ONE > $OUT/prof_lpb$v.log 2>&1
        python tools/rocpd_summary.py stats $OUT/prof_lpb$v/r_results.db > $OUT/stats_lpb$v.txt 2>&1; head -n 5 $OUT/stats_lpb$v.txt | cut -c1-130
      done
      find $OUT -name '*.db' -size +4M -delete ;;
    tower_quick)
      timeout 1500 python -m pytest tests/test_gpu_tower.py tests/test_gpu_groupwise.py tests/test_gpu_distributed.py tests/test_gpu_baseline_configs.py -x -q -m gpu > $OUT/t_tower.log 2>&1; echo "tower tests rc=$?"; tail -n 3 $OUT/t_tower.log | cut -c1-200
      for w in e2e_groupwise_gumbel e2e_softmax e2e_approx_ndcg_l1000; do ab "now" $w 50 TFR_DUMMY=0; done
      ab "now again" e2e_groupwise_gumbel 50 TFR_DUMMY=0 ;;
    softmax_sweep)
      for g in 768 1024 1280 1536; do ab "groups=$g nt=1" softmax_hbm 50 TFR_SOFTMAX_PACK_GROUPS=$g TFR_SOFTMAX_NT=1; done
      ab "groups=1024 nt=0" softmax_hbm 50 TFR_SOFTMAX_PACK_GROUPS=1024 TFR_SOFTMAX_NT=0
      ab "groups=1024 nt=1 again" softmax_hbm 50 TFR_SOFTMAX_PACK_GROUPS=1024 TFR_SOFTMAX_NT=1
      ab "groups=1024 nt=1 e2e" e2e_softmax 20 TFR_SOFTMAX_PACK_GROUPS=1024 TFR_SOFTMAX_NT=1
      ab "default e2e" e2e_softmax 20 TFR_DUMMY=0
      ab "groups=1024 nt=1 small" softmax 200 TFR_SOFTMAX_PACK_GROUPS=1024 TFR_SOFTMAX_NT=1
      ab "default small" softmax 200 TFR_DUMMY=0
      ab "default" softmax_hbm 50 TFR_DUMMY=0 ;;
    order_il)
      for v in 128 256 128 256; do ab "il_lpb=$v" approx_ndcg 200 TFR_ORDER_IL_LPB=$v; done
      for v in 128 256 128 256; do ab "il_lpb=$v" pairwise_lambda 200 TFR_ORDER_IL_LPB=$v; done ;;
    one:*)
      w=${what#one:}
      timeout 400 python3 bench.py --workload $w $ONE --steps 50 --warmup 5 > $OUT/one_$w.out 2> $OUT/one_$w.err
      echo "$w rc=$?"; tail -n 2 $OUT/one_$w.err | cut -c1-300; brief $OUT/one_$w.out ;;
    prof:*)
      w=${what#prof:}
      timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_$w -o r -- python bench.py --workload $w --steps 50 --warmup 5 $ONE > $OUT/prof_$w.log 2>&1
      python tools/rocpd_summary.py stats $OUT/prof_$w/r_results.db > $OUT/stats_$w.txt 2>&1
      head -n 14 $OUT/stats_$w.txt | cut -c1-130; find $OUT -name '*.db' -size +4M -delete ;;
    pmc:*)
      w=${what#pmc:}
      eager=""; st=20; case $w in *_hbm) eager="--no-graph --kernel-timing none"; st=2;; e2e_*) st=4;; esac
      for c in FETCH_SIZE WRITE_SIZE; do
        p=$(echo $c | cut -d_ -f1 | tr A-Z a-z)
        timeout 300 rocprofv3 --pmc $c -d $OUT/pmc_${p}_$w -o r -- python bench.py --workload $w --steps $st --warmup 2 $eager $ONE > $OUT/pmc_${p}_$w.log 2>&1
        python tools/rocpd_summary.py pmc $OUT/pmc_${p}_$w/r_results.db > $OUT/pmc_${p}_$w.txt 2>&1; grep -v "at::\|rocclr" $OUT/pmc_${p}_$w.txt | head -n 6 | cut -c1-160
      done
      timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $OUT/pmc_sq_$w -o r -- python bench.py --workload $w --steps $st --warmup 2 $eager $ONE > $OUT/pmc_sq_$w.log 2>&1
      python tools/rocpd_summary.py pmc $OUT/pmc_sq_$w/r_results.db > $OUT/pmc_sq_$w.txt 2>&1; grep -v "at::\|rocclr" $OUT/pmc_sq_$w.txt | head -n 12 | cut -c1-160
      find $OUT -name '*.db' -size +4M -delete ;;
    profiles)
      # ONE consolidated visit on the final tree: every bench line, kernel-trace stats of every dominant kernel, FETCH / WRITE /
      # SQ passes -- what profiles/r06_all_workloads.txt, r06_pmc.txt and r06_traffic.json are assembled from
      for w in approx_ndcg pairwise_lambda softmax ndcg_metric softmax_hbm ndcg_metric_hbm gumbel_approx_ndcg approx_ndcg_l1000 e2e_softmax e2e_pairwise_lambda e2e_approx_ndcg_l1000 e2e_groupwise_gumbel; do
        timeout 300 python3 bench.py --workload $w $ONE --steps 50 --warmup 5 > $OUT/one_$w.out 2> $OUT/one_$w.err; echo "$w rc=$?"; brief $OUT/one_$w.out
      done
      for w in approx_ndcg pairwise_lambda softmax_hbm ndcg_metric_hbm e2e_softmax e2e_approx_ndcg_l1000 e2e_groupwise_gumbel; do
        timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_$w -o r -- python bench.py --workload $w --steps 50 --warmup 5 $ONE > $OUT/prof_$w.log 2>&1
        python tools/rocpd_summary.py stats $OUT/prof_$w/r_results.db > $OUT/stats_$w.txt 2>&1; head -n 5 $OUT/stats_$w.txt | cut -c1-130
      done
      for w in approx_ndcg pairwise_lambda softmax_hbm ndcg_metric_hbm e2e_approx_ndcg_l1000; do
        eager=""; st=20; case $w in *_hbm) eager="--no-graph --kernel-timing none"; st=2;; e2e_*) st=4;; esac
        for c in FETCH_SIZE WRITE_SIZE; do
          p=$(echo $c | cut -d_ -f1 | tr A-Z a-z)
          timeout 300 rocprofv3 --pmc $c -d $OUT/pmc_${p}_$w -o r -- python bench.py --workload $w --steps $st --warmup 2 $eager $ONE > $OUT/pmc_${p}_$w.log 2>&1
          python tools/rocpd_summary.py pmc $OUT/pmc_${p}_$w/r_results.db > $OUT/pmc_${p}_$w.txt 2>&1; grep -v "at::\|rocclr" $OUT/pmc_${p}_$w.txt | head -n 4 | cut -c1-160
        done
      done
      for w in approx_ndcg softmax_hbm ndcg_metric_hbm pairwise_lambda; do
        eager=""; st=20; case $w in *_hbm) eager="--no-graph --kernel-timing none"; st=2;; esac
        timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $OUT/pmc_sq_$w -o r -- python bench.py --workload $w --steps $st --warmup 2 $eager $ONE > $OUT/pmc_sq_$w.log 2>&1
        python tools/rocpd_summary.py pmc $OUT/pmc_sq_$w/r_results.db > $OUT/pmc_sq_$w.txt 2>&1; grep -v "at::\|rocclr" $OUT/pmc_sq_$w.txt | head -n 10 | cut -c1-160
      done
      find $OUT -name '*.db' -size +4M -delete ;;
    final)
      ( time timeout 1200 python3 bench.py $DRV > $OUT/final_1.out 2> $OUT/final_1.err ) 2> $OUT/final_1.time; echo "final rc=$?"
      tail -n 2 $OUT/final_1.err | cut -c1-300; python tools/bench_summary.py $OUT/final_1.out; tail -n 3 $OUT/final_1.time ;;
    multi)
      n=$(python -c "import torch; print(torch.cuda.device_count())")
      echo "devices visible: $n"
      if [ "$n" -ge 2 ]; then
        timeout 600 python -m pytest tests/test_gpu_distributed.py -x -q -m gpu > $OUT/t_dist.log 2>&1; echo "RCCL test rc=$?"; tail -n 3 $OUT/t_dist.log
        timeout 900 python3 bench.py --gpus 2 --steps 20 --warmup 5 > $OUT/multi2.out 2> $OUT/multi2.err; echo "bench --gpus 2 rc=$?"; python tools/bench_summary.py $OUT/multi2.out; tail -n 2 $OUT/multi2.err | cut -c1-300
      fi ;;
    *) echo "unknown step $what" ;;
  esac
done
