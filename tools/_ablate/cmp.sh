for p in 1; do echo "PERSIST=$p"; TFR_TOWER_PERSIST=$p TFR_GEMM_FLAGS=0 MASKS=0 python tools/gemm_ablate.py run 2>&1 | tail -1; done
TFR_GEMM_FLAGS=0 MASKS=16 python tools/gemm_timeline.py > gpurun_out/gemm_timeline_p.txt 2>&1
