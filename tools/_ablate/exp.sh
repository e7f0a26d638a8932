for f in 0 4 8 12 16; do echo "FLAGS=$f"; TFR_GEMM_FLAGS=$f MASKS=0 python tools/gemm_ablate.py run 2>&1 | tail -1; done
