for s in "" 2,4 2,2 4,2 4,1 8,1 3,3; do echo "STAGGER=$s"; TFR_GEMM_STAGGER=$s MASKS=0 python tools/gemm_ablate.py run _st 2>&1 | tail -1; done
