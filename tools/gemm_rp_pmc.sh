#!/bin/bash
# PMC passes of one tower GEMM form, round-5 kernel and resident-panel kernel.  usage: bash tools/gemm_rp_pmc.sh <outdir> [form] [M]
export TMPDIR=/tmp
OUT=${1:-gpurun_out/gemm_pmc}; FORM=${2:-plain}; MM=${3:-512000}
mkdir -p $OUT
cat > /tmp/gemm_one.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
from ranking_amd import _tower_ops as t
M = int(os.environ['MM']); N = K = 512; dev = 'cuda'; form = os.environ['FORM']
A = torch.randn((M, K), device=dev).to(torch.bfloat16); W = (torch.randn((N, K), device=dev) * 0.05).to(torch.bfloat16)
Zp = torch.randn((M, N), device=dev).to(torch.bfloat16); bias = torch.randn(N, device=dev)
sc = torch.rand(K, device=dev) + 0.5; sh = torch.randn(K, device=dev) * 0.1
es = torch.rand(N, device=dev) + 0.5; eh = torch.randn(N, device=dev) * 0.1; em = torch.randn(N, device=dev) * 0.1; er = torch.rand(N, device=dev) + 0.5
out = torch.empty((M, N), dtype=torch.bfloat16, device=dev); aout = torch.empty((M, K), dtype=torch.bfloat16, device=dev)
d = t.Dropout.make(0.5, 7)
f = {'plain': lambda: t.gemm(A, W, N, K, out=out),
     'forward': lambda: t.gemm(A, W, N, K, prologue=2, a_scale=sc, a_shift=sh, bias=bias, epilogue=t.EPI_STATS, out=out),
     'forward_drop': lambda: t.gemm(A, W, N, K, prologue=2, a_scale=sc, a_shift=sh, bias=bias, epilogue=t.EPI_STATS, out=out, pro_dropout=d, a_out=aout),
     'dgrad': lambda: t.gemm(A, W, N, K, epilogue=t.EPI_RELU_BWD, Zp=Zp, e_scale=es, e_shift=eh, e_mean=em, e_rstd=er, out=out)}[form]
for _ in range(6):
    f()
torch.cuda.synchronize()
PY
for rp in 0 1; do
  for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE"; do
    tag=$(echo $c | tr ' ' '_' | cut -c1-24)
    MM=$MM FORM=$FORM TFR_GEMM_RP=$rp timeout 300 rocprofv3 --pmc $c -d $OUT/rp${rp}_$tag -o r -- python /tmp/gemm_one.py > $OUT/rp${rp}_$tag.log 2>&1
    echo "== RP=$rp $FORM M=$MM: $c"
    python tools/rocpd_summary.py pmc $OUT/rp${rp}_$tag/r_results.db tower_gemm 2>&1 | cut -c1-150
  done
  MM=$MM FORM=$FORM TFR_GEMM_RP=$rp timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/rp${rp}_trace -o r -- python /tmp/gemm_one.py > $OUT/rp${rp}_trace.log 2>&1
  python tools/rocpd_summary.py stats $OUT/rp${rp}_trace/r_results.db 2>&1 | grep tower_gemm | cut -c1-150
done
find $OUT -name '*.db' -delete
