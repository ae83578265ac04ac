#!/bin/bash
# SQ counters (MFMA busy, LDS conflicts, waits) of the F(4x4) GEMM launch on the headline's three deep layer shapes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O $R/gpurun_out/pmc_w4
for s in "128 256 80" "256 512 40" "512 1024 20"; do set -- $s
  echo "=== F(4x4) $1->$2 @$3^2, batch 32 (conv_wino4_kernel: whole-item launch and K-piece launch listed separately by grid size)"
  CIN=$1 COUT=$2 HW=$3 bash $R/tools/pmc_wino4.sh 2>&1 | grep -v amdgpu.ids
done > $O/r04_pmc_sq_wino4.txt 2>&1
tail -40 $O/r04_pmc_sq_wino4.txt
