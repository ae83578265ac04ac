#!/bin/bash
# SQ counters of the F(4x4) Winograd kernel on one layer (separate --pmc passes, no other trace domains).
# every GPU command runs under `timeout -k 5`: an abort or a stuck process cannot hold the GPU lease for minutes
T=${MYDET_TOOL_TIMEOUT:-300}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="--cin ${CIN:-128} --cout ${COUT:-256} --hw ${HW:-80} --res --wino4 --reps 3"
i=0
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
           "SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INST_CYCLES_VMEM" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmc_w4/p$i; mkdir -p $R/gpurun_out/pmc_w4
  timeout -k 5 $T rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $R/gpurun_out/pmc_w4/p$i -- python3 $R/tools/bench_conv.py $ARGS > $R/gpurun_out/pmc_w4/p$i.log 2>&1 || { tail -5 $R/gpurun_out/pmc_w4/p$i.log; exit 1; }
  python3 $R/tools/pmc_kernel.py $R/gpurun_out/pmc_w4/p$i conv_wino4_kernel
done
