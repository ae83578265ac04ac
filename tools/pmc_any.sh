#!/bin/bash
# SQ counters of one kernel family under any of the micro-benchmarks (separate --pmc passes, kernel trace only).
#   pmc_any.sh <kernel substring> <script under tools/> [script args]
T=${MYDET_TOOL_TIMEOUT:-200}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; SUB=$1; SCRIPT=$2; shift 2
O=$R/gpurun_out/pmc_any; rm -rf $O; mkdir -p $O
i=0
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" \
           "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" \
           "SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout -k 5 $T rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $O/p$i -- python3 $R/tools/$SCRIPT "$@" > $O/p$i.log 2>&1 || { tail -5 $O/p$i.log; continue; }
  python3 $R/tools/pmc_kernel.py $O/p$i $SUB
done
