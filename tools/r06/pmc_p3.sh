#!/bin/bash
# SQ counters of conv_p3_kernel on two headline layers (separate --pmc passes, kernel trace only)
T=${MYDET_TOOL_TIMEOUT:-200}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_p3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for shape in "32 128 256 2 160" "32 32 64 1 320"; do
  echo "=== conv_p3_kernel, B Cin Cout stride H = $shape"
  i=0
  for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
             "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
             "SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
             "SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INST_CYCLES_VMEM" \
             "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_VALU"; do
    i=$((i+1)); rm -rf $O/p$i
    timeout -k 5 $T rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $O/p$i -- python3 $R/tools/r06/p3_one.py $shape > $O/p$i.log 2>&1 || { tail -5 $O/p$i.log; exit 1; }
    python3 $R/tools/pmc_kernel.py $O/p$i conv_p3_kernel
  done
done
