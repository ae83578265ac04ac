#!/bin/bash
# counters of the F(4x4) GEMM launch on one layer: SQ (waits, LDS), TCP / TCC (hit rates, stalls)
T=200
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04/pmc_w4; rm -rf $O; mkdir -p $O
timeout -k 5 60 rocprofv3 -L > $O/counters.txt 2>&1
ARGS="--cin ${CIN:-256} --cout ${COUT:-512} --hw ${HW:-40} --res --wino4 --reps 3"
i=0
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS" \
           "SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum" \
           "TA_BUSY_avr TA_TA_BUSY_sum TA_BUFFER_LOAD_WAVEFRONTS_sum" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -k 5 $T rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $O/p$i -- python3 $R/tools/bench_conv.py $ARGS > $O/p$i.log 2>&1 || { echo "pass $i failed: $pmc"; tail -3 $O/p$i.log; continue; }
  python3 $R/tools/pmc_kernel.py $O/p$i conv_wino4_kernel
done
