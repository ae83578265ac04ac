#!/bin/bash
# Launch-chain trace of one replayed EfficientDet-D1 lane (batch 8, one lane) at HEAD.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for spec in "d1_b8_l1 efficientdet-d1 8 1" "fcos_b16_l1 d1_fcs2_atss 16 1"; do
  set -- $spec
  D=$O/trace_$1; rm -rf $D; mkdir -p $D
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $R/bench.py --profile --config $2 --batch $3 --lanes $4 --steps 5 --warmup 3 > $D/log.txt 2>&1 || { tail -5 $D/log.txt; }
  python3 $R/tools/chain_trace.py $D $4 > $O/r06_chain_$1.txt 2>&1
  tail -1 $O/r06_chain_$1.txt
  rm -rf $D
done
