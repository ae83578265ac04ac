#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
{ timeout -k 5 300 python tools/solo_vs_batch.py --config efficientdet-d1 --batch 16 --image 15
  timeout -k 5 300 python tools/solo_vs_batch.py --config d1_fcs2_atss --batch 32 --image 31
} 2>&1 | grep -v amdgpu.ids > $O/solo_vs_batch.txt
cat $O/solo_vs_batch.txt
