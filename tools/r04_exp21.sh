#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
{ timeout -k 5 300 python tools/profile_layers.py --config efficientdet-d1 --batch 8 --reps 10
} 2>&1 | grep -v amdgpu.ids > $O/layers_d1_b8.txt
cat $O/layers_d1_b8.txt
