#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
{
timeout -k 5 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "igemm or winograd4" 2>&1 | tail -2
timeout -k 5 300 python tools/profile_layers.py
} 2>&1 | grep -v amdgpu.ids > $O/exp19.txt
cat $O/exp19.txt
