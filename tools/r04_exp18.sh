#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
{
timeout -k 5 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "stem" 2>&1 | tail -3
timeout -k 5 300 python tools/profile_layers.py | grep -E "stem|32->64 k3s2|total"
} 2>&1 | grep -v amdgpu.ids > $O/exp18.txt
cat $O/exp18.txt
