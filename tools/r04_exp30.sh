#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
{ timeout -k 5 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "igemm or winograd" 2>&1 | tail -3
  timeout -k 5 300 python bench.py --steps 200 --warmup 20 --batch 1 --size 512 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-230
  timeout -k 5 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-230
} 2>&1 | grep -v amdgpu.ids > $O/exp30.txt
cat $O/exp30.txt
