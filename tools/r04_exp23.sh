#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
{ timeout -k 5 300 python tools/profile_layers.py --batch 1 --size 512 --reps 20
  timeout -k 5 300 python bench.py --steps 200 --warmup 20 --batch 1 --size 512 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-230
} 2>&1 | grep -v amdgpu.ids > $O/layers_yolo_b1.txt
cat $O/layers_yolo_b1.txt
