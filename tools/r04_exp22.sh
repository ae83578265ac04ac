#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
{
for m in 64 32 64 32; do
  echo "== MYDET_WINO4_MIN_CIN=$m"
  MYDET_WINO4_MIN_CIN=$m timeout -k 5 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-215
done
MYDET_WINO4_MIN_CIN=32 timeout -k 5 300 python tools/profile_layers.py | grep -E "32->64 k3s1|total"
MYDET_WINO4_MIN_CIN=32 timeout -k 5 300 python -m pytest tests/test_gpu_model.py -q -x -k "yolo or golden or full_size" 2>&1 | tail -3
} 2>&1 | grep -v amdgpu.ids > $O/exp22.txt
cat $O/exp22.txt
