#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
{
for m in 999 512 256 999 512 256; do
  echo "== W4B_MIN_CIN=$m (999 = never) W4_INPUT=1"
  MYDET_W4B_MIN_CIN=$m MYDET_W4_INPUT=1 timeout -k 5 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-215
done
MYDET_W4B_MIN_CIN=999 MYDET_W4_INPUT=1 timeout -k 5 300 python tools/profile_layers.py | grep wino4
MYDET_W4B_MIN_CIN=256 MYDET_W4_INPUT=1 timeout -k 5 300 python tools/profile_layers.py | grep wino4
} 2>&1 | grep -v amdgpu.ids > $O/exp11.txt
cat $O/exp11.txt
