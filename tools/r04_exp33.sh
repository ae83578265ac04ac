#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
{
for m in 768 512 768 512; do
  echo "== yolov3 b32 512 MYDET_WINO4_MIN_ITEMS=$m"
  MYDET_WINO4_MIN_ITEMS=$m timeout -k 5 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --size 512 2>/dev/null | tail -1 | cut -c1-215
  echo "== d1 MYDET_WINO4_MIN_ITEMS=$m"
  MYDET_WINO4_MIN_ITEMS=$m timeout -k 5 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --config efficientdet-d1 2>/dev/null | tail -1 | cut -c1-215
done
} 2>&1 | grep -v amdgpu.ids > $O/exp33.txt
cat $O/exp33.txt
