#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
{
for l in 2 4 2 4; do
  echo "== fcos lanes $l"
  timeout -k 5 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --config d1_fcs2_atss --lanes $l 2>/dev/null | tail -1 | cut -c1-215
done
for l in 2 4; do
  echo "== d1 lanes $l"
  timeout -k 5 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --config efficientdet-d1 --lanes $l 2>/dev/null | tail -1 | cut -c1-215
done
} 2>&1 | grep -v amdgpu.ids > $O/exp25.txt
cat $O/exp25.txt
