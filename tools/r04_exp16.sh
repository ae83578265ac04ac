#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
{
for f in 0 24 15 0 24; do
  echo "== DEEP_LANES_FROM=$f (eager)"
  MYDET_DEEP_LANES_FROM=$f timeout -k 5 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --eager 2>&1 | tail -1 | cut -c1-215
done
} 2>&1 | grep -v amdgpu.ids > $O/exp16.txt
cat $O/exp16.txt
