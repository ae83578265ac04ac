#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
{
for b in 4 8 10 12 16 20 30 32; do
  MYDET_W4_DBG=8 SOAK_S=0.5 timeout -k 5 120 python tools/r04_clock.py 512 1024 20 $b || exit 1
done
} 2>&1 | grep -v amdgpu.ids > $O/exp6.txt
cat $O/exp6.txt
