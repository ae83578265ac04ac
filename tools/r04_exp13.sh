#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
{
for d in 8 9 11; do
  for s in "256 128 1 1 80" "512 256 1 1 40" "1024 512 1 1 20" "128 256 3 2 160"; do
    MYDET_IG_DBG=$d timeout -k 5 120 python tools/r04_igclock.py $s 32 || exit 1
  done
done
} 2>&1 | grep -v amdgpu.ids > $O/exp13.txt
cat $O/exp13.txt
