#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
{
for b in 20 32 40 60; do
for d in 0 2 4; do
  echo "== batch $b dbg $d (0 normal, 2 no DMA after stage 1, 4 transform launch only)"
  MYDET_W4_DBG=$d timeout -k 5 120 python tools/bench_conv.py --cin 512 --cout 1024 --hw 20 --res --wino4 --batch $b || exit 1
done
done
} 2>&1 | grep -v amdgpu.ids > $O/exp4.txt
cat $O/exp4.txt
