#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
{
for d in 8 9 10 11; do
  echo "== dbg $d  (8 normal, 9 every stage the same 36 KB, 10 no DMA after stage 1, 11 U always stage 0)"
  MYDET_W4_DBG=$d SOAK_S=1 timeout -k 5 120 python tools/r04_clock.py 512 1024 20 20 || exit 1
  MYDET_W4_DBG=$d SOAK_S=1 timeout -k 5 120 python tools/r04_clock.py 256 512 40 32 || exit 1
  MYDET_W4_DBG=$d SOAK_S=1 timeout -k 5 120 python tools/r04_clock.py 128 256 80 32 || exit 1
done
} 2>&1 | grep -v amdgpu.ids > $O/exp5.txt
cat $O/exp5.txt
