#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
{
for d in 0 1 2 3; do
  echo "== dbg $d (0 normal, 1 every stage reads the same 36 KB, 2 no DMA after stage 1, 3 U always stage 0)"
  MYDET_W4_DBG=$d timeout -k 5 120 python tools/bench_conv.py --cin 128 --cout 256 --hw 80 --res --wino4 || exit 1
  MYDET_W4_DBG=$d timeout -k 5 120 python tools/bench_conv.py --cin 256 --cout 512 --hw 40 --res --wino4 || exit 1
  MYDET_W4_DBG=$d timeout -k 5 120 python tools/bench_conv.py --cin 512 --cout 1024 --hw 20 --res --wino4 || exit 1
done
} 2>&1 | grep -v amdgpu.ids > $O/exp3.txt
cat $O/exp3.txt
