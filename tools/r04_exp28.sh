#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
{
for d in 0 1 2 0; do
  echo "== dbg $d (0 normal; 1 no U DMA after stage 0; 2 = 1 + U fragments by 9 register loads per wave and stage)"
  for s in "128 256 80" "256 512 40" "512 1024 20"; do set -- $s
    MYDET_W4_DBG=$d timeout -k 5 120 python tools/bench_conv.py --cin $1 --cout $2 --hw $3 --res --wino4 || exit 1
  done
done
} 2>&1 | grep -v amdgpu.ids > $O/exp28.txt
cat $O/exp28.txt
