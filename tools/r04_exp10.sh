#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
{
for cfg in "0 0" "8 32" "8 33" "8 34" "8 35" "8 36" "8 37" "8 39" "0 0"; do set -- $cfg
  echo "== variant $1 dbg $2 (stagger at group dbg-32)"
  for s in "128 256 80" "256 512 40" "512 1024 20"; do set -- $cfg $s
    MYDET_WINO4_VARIANT=$1 MYDET_W4_DBG=$2 timeout -k 5 120 python tools/bench_conv.py --cin $3 --cout $4 --hw $5 --res --wino4 --reps 40 || exit 1
  done
done
} 2>&1 | grep -v amdgpu.ids > $O/exp10.txt
cat $O/exp10.txt
