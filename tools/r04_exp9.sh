#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
{
echo "== variant 8 staggered: correctness"
MYDET_WINO4_VARIANT=8 timeout -k 5 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "winograd4" 2>&1 | tail -3
for cfg in "0 0" "8 16" "8 0"; do set -- $cfg
  echo "== variant $1 dbg $2 (dbg 16 = no stagger)"
  for s in "128 256 80" "256 512 40" "512 1024 20"; do set -- $cfg $s
    MYDET_WINO4_VARIANT=$1 MYDET_W4_DBG=$2 timeout -k 5 120 python tools/bench_conv.py --cin $3 --cout $4 --hw $5 --res --wino4 || exit 1
  done
done
} 2>&1 | grep -v amdgpu.ids > $O/exp9.txt
cat $O/exp9.txt
