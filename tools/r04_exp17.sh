#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
{
timeout -k 5 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "streamk" 2>&1 | tail -15
for v in 0 1 2; do
  echo "== MYDET_PW_STREAMK=$v"
  for s in "256 128 80" "512 256 40" "1024 512 20" "128 64 160" "64 32 320" "256 255 80" "384 128 80" "768 256 40"; do set -- $s
    MYDET_PW_STREAMK=$v timeout -k 5 120 python tools/bench_conv.py --cin $1 --cout $2 --k 1 --hw $3 || exit 1
  done
done
} 2>&1 | grep -v amdgpu.ids > $O/exp17.txt
cat $O/exp17.txt
