#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
{
for v in 1 2; do
  echo "== MYDET_W4_INPUT=$v correctness"
  MYDET_W4_INPUT=$v timeout -k 5 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "winograd4" 2>&1 | tail -2 || exit 1
done
for v in 0 1 2; do
  echo "== MYDET_W4_INPUT=$v: transform launch only, then the pair"
  for s in "64 128 160" "128 256 80" "256 512 40" "512 1024 20"; do set -- $s
    MYDET_W4_INPUT=$v MYDET_W4_DBG=4 timeout -k 5 120 python tools/bench_conv.py --cin $1 --cout $2 --hw $3 --res --wino4 || exit 1
    MYDET_W4_INPUT=$v timeout -k 5 120 python tools/bench_conv.py --cin $1 --cout $2 --hw $3 --res --wino4 || exit 1
  done
done
for v in 0 1 2; do
  echo "== MYDET_W4_INPUT=$v: model"
  MYDET_W4_INPUT=$v timeout -k 5 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-330
done
} 2>&1 | grep -v amdgpu.ids > $O/exp8.txt
cat $O/exp8.txt
