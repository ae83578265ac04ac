#!/bin/bash
# F(4x4) GEMM launch: 64-tile x 32-channel workgroup variants (MYDET_WINO4_VARIANT = 4 | 8) against the shipped 32 x 32
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
{
for v in 4 8; do
  echo "== variant $v: correctness"
  MYDET_WINO4_VARIANT=$v timeout -k 5 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "winograd4" 2>&1 | tail -5 || exit 1
done
for v in 0 4 8; do
  echo "== variant $v"
  MYDET_WINO4_VARIANT=$v timeout -k 5 120 python tools/bench_conv.py --cin 64 --cout 128 --hw 160 --res --wino4 || exit 1
  MYDET_WINO4_VARIANT=$v timeout -k 5 120 python tools/bench_conv.py --cin 128 --cout 256 --hw 80 --res --wino4 || exit 1
  MYDET_WINO4_VARIANT=$v timeout -k 5 120 python tools/bench_conv.py --cin 256 --cout 512 --hw 40 --res --wino4 || exit 1
  MYDET_WINO4_VARIANT=$v timeout -k 5 120 python tools/bench_conv.py --cin 512 --cout 1024 --hw 20 --res --wino4 || exit 1
done
} 2>&1 | grep -v amdgpu.ids > $O/exp2.txt
cat $O/exp2.txt
