#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
{
for cfg in "" 0 1 3 6 8; do
  echo "== cfg ${cfg:-default}"
  MYDET_CONV_CFG=$cfg timeout -k 5 120 python tools/bench_conv.py --cin 32 --cout 64 --k 3 --s 2 --hw 640 || exit 1
  MYDET_CONV_CFG=$cfg timeout -k 5 120 python tools/bench_conv.py --cin 64 --cout 128 --k 3 --s 2 --hw 320 || exit 1
  MYDET_CONV_CFG=$cfg timeout -k 5 120 python tools/bench_conv.py --cin 128 --cout 256 --k 3 --s 2 --hw 160 || exit 1
  MYDET_CONV_CFG=$cfg timeout -k 5 120 python tools/bench_conv.py --cin 128 --cout 64 --k 1 --s 1 --hw 160 || exit 1
  MYDET_CONV_CFG=$cfg timeout -k 5 120 python tools/bench_conv.py --cin 256 --cout 255 --k 1 --s 1 --hw 80 || exit 1
done
} 2>&1 | grep -v amdgpu.ids > $O/exp24.txt
cat $O/exp24.txt
