#!/bin/bash
# quantisation experiment: F(4x4) GEMM launch vs number of items; 1x1 igemm vs tile config
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
{
for co in 448 480 512 544; do timeout -k 5 120 python tools/bench_conv.py --cin 256 --cout $co --hw 40 --res --wino4 || exit 1; done
for co in 224 240 256 288; do timeout -k 5 120 python tools/bench_conv.py --cin 128 --cout $co --hw 80 --res --wino4 || exit 1; done
for co in 960 1024 1088; do timeout -k 5 120 python tools/bench_conv.py --cin 512 --cout $co --hw 20 --res --wino4 || exit 1; done
for cfg in 0 1 3 8; do
  echo "cfg $cfg"
  MYDET_CONV_CFG=$cfg timeout -k 5 120 python tools/bench_conv.py --cin 256 --cout 128 --k 1 --hw 80 || exit 1
  MYDET_CONV_CFG=$cfg timeout -k 5 120 python tools/bench_conv.py --cin 512 --cout 256 --k 1 --hw 40 || exit 1
  MYDET_CONV_CFG=$cfg timeout -k 5 120 python tools/bench_conv.py --cin 1024 --cout 512 --k 1 --hw 20 || exit 1
done
} > $O/exp1.txt 2>&1
cat $O/exp1.txt
