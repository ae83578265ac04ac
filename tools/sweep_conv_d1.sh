#!/bin/bash
# tile-config sweep on EfficientDet-D1 pointwise shapes (batch 16)
# every GPU command runs under `timeout -k 5`: an abort or a stuck process cannot hold the GPU lease for minutes
T=${MYDET_TOOL_TIMEOUT:-300}
CFGS=${1:-"0 1 2 3 6 8"}
for shape in "--cin 16 --cout 96 --k 1 --hw 320 --act 2" "--cin 24 --cout 144 --k 1 --hw 160 --act 2" "--cin 40 --cout 240 --k 1 --hw 80 --act 2" "--cin 112 --cout 672 --k 1 --hw 40 --act 2" "--cin 192 --cout 1152 --k 1 --hw 20 --act 2" "--cin 144 --cout 24 --k 1 --hw 160 --act 0 --gate --res" "--cin 672 --cout 112 --k 1 --hw 40 --act 0 --gate --res" "--cin 1152 --cout 192 --k 1 --hw 20 --act 0 --gate --res" "--cin 88 --cout 88 --k 1 --hw 80 --act 2" "--cin 88 --cout 720 --k 1 --hw 80 --act 0"; do
  for cfg in $CFGS; do
    echo -n "cfg=$cfg "; MYDET_CONV_CFG=$cfg timeout -k 5 $T python tools/bench_conv.py $shape --batch 16 --reps 10 2>&1 | tail -1
  done
done
