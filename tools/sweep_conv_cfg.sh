#!/bin/bash
# sweep tile configs on the main layer shapes:  bash tools/sweep_conv_cfg.sh "0 3 8 9"
# every GPU command runs under `timeout -k 5`: an abort or a stuck process cannot hold the GPU lease for minutes
T=${MYDET_TOOL_TIMEOUT:-300}
CFGS=${1:-"0 1 3 6 8 9 10 11"}
for shape in "--cin 128 --cout 256 --hw 80 --res" "--cin 256 --cout 512 --hw 40 --res" "--cin 512 --cout 1024 --hw 20 --res" "--cin 64 --cout 128 --hw 160 --res" "--cin 32 --cout 64 --hw 320 --res" "--cin 256 --cout 128 --k 1 --hw 80" "--cin 512 --cout 256 --k 1 --hw 40" "--cin 1024 --cout 512 --k 1 --hw 20"; do
  for cfg in $CFGS; do
    echo -n "cfg=$cfg "; MYDET_CONV_CFG=$cfg timeout -k 5 $T python tools/bench_conv.py $shape --reps 10 2>&1 | tail -1
  done
done
