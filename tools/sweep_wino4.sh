#!/bin/bash
# F(2x2) vs F(4x4) Winograd on the 3x3 stride-1 layers of YOLOv3-80 at batch 32, 640x640.
# every GPU command runs under `timeout -k 5`: an abort or a stuck process cannot hold the GPU lease for minutes
T=${MYDET_TOOL_TIMEOUT:-300}
for cfg in "64 128 160" "128 256 80" "256 512 40" "512 1024 20"; do
  set -- $cfg
  timeout -k 5 $T python tools/bench_conv.py --cin $1 --cout $2 --hw $3 --res --wino || exit 1
  timeout -k 5 $T python tools/bench_conv.py --cin $1 --cout $2 --hw $3 --res --wino4 || exit 1
done
