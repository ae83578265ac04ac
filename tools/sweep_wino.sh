#!/bin/bash
# Direct vs Winograd (both workgroup shapes) on the 3x3 stride-1 layers of YOLOv3-80 at batch 32, 640x640.
# every GPU command runs under `timeout -k 5`: an abort or a stuck process cannot hold the GPU lease for minutes
T=${MYDET_TOOL_TIMEOUT:-300}
for cfg in "32 64 320" "64 128 160" "128 256 80" "256 512 40" "512 1024 20"; do
  set -- $cfg
  [ -n "$SKIP_DIRECT" ] || timeout -k 5 $T python tools/bench_conv.py --cin $1 --cout $2 --hw $3 --res || exit 1
  MYDET_WINO_NW=4 timeout -k 5 $T python tools/bench_conv.py --cin $1 --cout $2 --hw $3 --res --wino || exit 1
  MYDET_WINO_NW=8 timeout -k 5 $T python tools/bench_conv.py --cin $1 --cout $2 --hw $3 --res --wino || exit 1
done
