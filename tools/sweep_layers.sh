#!/bin/bash
# F(2x2) Winograd, direct 1x1 and stride-2 layers of YOLOv3-80 at batch 32, 640x640 (compile-flag A/B runs).
# every GPU command runs under `timeout -k 5`: an abort or a stuck process cannot hold the GPU lease for minutes
T=${MYDET_TOOL_TIMEOUT:-300}
for cfg in "64 128 160" "128 256 80" "256 512 40" "512 1024 20"; do
  set -- $cfg
  timeout -k 5 $T python tools/bench_conv.py --cin $1 --cout $2 --hw $3 --res --wino || exit 1
done
timeout -k 5 $T python tools/bench_conv.py --cin 256 --cout 128 --hw 80 --k 1 || exit 1
timeout -k 5 $T python tools/bench_conv.py --cin 512 --cout 256 --hw 40 --k 1 || exit 1
timeout -k 5 $T python tools/bench_conv.py --cin 1024 --cout 512 --hw 20 --k 1 || exit 1
timeout -k 5 $T python tools/bench_conv.py --cin 128 --cout 256 --hw 160 --k 3 --s 2 || exit 1
timeout -k 5 $T python tools/bench_conv.py --cin 256 --cout 512 --hw 80 --k 3 --s 2 || exit 1
