#!/bin/bash
# Round-4 baseline at the start of the round: headline bench line + per-layer table (HIP events)
T=${MYDET_TOOL_TIMEOUT:-300}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout -k 5 $T python bench.py --steps 20 --warmup 5 2>$O/base_bench.err | tail -1 > $O/base_bench_yolov3.json &&
timeout -k 5 $T python tools/profile_layers.py > $O/base_layers_yolov3.txt 2>&1 &&
timeout -k 5 $T python bench.py --steps 200 --warmup 20 --batch 1 --size 512 --no-cpu-baseline 2>>$O/base_bench.err | tail -1 > $O/base_bench_b1.json &&
cut -c1-300 $O/base_bench_yolov3.json && cat $O/base_layers_yolov3.txt
