#!/bin/bash
# Round-2 measurement battery, part 1: bench lines (eager + hipGraph replay) for every configuration.
# Outputs under gpurun_out/r02/; copy the judged ones into profiles/.
# every GPU command runs under `timeout -k 5`: an abort or a stuck process cannot hold the GPU lease for minutes
T=${MYDET_TOOL_TIMEOUT:-300}
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O
cd $R
b() { name=$1; shift; timeout -k 5 $T python bench.py "$@" 2>$O/$name.err | tail -1 > $O/$name.json; python - <<PY
import json; d=json.load(open('$O/$name.json')); print('$name', d.get('value'), d.get('unit'), 'ms/step', d.get('ms_per_step'), 'frac', (d.get('roofline') or {}).get('frac'))
PY
}
b r02_bench_yolov3_b32_640 --steps 20 --warmup 5
b r02_bench_yolov3_b32_640_graph --steps 20 --warmup 5 --graph
b r02_bench_yolov3_b32_512 --steps 20 --warmup 5 --size 512
b r02_bench_yolov3_b1_512_graph --steps 200 --warmup 20 --batch 1 --size 512 --graph
b r02_bench_efficientdet-d1_b16_640 --steps 20 --warmup 5 --config efficientdet-d1
b r02_bench_efficientdet-d1_b16_640_graph --steps 20 --warmup 5 --config efficientdet-d1 --graph
b r02_bench_d1_fcs2_atss_b32_640 --steps 20 --warmup 5 --config d1_fcs2_atss
b r02_bench_d1_fcs2_atss_b32_640_graph --steps 20 --warmup 5 --config d1_fcs2_atss --graph
timeout -k 5 $T python bench.py --nms-worst 2>/dev/null | tail -1 > $O/r02_nms_worst.json; cut -c1-400 $O/r02_nms_worst.json
