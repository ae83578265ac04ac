#!/bin/bash
# Round-3 measurement battery, part 1: bench lines (hipGraph replay = the default, and --eager) for every configuration.
# Outputs under gpurun_out/r03/; copy the judged ones into profiles/.
# every GPU command runs under `timeout -k 5`: an abort or a stuck process cannot hold the GPU lease for minutes
T=${MYDET_TOOL_TIMEOUT:-300}
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
b() { name=$1; shift; timeout -k 5 $T python bench.py "$@" 2>$O/$name.err | tail -1 > $O/$name.json; python - <<PY
import json; d=json.load(open('$O/$name.json')); print('$name', d.get('value'), d.get('unit'), 'ms/step', d.get('ms_per_step'), 'frac', (d.get('roofline') or {}).get('frac'))
PY
}
b r03_bench_yolov3_b32_640 --steps 20 --warmup 5
b r03_bench_yolov3_b32_640_eager --steps 20 --warmup 5 --eager
b r03_bench_yolov3_b32_512 --steps 20 --warmup 5 --size 512
b r03_bench_yolov3_b1_512 --steps 200 --warmup 20 --batch 1 --size 512
b r03_bench_efficientdet-d1_b16_640 --steps 20 --warmup 5 --config efficientdet-d1
b r03_bench_efficientdet-d1_b16_640_eager --steps 20 --warmup 5 --config efficientdet-d1 --eager
b r03_bench_d1_fcs2_atss_b32_640 --steps 20 --warmup 5 --config d1_fcs2_atss
b r03_bench_d1_fcs2_atss_b32_640_eager --steps 20 --warmup 5 --config d1_fcs2_atss --eager
timeout -k 5 $T python bench.py --nms-worst 2>/dev/null | tail -1 > $O/r03_nms_worst.json; cut -c1-400 $O/r03_nms_worst.json
