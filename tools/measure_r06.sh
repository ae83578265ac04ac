#!/bin/bash
# Round-6 measurement battery (one GPU call): bench lines of every configuration (hipGraph replay = default, and --eager),
# rocprofv3 kernel stats of `bench.py --profile` (program directly behind `--`; kernel counts = launches per step x passes),
# PMC traffic passes (FETCH_SIZE / WRITE_SIZE separately, eager), per-layer table.  Outputs under gpurun_out/r06m/; the
# judged ones are copied into profiles/ by hand.  Every GPU command runs under `timeout -k 5`.
T=${MYDET_TOOL_TIMEOUT:-300}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06m; mkdir -p $O
cd $R
b() { name=$1; shift; timeout -k 5 $T python bench.py "$@" 2>$O/$name.err | tail -1 > $O/$name.json || echo "$name rc $?"; python - <<PY
import json; d=json.load(open('$O/$name.json')); print('$name', d.get('value'), 'ms/step', d.get('ms_per_step'), 'frac', (d.get('roofline') or {}).get('frac'), 'parity', (d.get('parity_check') or {}).get('ok'))
PY
}
b r06_bench_yolov3_b32_640 --steps 20 --warmup 5 &&
b r06_bench_yolov3_b32_640_eager --steps 20 --warmup 5 --eager &&
b r06_bench_yolov3_b32_512 --steps 20 --warmup 5 --size 512 &&
b r06_bench_yolov3_b1_512 --steps 200 --warmup 20 --batch 1 --size 512 &&
b r06_bench_efficientdet-d1_b16_640 --steps 20 --warmup 5 --config efficientdet-d1 &&
b r06_bench_efficientdet-d1_b16_640_eager --steps 20 --warmup 5 --config efficientdet-d1 --eager &&
b r06_bench_d1_fcs2_atss_b32_640 --steps 20 --warmup 5 --config d1_fcs2_atss &&
b r06_bench_d1_fcs2_atss_b32_640_eager --steps 20 --warmup 5 --config d1_fcs2_atss --eager || exit 1
timeout -k 5 $T python bench.py --nms-worst 2>/dev/null | tail -1 > $O/r06_nms_worst.json
timeout -k 5 $T python tools/profile_layers.py > $O/r06_layers_yolov3_b32_640.txt 2>/dev/null
timeout -k 5 $T python tools/profile_layers.py --config efficientdet-d1 --batch 8 --reps 10 > $O/r06_layers_d1_lane_b8_640.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp
prof() { TAG=$1; shift
  rm -rf $O/stats_$TAG $O/fetch_$TAG $O/write_$TAG
  timeout -k 5 $T rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$TAG -- python3 $R/bench.py --profile --steps 10 --warmup 8 "$@" > $O/stats_$TAG.log 2>&1 || { echo "stats $TAG failed"; tail -3 $O/stats_$TAG.log; return 1; }
  cp $(find $O/stats_$TAG -name '*kernel_stats.csv' | head -1) $O/r06_kernel_stats_$TAG.csv
  grep "\"metric\"" $O/stats_$TAG.log | tail -1 | cut -c1-600 > $O/r06_kernel_stats_$TAG.cmdline.json
  timeout -k 5 $T rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_$TAG -- python3 $R/bench.py --profile --eager --steps 3 --warmup 1 "$@" > $O/fetch_$TAG.log 2>&1 || { echo "fetch $TAG failed"; return 1; }
  timeout -k 5 $T rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write_$TAG -- python3 $R/bench.py --profile --eager --steps 3 --warmup 1 "$@" > $O/write_$TAG.log 2>&1 || { echo "write $TAG failed"; return 1; }
  python3 $R/tools/pmc_traffic.py $O/fetch_$TAG $O/write_$TAG > $O/r06_pmc_traffic_$TAG.json
  rm -rf $O/fetch_$TAG $O/write_$TAG $O/stats_$TAG
  head -8 $O/r06_kernel_stats_$TAG.csv | cut -c1-160
}
prof yolov3_b32_640 && prof d1_b16_640 --config efficientdet-d1 && prof fcos_b32_640 --config d1_fcs2_atss
