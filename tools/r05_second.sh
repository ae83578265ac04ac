#!/bin/bash
# Round-5 second GPU call: full -m gpu suite, default bench line, cfg-6 slot A/B, launch-chain traces of the D1 lane.
T=${MYDET_TOOL_TIMEOUT:-420}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O
cd $R
timeout -k 10 $T python -m pytest tests -m gpu -q > $O/second_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/second_pytest.log
tail -8 $O/second_pytest.log
timeout -k 10 $T python bench.py --steps 20 --warmup 5 2>$O/second_bench.err | tail -1 > $O/second_bench.json; echo "bench rc=$?"
cut -c1-300 $O/second_bench.json
for per in 5 4; do
  for cfg in "efficientdet-d1" "d1_fcs2_atss" "yolov3_80"; do
    MYDET_CFG6_PER_CU=$per timeout -k 10 120 python bench.py --config $cfg --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs --parity-images 1 2>/dev/null | tail -1 | python -c "import sys,json; o=json.loads(sys.stdin.read()); print('cfg6_per_cu=$per', '$cfg', o['value'], o['ms_per_step'])" | tee -a $O/second_cfg6_ab.txt
  done
done
cd /tmp && export TMPDIR=/tmp
for spec in "d1_b8_l1 efficientdet-d1 8 1" "d1_b16_l2 efficientdet-d1 16 2" "d1_b1_l1 efficientdet-d1 1 1"; do
  set -- $spec
  D=$O/trace_$1; rm -rf $D; mkdir -p $D
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $R/bench.py --profile --config $2 --batch $3 --lanes $4 --steps 5 --warmup 3 > $D/log.txt 2>&1 || { tail -5 $D/log.txt; }
  python3 $R/tools/chain_trace.py $D $4 > $O/chain_$1.txt 2>&1
  tail -1 $O/chain_$1.txt
  rm -rf $D
done
