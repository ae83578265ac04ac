#!/bin/bash
# Round 6, first GPU call: (1) stand-alone v_pk_fma_f32 probe, (2) GPU suite on the no-packed-f32 build, (3) A/B of the round-5
# (packed) library against this build on the three bench configurations, alternating, two runs each.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06a; mkdir -p $O; cd $R
timeout -k 5 120 tools/bin/hw_pk_probe 64 > $O/pk_probe_64.txt 2>&1; echo "probe rc $?"; tail -40 $O/pk_probe_64.txt
timeout -k 5 120 tools/bin/hw_pk_probe 16 > $O/pk_probe_16.txt 2>&1; echo "probe16 rc $?"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "tests rc $?"; tail -3 $O/gpu_tests.log
ab() { for rep in 1 2; do for lib in tmp_libs/libmydet_r05_pk.so mydetection_amd/lib/libmydet_hip.so; do
  MYDET_LIB_PATH=$R/$lib timeout -k 5 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-power-probe "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', '$*', d['value'], d['ms_per_step'], (d.get('parity_check') or {}).get('ok'))"; done; done; }
ab > $O/ab_yolo.txt 2>&1; cat $O/ab_yolo.txt
ab --config efficientdet-d1 > $O/ab_d1.txt 2>&1; cat $O/ab_d1.txt
ab --config d1_fcs2_atss > $O/ab_fcos.txt 2>&1; cat $O/ab_fcos.txt
