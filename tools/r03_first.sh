#!/bin/bash
# Round-3 first GPU pass: parity suite, bench lines, per-layer tables.
T=${MYDET_TOOL_TIMEOUT:-400}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout -k 5 $T python bench.py --steps 20 --warmup 5 > $O/bench_yolov3.json 2>$O/bench_yolov3.err; tail -c 600 $O/bench_yolov3.json | cut -c1-300
timeout -k 5 $T python bench.py --steps 20 --warmup 5 --config efficientdet-d1 --no-cpu-baseline > $O/bench_d1_graph.json 2>$O/bench_d1.err; cut -c1-200 $O/bench_d1_graph.json
timeout -k 5 $T python bench.py --steps 20 --warmup 5 --config d1_fcs2_atss --no-cpu-baseline > $O/bench_fcos_graph.json 2>$O/bench_fcos.err; cut -c1-200 $O/bench_fcos_graph.json
timeout -k 5 $T python tools/profile_layers.py --config efficientdet-d1 --batch 16 > $O/layers_d1.txt 2>&1
timeout -k 5 $T python tools/profile_layers.py --config d1_fcs2_atss --batch 32 > $O/layers_fcos.txt 2>&1
timeout -k 5 $T python tools/profile_layers.py > $O/layers_yolov3.txt 2>&1
echo done
