#!/bin/bash
# in-launch squeeze-excite tail: kernel tests, EfficientDet model tests, A/B against the separate launch
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "se_gate or dwconv or mbconv or stem_dw or se_tail" > $O/se_pytest.log 2>&1; echo "pytest kernels rc=$?"; tail -4 $O/se_pytest.log
timeout -k 10 500 python -m pytest tests/test_gpu_model.py -x -q -k "effdet or lanes or retina" > $O/se_pytest_model.log 2>&1; echo "pytest model rc=$?"; tail -4 $O/se_pytest_model.log
: > $O/se_ab.txt
for rep in 1 2; do
for v in 1 0; do
  for spec in "efficientdet-d1 16" "d1_fcs2_atss 32"; do
    set -- $spec
    MYDET_SE_IN_DW=$v timeout -k 10 120 python bench.py --config $1 --batch $2 --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs --parity-images 2 2>/dev/null | tail -1 | python -c "import sys,json; o=json.loads(sys.stdin.read()); print('se_in_dw=$v', '$1', o['value'], o['ms_per_step'], o['launches_per_lane'], o['parity_check']['ok'], {k: round(v['ms_per_step'],3) for k,v in o['stages'].items()})" | tee -a $O/se_ab.txt
  done
done
done
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  D=$O/trace_se$v; rm -rf $D; mkdir -p $D
  MYDET_SE_IN_DW=$v timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $R/bench.py --profile --config efficientdet-d1 --batch 8 --lanes 1 --steps 5 --warmup 3 > $D/log.txt 2>&1 || { tail -5 $D/log.txt; }
  python3 $R/tools/chain_trace.py $D 1 > $O/chain_d1_b8_se$v.txt 2>&1
  tail -1 $O/chain_d1_b8_se$v.txt
  rm -rf $D
done
