#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -q -k "split_bf16 or conv_igemm or upcat" > $O/b3_pytest.log 2>&1; echo "kernels rc=$?"; tail -4 $O/b3_pytest.log
timeout -k 10 400 python -m pytest tests/test_gpu_model.py -q -k "golden or oracle or full_size or consistency or hipgraph" > $O/b3_pytest_model.log 2>&1; echo "model rc=$?"; tail -4 $O/b3_pytest_model.log
for v in 1 0; do
  MYDET_CONV_SPLIT_BF16=$v timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; o=json.loads(sys.stdin.read()); print('split_bf16=$v', o['value'], o['ms_per_step'], {k:(v['launches_per_step'], v['ms_per_step'], v.get('mfma_frac')) for k,v in o['stages'].items()}, o['parity_check']['ok'], o['parity_check']['max_score_err'], o['parity_check']['kept_ids_jaccard']['min'])"
done
