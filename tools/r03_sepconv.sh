#!/bin/bash
T=${MYDET_TOOL_TIMEOUT:-300}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "sepconv" 2>&1 | tail -3 || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "effdet_family_vs_reference or stiff" 2>&1 | tail -3 || exit 1
for ws in 1 0; do
  MYDET_SEPCONV_WS=$ws timeout -k 5 $T python bench.py --steps 20 --warmup 5 --config efficientdet-d1 --graph --no-cpu-baseline > $O/d1_ws$ws.json 2>$O/d1_ws$ws.err
  python - <<PY
import json; d=json.loads(open('$O/d1_ws$ws.json').read().strip().split('\n')[-1]); print('D1 ws=$ws', d['value'], d['ms_per_step'], d['stages']['sepconv_nodes'])
PY
done
MYDET_SEPCONV_WS=1 timeout -k 5 $T python bench.py --steps 20 --warmup 5 --config d1_fcs2_atss --graph --no-cpu-baseline > $O/fcos_ws1.json 2>$O/fcos_ws1.err
python - <<PY
import json; d=json.loads(open('$O/fcos_ws1.json').read().strip().split('\n')[-1]); print('FCOS ws=1', d['value'], d['ms_per_step'], d['stages']['sepconv_nodes'])
PY
