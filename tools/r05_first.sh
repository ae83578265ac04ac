#!/bin/bash
# Round-5 first GPU call: the whole -m gpu suite, the default bench line (headline + other_configs), the occupancy hook,
# and a kernel trace of one D1 lane (batch 8) for the launch-chain analysis.
T=${MYDET_TOOL_TIMEOUT:-420}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O
cd $R
timeout -k 10 $T python -m pytest tests -m gpu -x -q > $O/first_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/first_pytest.log
tail -5 $O/first_pytest.log
timeout -k 10 $T python bench.py --steps 20 --warmup 5 2>$O/first_bench.err | tail -1 > $O/first_bench.json; echo "bench rc=$?"
cut -c1-400 $O/first_bench.json
timeout -k 10 60 python - > $O/first_occupancy.txt 2>&1 <<'PY'
import ctypes, torch
from mydetection_amd import _lib
torch.zeros(1, device='cuda')
lib = _lib.lib()
for cfg in (0, 1, 2, 3, 6, 8, 9):
    a = ctypes.c_int(0)
    print('cfg', cfg, 'occupancy', lib.mydet_conv_igemm_occupancy(cfg, ctypes.byref(a)), 'assumed', a.value)
PY
cat $O/first_occupancy.txt
