#!/bin/bash
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "se_gate or dwconv or mbconv or stem_dw" 2>&1 | tail -2 || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_model.py -x -q -k "effdet or lanes or two_lane or retina" 2>&1 | tail -2 || exit 1
SOAK=200 timeout -k 10 300 python tools/soak_lanes.py efficientdet-d1 16 2>&1 | grep lanes
SOAK=200 timeout -k 10 300 python tools/soak_lanes.py d1_fcs2_atss 32 2>&1 | grep lanes
for c in "efficientdet-d1 16" "d1_fcs2_atss 32"; do set -- $c; for r in 1 2; do timeout -k 10 100 python bench.py --config $1 --batch $2 --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "
import sys,json
o=json.loads(sys.stdin.read()); print('$1', o['value'], o['ms_per_step'], o['parity_check']['ok'])"; done; done
