#!/bin/bash
MYDET_B3_HALF_TILES=100000 timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "split_bf16" 2>&1 | tail -2 || exit 1
for r in 1 2; do for v in "0 20000" "128 20000" "128 3000" "128 6000"; do set -- $v; HT=$1; GR=$2; for c in "efficientdet-d1 16" "d1_fcs2_atss 32"; do set -- $c; MYDET_B3_HALF_TILES=$HT MYDET_B3_GATED_MIN_ROWS=$GR timeout -k 10 100 python bench.py --config $1 --batch $2 --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs --no-power-probe 2>/dev/null | tail -1 | python -c "
import sys,json
o=json.loads(sys.stdin.read()); print('half_tiles=$HT gated_rows=$GR $1', o['value'], o['ms_per_step'], o['parity_check']['ok'])"; done; done; done
