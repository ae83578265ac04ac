#!/bin/bash
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "split_bf16" 2>&1 | tail -3 || exit 1
for r in 1 2; do for v in "0 6000" "64 20000" "0 6000" "64 20000"; do set -- $v; C=$1; R=$2; for c in "efficientdet-d1 16" "d1_fcs2_atss 32"; do set -- $c; MYDET_B3_GATED_MIN_COUT=$C MYDET_B3_GATED_MIN_ROWS=$R timeout -k 10 100 python bench.py --config $1 --batch $2 --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs --no-power-probe 2>/dev/null | tail -1 | python -c "
import sys,json
o=json.loads(sys.stdin.read()); print('gated cout>=$C rows>=$R $1', o['value'], o['ms_per_step'], o['parity_check']['ok'], o['parity_check']['max_score_err'], {k:round(v['ms_per_step'],3) for k,v in o['stages'].items() if 'igemm' in k})"; done; done; done
