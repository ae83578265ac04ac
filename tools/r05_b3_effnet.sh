#!/bin/bash
# split-bf16 on the EfficientNet expand convs: A/B in one call
O=gpurun_out/b3_effnet; mkdir -p $O; : > $O/ab.txt
for rep in 1 2; do
for v in "0 0" "1 8192" "1 3000"; do
  set -- $v
  for spec in "efficientdet-d1 16" "d1_fcs2_atss 32"; do
    set -- $v $spec
    MYDET_B3_EXPAND_MIN_ROWS=$2 timeout -k 10 120 python bench.py --config $3 --batch $4 --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs --parity-images 2 2>/dev/null | tail -1 | python -c "import sys,json; o=json.loads(sys.stdin.read()); print('b3_effnet=$1 min_rows=$2', '$3', o['value'], o['ms_per_step'], o['parity_check']['ok'], o['parity_check']['max_score_err'], {k: round(v['ms_per_step'],3) for k,v in o['stages'].items() if 'igemm' in k})" | tee -a $O/ab.txt
  done
done
done
