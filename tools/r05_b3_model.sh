#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O
cd $R
for rep in 1 2; do
for v in "1 0 4" "1 0 8" "0 0 4"; do
  set -- $v
  MYDET_CONV_SPLIT_BF16=$1 MYDET_B3_WIDE=$2 MYDET_B3_WAVES=$3 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline --parity-images 1 2>/dev/null | tail -1 | python -c "import sys,json; o=json.loads(sys.stdin.read()); print('split_bf16=$1 wide=$2 waves=$3', o['value'], o['ms_per_step'], {k:(v['ms_per_step']) for k,v in o['stages'].items() if k.startswith('conv')})"
done
done
