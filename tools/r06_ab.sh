#!/bin/bash
# A/B inside one GPU call: bench.py with and without an environment switch, alternating, N runs each.
#   tools/r06_ab.sh "<ENV=VAL for B>" "<bench args>" [runs]
R=$GRAFT_REPO_ROOT; cd $R
SW="$1"; ARGS="$2"; N=${3:-2}
for rep in $(seq 1 $N); do
  for mode in A B; do
    if [ $mode = B ]; then E="$SW"; else E="MYDET_NOP=1"; fi
    env $E timeout -k 5 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-power-probe $ARGS 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); st=d['stages']; print('$mode', '$E', '$ARGS', d['value'], d['ms_per_step'], 'parity', (d.get('parity_check') or {}).get('ok'), 'err', (d.get('parity_check') or {}).get('max_score_err'), {k: v['ms_per_step'] for k, v in st.items() if k.startswith('conv')})"
  done
done
