#!/bin/bash
# board power / clocks sampled by rocm-smi while the headline step replays (split-bf16 on and off)
for s in 1 0; do
  echo "== MYDET_CONV_SPLIT_BF16=$s"
  MYDET_CONV_SPLIT_BF16=$s timeout -k 10 120 python bench.py --steps 1500 --warmup 5 --no-cpu-baseline --no-other-configs --parity-images 1 > /tmp/pw_bench_$s.json 2>/dev/null &
  BP=$!
  sleep 14
  for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | tr '\n' ';'; echo; sleep 2; done
  wait $BP
  python -c "
import json
o=json.loads(open('/tmp/pw_bench_$s.json').read().strip().splitlines()[-1]); print('value', o['value'], 'ms', o['ms_per_step'])"
done
rocm-smi --showmaxpower 2>/dev/null | grep -i power
