#!/bin/bash
# Batch lanes (MYDET_LANES): the same bench line with the batch cut into 1 / 2 / 4 concurrent lanes.
T=${MYDET_TOOL_TIMEOUT:-300}
O=gpurun_out/r03; mkdir -p $O
for cfg in "efficientdet-d1" "d1_fcs2_atss" "yolov3_80"; do
  for L in 1 2 4; do
    MYDET_LANES=$L timeout -k 5 $T python bench.py --config $cfg --steps 30 --warmup 5 --no-cpu-baseline > $O/lanes_${cfg}_$L.json 2> $O/lanes_${cfg}_$L.err || { tail -5 $O/lanes_${cfg}_$L.err; exit 1; }
    python - <<PY
import json
d = json.loads(open('$O/lanes_${cfg}_$L.json').read().strip().splitlines()[-1])
print('$cfg', 'lanes', $L, d['value'], d['ms_per_step'])
PY
  done
done
