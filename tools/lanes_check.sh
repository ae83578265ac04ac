#!/bin/bash
# Batch lanes: the graph tests, then the default bench line of the three model configs ('auto' picks the lanes).
T=${MYDET_TOOL_TIMEOUT:-300}
O=gpurun_out/r03; mkdir -p $O
timeout -k 5 600 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "lanes or hipgraph or graph or detector or predict or full_size" > $O/lanes_pytest.log 2>&1 || { tail -30 $O/lanes_pytest.log; exit 1; }
tail -2 $O/lanes_pytest.log
for cfg in "efficientdet-d1" "d1_fcs2_atss" "yolov3_80"; do
  timeout -k 5 $T python bench.py --config $cfg --steps 30 --warmup 5 --no-cpu-baseline > $O/auto_${cfg}.json 2> $O/auto_${cfg}.err || { tail -5 $O/auto_${cfg}.err; exit 1; }
  python - <<PY
import json
d = json.loads(open('$O/auto_${cfg}.json').read().strip().splitlines()[-1])
print('$cfg', d['config']['batch_lanes'], d['value'], d['ms_per_step'])
PY
done
