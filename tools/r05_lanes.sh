#!/bin/bash
# lanes x hardware-queue sweep for the EfficientDet-family configs (hipGraph replay)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O
cd $R
: > $O/lanes_sweep.txt
for q in "" 8 16; do
  for spec in "efficientdet-d1 16" "d1_fcs2_atss 32"; do
    set -- $spec
    for lanes in 1 2 4 8; do
      if [ -n "$q" ]; then export GPU_MAX_HW_QUEUES=$q; else unset GPU_MAX_HW_QUEUES; fi
      timeout -k 10 120 python bench.py --config $1 --batch $2 --lanes $lanes --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs --parity-images 1 2>/dev/null | tail -1 | python -c "import sys,json; o=json.loads(sys.stdin.read()); print('hwq=${q:-default}', '$1', 'lanes', $lanes, o['value'], o['ms_per_step'])" | tee -a $O/lanes_sweep.txt
    done
  done
done
