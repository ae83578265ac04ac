#!/bin/bash
# per-layer HIP-event times of the YOLOv3 headline with the split-bf16 direct convs on and off, alternating, in ONE call
O=gpurun_out/b3_layers; mkdir -p $O
for rep in 1 2; do
  for s in 1 0; do
    MYDET_CONV_SPLIT_BF16=$s timeout -k 5 200 python tools/profile_layers.py > $O/layers_split${s}_rep${rep}.txt 2>/dev/null || exit 1
    echo "split=$s rep=$rep $(tail -1 $O/layers_split${s}_rep${rep}.txt)"
  done
done
