#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
{
for cfg in "0 5" "8 5" "4 5" "2 5" "4 2" "4 3" "4 8" "8 8" "0 5"; do set -- $cfg
  echo "== EARLY_CHUNK=$1 EARLY_LAYERS=$2"
  MYDET_EARLY_CHUNK=$1 MYDET_EARLY_LAYERS=$2 MYDET_W4B_MIN_CIN=999 MYDET_W4_INPUT=1 timeout -k 5 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-215
done
} 2>&1 | grep -v amdgpu.ids > $O/exp12.txt
cat $O/exp12.txt
