#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
{
for d in 0 64 65 66 81 82 90 91 92 0; do
  echo "== V0 dbg $d"
  MYDET_WINO4_VARIANT=0 MYDET_W4_DBG=$d MYDET_W4_INPUT=1 timeout -k 5 300 python tools/profile_layers.py | grep -E "wino4|total"
done
} 2>&1 | grep -v amdgpu.ids > $O/exp14.txt
cat $O/exp14.txt
