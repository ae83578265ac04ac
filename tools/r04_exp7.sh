#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
{ timeout -k 5 120 python tools/r04_bw.py
for s in "64 128 160" "128 256 80" "256 512 40" "512 1024 20"; do set -- $s
 MYDET_W4_DBG=4 timeout -k 5 120 python tools/bench_conv.py --cin $1 --cout $2 --hw $3 --res --wino4
done
} 2>&1 | grep -v amdgpu.ids > $O/exp7.txt
cat $O/exp7.txt
