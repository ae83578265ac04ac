#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
{
for v in "" 2 3 4 ""; do
  echo "== MYDET_W4_PART=${v:-off}"
  for s in "64 128 160" "128 256 80" "256 512 40" "512 1024 20"; do set -- $s
    if [ -z "$v" ]; then timeout -k 5 120 python tools/bench_conv.py --cin $1 --cout $2 --hw $3 --res --wino4 || exit 1
    else MYDET_W4_PART=$v timeout -k 5 120 python tools/bench_conv.py --cin $1 --cout $2 --hw $3 --res --wino4 || exit 1; fi
  done
done
} 2>&1 | grep -v amdgpu.ids > $O/exp27.txt
cat $O/exp27.txt
