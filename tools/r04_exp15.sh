#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
{
for d in 66 82 90; do
echo "== correctness sched $d"
MYDET_WINO4_VARIANT=0 MYDET_W4_SCHED=$d timeout -k 5 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "winograd4" 2>&1 | tail -2
done
for d in 0 66 82 90 65 67 0 66; do
  echo "== V0 sched $d"
  MYDET_WINO4_VARIANT=0 MYDET_W4_SCHED=$d MYDET_W4_INPUT=1 timeout -k 5 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-215
done
} 2>&1 | grep -v amdgpu.ids > $O/exp15.txt
cat $O/exp15.txt
