#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
{ timeout -k 5 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "upcat or igemm" 2>&1 | tail -4
for f in 0 1 0 1; do
  echo "== MYDET_FUSED_UPCAT=$f"
  MYDET_FUSED_UPCAT=$f timeout -k 5 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-215
done
MYDET_FUSED_UPCAT=1 timeout -k 5 300 python tools/profile_layers.py | grep -E "\^|upsample|384->|768->|total"
} 2>&1 | grep -v amdgpu.ids > $O/upcat.txt
cat $O/upcat.txt
