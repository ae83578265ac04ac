#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
line() { tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('parity_check',{}).get('ok'))"; }
{   echo "== d1 batch 8 lanes 1: $(timeout -k 5 300 python bench.py --config efficientdet-d1 --batch 8 --lanes 1 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | line)"
    echo "== d1 batch 16 lanes 2: $(timeout -k 5 300 python bench.py --config efficientdet-d1 --batch 16 --lanes 2 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | line)"
    echo "== d1 batch 16 lanes 1: $(timeout -k 5 300 python bench.py --config efficientdet-d1 --batch 16 --lanes 1 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | line)"
    echo "== d1 batch 4 lanes 1: $(timeout -k 5 300 python bench.py --config efficientdet-d1 --batch 4 --lanes 1 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | line)"
    echo "== d1 batch 16 lanes 4: $(timeout -k 5 300 python bench.py --config efficientdet-d1 --batch 16 --lanes 4 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | line)"
    echo "== d1 batch 1 lanes 1: $(timeout -k 5 300 python bench.py --config efficientdet-d1 --batch 1 --lanes 1 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | line)"
} 2>&1 | grep -v amdgpu.ids > $O/lanes.txt
cat $O/lanes.txt
