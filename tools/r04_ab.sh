#!/bin/bash
# A/B of library builds in one call: tools/r04_ab.sh <lib-a> <lib-b> ...   (paths under mydetection_amd/lib/ab/)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
line() { tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('parity_check',{}).get('ok'))"; }
{ for rep in 1 2; do for l in "$@"; do
    export MYDET_LIB_PATH=$R/mydetection_amd/lib/ab/$l.so
    echo "== $l b1_512: $(timeout -k 5 300 python bench.py --batch 1 --size 512 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | line)"
    echo "== $l d1: $(timeout -k 5 300 python bench.py --config efficientdet-d1 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | line)"
    echo "== $l fcos: $(timeout -k 5 300 python bench.py --config d1_fcs2_atss --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line)"
    echo "== $l yolov3: $(timeout -k 5 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line)"
done; done; } 2>&1 | grep -v amdgpu.ids > $O/ab.txt
cat $O/ab.txt
