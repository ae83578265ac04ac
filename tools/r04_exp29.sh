#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/st_b1
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_b1 -- python3 $R/bench.py --profile --batch 1 --size 512 --steps 50 --warmup 8 > $O/st_b1.log 2>&1 || { tail -3 $O/st_b1.log; exit 1; }
cp $(find $O/st_b1 -name '*kernel_stats.csv' | head -1) $O/r04_kernel_stats_yolov3_b1_512.csv
rm -rf $O/st_b1
cut -c1-200 $O/r04_kernel_stats_yolov3_b1_512.csv | head -24
