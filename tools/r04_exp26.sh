#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/tl_d1
timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $O/tl_d1 -- python3 $R/bench.py --profile --config efficientdet-d1 --steps 3 --warmup 1 > $O/tl_d1.log 2>&1 || { tail -3 $O/tl_d1.log; exit 1; }
python3 $R/tools/timeline.py $O/tl_d1 2 > $O/timeline_d1.txt; cat $O/timeline_d1.txt
rm -rf $O/tl_d1
timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $O/tl_d1 -- python3 $R/bench.py --profile --config efficientdet-d1 --steps 3 --warmup 1 > $O/tl_d1.log 2>&1 || { tail -3 $O/tl_d1.log; exit 1; }
python3 $R/tools/timeline.py $O/tl_d1 > $O/timeline_d1.txt; cat $O/timeline_d1.txt
N=$(python3 - <<PY
import csv,glob
f=glob.glob('$O/tl_d1/**/*kernel_trace.csv', recursive=True)[0]
n=sum(1 for _ in csv.DictReader(open(f)))
print(n//6)
PY
)
echo "launches per pass: $N"
python3 $R/tools/timeline.py $O/tl_d1 $N | tee -a $O/timeline_d1.txt
rm -rf $O/tl_d1
