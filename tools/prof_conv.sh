#!/bin/bash
# Per-kernel durations of one bench_conv.py configuration (rocprofv3 kernel trace).  usage: prof_conv.sh <bench_conv args>
# every GPU command runs under `timeout -k 5`: an abort or a stuck process cannot hold the GPU lease for minutes
T=${MYDET_TOOL_TIMEOUT:-300}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_conv; mkdir -p $R/gpurun_out/prof_conv
timeout -k 5 $T rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_conv -- python3 $R/tools/bench_conv.py "$@" > $R/gpurun_out/prof_conv/log.txt 2>&1 || { tail -5 $R/gpurun_out/prof_conv/log.txt; exit 1; }
python3 - <<PY
import csv, glob, collections
f = glob.glob('$R/gpurun_out/prof_conv/**/*kernel_trace.csv', recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][-70:]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v = sorted(v)
    print(f'{k:70s} n={len(v):4d} median={v[len(v)//2]:9.1f} us min={v[0]:9.1f}')
PY
