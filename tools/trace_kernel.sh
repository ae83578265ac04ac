#!/bin/bash
# Durations of every dispatch of kernels matching $1 in one bench.py run, grouped by grid size.
#   trace_kernel.sh <substring> <bench args...>
# every GPU command runs under `timeout -k 5`: an abort or a stuck process cannot hold the GPU lease for minutes
T=${MYDET_TOOL_TIMEOUT:-300}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; SUB=$1; shift
rm -rf $R/gpurun_out/trace; mkdir -p $R/gpurun_out/trace
timeout -k 5 $T rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline "$@" > $R/gpurun_out/trace/log.txt 2>&1 || { tail -5 $R/gpurun_out/trace/log.txt; exit 1; }
python3 - "$SUB" <<PY
import csv, glob, collections, sys
sub = sys.argv[1]
f = glob.glob('$R/gpurun_out/trace/**/*kernel_trace.csv', recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if sub in r['Kernel_Name']:
        d[int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000)
tot = sum(sum(v) for v in d.values())
for k, v in sorted(d.items()):
    v = sorted(v)
    print(f'workgroups {k:6d}  n={len(v):4d}  median {v[len(v)//2]:8.1f} us  sum {sum(v)/1000:7.3f} ms ({100*sum(v)/tot:4.1f} %)')
PY
