#!/bin/bash
# Per-dispatch durations of EVERY kernel of one bench.py run, grouped by (kernel, grid size): median, count, share.
#   trace_all.sh <out name> <bench args...>
T=${MYDET_TOOL_TIMEOUT:-300}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; NAME=$1; shift
D=$R/gpurun_out/trace_$NAME; rm -rf $D; mkdir -p $D
timeout -k 5 $T rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline "$@" > $D/log.txt 2>&1 || { tail -5 $D/log.txt; exit 1; }
python3 - > $R/gpurun_out/trace_$NAME.txt <<PY
import csv, glob, collections
f = glob.glob('$D/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
# keep the last 3 steps' worth: drop dispatches before the last third of postprocess launches is not needed -- report per launch medians
d = collections.defaultdict(list)
for r in rows:
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0]
    name = name.split('<')[0][-40:] + ('<' + r['Kernel_Name'].split('<', 1)[1][:40] if '<' in r['Kernel_Name'] else '')
    wg = int(r['Grid_Size_X']) * max(int(r.get('Grid_Size_Y', 1) or 1), 1) // max(int(r['Workgroup_Size_X']), 1)
    d[(name, wg)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000)
tot = sum(sum(v) for v in d.values())
print(f'{"kernel":82s} {"WGs":>8s} {"n":>5s} {"median us":>10s} {"sum ms":>8s} {"%":>5s}')
for (k, wg), v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v = sorted(v)
    print(f'{k:82s} {wg:8d} {len(v):5d} {v[len(v)//2]:10.1f} {sum(v)/1000:8.3f} {100*sum(v)/tot:5.1f}')
PY
rm -rf $D
head -50 $R/gpurun_out/trace_$NAME.txt
