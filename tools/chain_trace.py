"""The dependent launch chain of ONE replayed step from a rocprofv3 --kernel-trace CSV, in execution order: start offset,
duration and the idle gap before every kernel, plus totals (sum of durations, sum of gaps) -- what a lane of the EfficientDet
family spends in kernels and what it spends between them.   python tools/chain_trace.py <dir> [lanes]"""
import csv
import glob
import re
import sys

d = sys.argv[1]
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 1
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]


def short(n):
    n = re.sub(r'^void ', '', n).replace('(anonymous namespace)::', '')
    return n.split('(')[0][:60]


rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']),
               int(r['Grid_Size_X']) * max(int(r.get('Grid_Size_Y', 1) or 1), 1) // max(int(r['Workgroup_Size_X']), 1))
              for r in csv.DictReader(open(f)))
pp = [r for r in rows if r[2].startswith('postprocess_kernel')]
prev_end = max(r[1] for r in pp[-2 * lanes:-lanes])
rows = [r for r in rows if r[0] >= prev_end]
t0 = rows[0][0]
end = t0
gaps = durs = 0
print(f'{"t us":>9s} {"gap us":>7s} {"dur us":>8s} {"WGs":>7s}  kernel')
for s, e, n, wg in rows:
    gap = (s - end) / 1e3
    print(f'{(s - t0) / 1e3:9.1f} {gap:7.1f} {(e - s) / 1e3:8.1f} {wg:7d}  {n}')
    if lanes == 1:
        gaps += max(gap, 0.0)
    durs += (e - s) / 1e3
    end = max(end, e)
print(f'{len(rows)} kernels, wall {(end - t0) / 1e3:.1f} us, sum of durations {durs:.1f} us' + (f', sum of gaps {gaps:.1f} us' if lanes == 1 else ''))
