"""Timeline of one replayed step from a rocprofv3 --kernel-trace CSV: how much of the step has 0 / 1 / 2+ kernels in flight,
the gaps, and the longest kernels.   python tools/timeline.py <dir> [steps-from-the-end]"""
import csv
import glob
import sys

d = sys.argv[1]
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
import re


def short(n):
    n = re.sub(r'^void ', '', n)
    n = n.replace('(anonymous namespace)::', '')
    return n.split('(')[0][:52]


rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])) for r in csv.DictReader(open(f))]
rows.sort()
# the last replayed step: everything after the end of the previous step's last post-process launch(es)
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 1
pp = [r for r in rows if r[2].startswith('postprocess_kernel')]
if len(pp) >= 2 * lanes:
    prev_end = max(r[1] for r in pp[-2 * lanes:-lanes])
    rows = [r for r in rows if r[0] >= prev_end]
t0, t1 = rows[0][0], max(r[1] for r in rows)
ev = sorted([(s, 1) for s, e, _ in rows] + [(e, -1) for s, e, _ in rows])
depth, last, hist = 0, t0, {}
for t, dlt in ev:
    hist[min(depth, 3)] = hist.get(min(depth, 3), 0) + (t - last)
    depth += dlt
    last = t
tot = t1 - t0
print(f'{len(rows)} kernels over {tot / 1e3:.1f} us; sum of kernel durations {sum(e - s for s, e, _ in rows) / 1e3:.1f} us')
for k in sorted(hist):
    print(f'  {k}{"+" if k == 3 else ""} kernels in flight: {hist[k] / 1e3:8.1f} us  {100 * hist[k] / tot:5.1f} %')
agg = {}
for s, e, n in rows:
    a = agg.setdefault(n, [0, 0])
    a[0] += 1; a[1] += e - s
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f'  {n:46s} {c:4d} x {t / c / 1e3:7.1f} us = {t / 1e3:8.1f} us')
