"""Sum rocprofv3 --pmc counter_collection.csv per kernel name and counter (averaged per dispatch).
    python tools/pmc_summary.py <dir> [kernel-substring]"""
import csv
import glob
import sys
from collections import defaultdict

d, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else '')
acc, cnt = defaultdict(float), defaultdict(set)
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if sub not in k:
            continue
        key = (k[:60], r['Counter_Name'])
        acc[key] += float(r['Counter_Value'])
        cnt[key].add(r['Dispatch_Id'])
for (k, c), v in sorted(acc.items()):
    print(f'{k:60s} {c:32s} {v / len(cnt[(k, c)]):16.1f} per dispatch ({len(cnt[(k, c)])} dispatches)')
