"""Per-dispatch SQ counters of one kernel family from a rocprofv3 --pmc CSV run (--output-format csv).
    python tools/pmc_kernel.py <dir> <kernel-name-substring>"""
import collections
import csv
import glob
import sys

d, sub = sys.argv[1], sys.argv[2]
cc = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
kt = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)
dur = {}
if kt:
    for r in csv.DictReader(open(kt[0])):
        dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000
vals, meta = collections.defaultdict(dict), {}
for r in csv.DictReader(open(cc)):
    if sub in r['Kernel_Name']:
        vals[r['Dispatch_Id']][r['Counter_Name']] = float(r['Counter_Value'])
        meta[r['Dispatch_Id']] = (r['Kernel_Name'].split('(')[0][-60:], r['Grid_Size'], r['VGPR_Count'], r['LDS_Block_Size'])
seen = set()
for k, v in vals.items():
    key = meta[k][:2]
    if key in seen:
        continue
    seen.add(key)
    wc = v.get('SQ_WAVE_CYCLES', 0) or 1
    print(meta[k], f'dur {dur.get(k, 0):.1f} us')
    print('   ' + '  '.join(f'{n}={x:.3g}' + (f' ({x / wc:.2f}/wave-cycle)' if n.startswith('SQ_') and n != 'SQ_WAVE_CYCLES' else '') for n, x in sorted(v.items())))
