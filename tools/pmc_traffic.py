"""Fabric traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass).

    python tools/pmc_traffic.py <fetch_dir> <write_dir> > profiles/rNN_pmc_traffic_b32_640.json

Counters are in KB; FETCH_SIZE is doubled for this code's 16-byte-per-lane loads as MI355X_MICROARCH.md (HBM /
rocprofv3 section) prescribes for gfx950, WRITE_SIZE is exact.  Kernels are grouped into the families bench.py names.
"""
import csv
import glob
import json
import sys
from collections import defaultdict

FAMILIES = [('conv_wino_kernel', 'conv_wino'), ('conv_igemm_kernel', 'conv_igemm'), ('conv_fixup', 'conv_igemm_fixup'),
            ('conv_stem', 'conv_stem'), ('upsample_concat', 'upsample_concat'), ('decode_kernel', 'decode'),
            ('postprocess', 'postprocess'), ('wino_weights', None)]


def family(kernel):
    for sub, fam in FAMILIES:
        if sub in kernel:
            return fam
    return None


def collect(d, counter):
    tot, ids = defaultdict(float), defaultdict(set)
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            fam = family(r['Kernel_Name'])
            if fam and r['Counter_Name'] == counter:
                tot[fam] += float(r['Counter_Value'])
                ids[fam].add(r['Dispatch_Id'])
    return {k: (tot[k], len(ids[k])) for k in tot}


fetch, write = collect(sys.argv[1], 'FETCH_SIZE'), collect(sys.argv[2], 'WRITE_SIZE')
out = {}
for fam in fetch:
    f, n = fetch[fam]
    w, nw = write.get(fam, (0.0, n))
    out[fam] = {'launches': n, 'FETCH_SIZE_KB_per_launch': f / n, 'WRITE_SIZE_KB_per_launch': w / max(nw, 1),
                'hbm_read_bytes_per_launch_x2corr': 2.0 * 1024.0 * f / n, 'hbm_write_bytes_per_launch': 1024.0 * w / max(nw, 1)}
print(json.dumps(out, indent=1))
