"""Fabric traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass).

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <fetch_dir> -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d <write_dir> -- python3 bench.py ...
    python tools/pmc_traffic.py <fetch_dir> <write_dir> > profiles/rNN_pmc_traffic_<config>_b<B>_<size>.json

Counters are in KB.  MI355X_MICROARCH.md (HBM / rocprofv3): on gfx950 FETCH_SIZE reports half the bytes of a wide
coalesced streaming read (16 B per lane), so it is doubled for the kernel families whose global loads are 16 B per lane;
families that load dwords (the F(2x2) Winograd kernel's activation patches, raw_buffer_load_b32) are outside the calibrated
case and are reported with both factors.  WRITE_SIZE is exact for 16-byte-per-lane stores.
Kernels are grouped into the families bench.py names.
"""
import csv
import glob
import json
import sys
from collections import defaultdict

# (substring of the kernel name, family, FETCH_SIZE correction or None = uncalibrated: report x1 and x2)
FAMILIES = [('conv_wino4_kernel', 'conv_wino4_gemm', 2.0), ('wino4_input_kernel', 'wino4_input', 2.0), ('wino4_fixup', 'wino4_fixup', 2.0),
            ('wino4_weights', None, None),
            ('conv_wino_kernel', 'conv_wino', None), ('conv_igemm_b3_kernel', 'conv_igemm_b3', 2.0), ('conv_p3_kernel', 'conv_p3', 2.0), ('split_bf16', None, None),
            ('conv_igemm_kernel', 'conv_igemm', 2.0), ('conv_fixup', 'conv_igemm_fixup', 2.0),
            ('conv_stem', 'conv_stem', None), ('stem_dw', 'stem_dw', None), ('upsample_concat', 'upsample_concat', 2.0), ('decode_kernel', 'decode', 2.0),
            ('postprocess', 'postprocess', None), ('dwconv', 'dwconv', 2.0), ('sepconv_decode', 'sepconv_decode', 2.0),
            ('sepconv', 'sepconv_nodes', 2.0), ('pw_skinny', 'conv_igemm', 2.0), ('se_fused', 'se_gate', None),
            ('mbconv_expand_dw', 'mbconv_expand_dw', 2.0), ('se_gate', 'se_gate', None), ('se_mean', 'se_gate', None),
            ('bifpn_fuse', 'bifpn_fuse', 2.0), ('maxpool', 'maxpool', 2.0), ('wino_weights', None, None)]


def family(kernel):
    for sub, fam, corr in FAMILIES:
        if sub in kernel:
            return fam, corr
    return None, None


def collect(d, counter):
    tot, ids, corr = defaultdict(float), defaultdict(set), {}
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            fam, c = family(r['Kernel_Name'])
            if fam and r['Counter_Name'] == counter:
                tot[fam] += float(r['Counter_Value'])
                ids[fam].add(r['Dispatch_Id'])
                corr[fam] = c
    return {k: (tot[k], len(ids[k]), corr[k]) for k in tot}


fetch, write = collect(sys.argv[1], 'FETCH_SIZE'), collect(sys.argv[2], 'WRITE_SIZE')
out = {}
for fam in fetch:
    f, n, corr = fetch[fam]
    w, nw, _ = write.get(fam, (0.0, n, None))
    rd, wr = 1024.0 * f / n, 1024.0 * w / max(nw, 1)
    e = {'launches': n, 'FETCH_SIZE_KB_per_launch': f / n, 'WRITE_SIZE_KB_per_launch': w / max(nw, 1), 'hbm_write_bytes_per_launch': wr}
    if corr is None:
        e.update({'fetch_correction': 'uncalibrated (loads are not 16 B per lane): x1 .. x2',
                  'hbm_read_bytes_per_launch_x1': rd, 'hbm_read_bytes_per_launch_x2': 2.0 * rd,
                  'hbm_bytes_per_launch': rd + wr, 'hbm_bytes_per_launch_upper': 2.0 * rd + wr})
    else:
        e.update({'fetch_correction': corr, 'hbm_read_bytes_per_launch': corr * rd, 'hbm_bytes_per_launch': corr * rd + wr})
    out[fam] = e
# bench.py times the F(4x4) Winograd layer as one unit: the input-transform launch, the GEMM launch(es) -- whole items and, where
# the K-cut tail applies, its pieces -- and the tail's fixup launch.  The same unit here: all their bytes over the number of
# layers (= input-transform launches).  All of these kernels load 16 bytes per lane (x2 correction).
if 'conv_wino4_gemm' in out and 'wino4_input' in out:
    layers = out['wino4_input']['launches']
    parts = [out[k] for k in ('conv_wino4_gemm', 'wino4_input', 'wino4_fixup') if k in out]
    rd = sum(e['hbm_read_bytes_per_launch'] * e['launches'] for e in parts) / layers
    wr = sum(e['hbm_write_bytes_per_launch'] * e['launches'] for e in parts) / layers
    out['conv_wino4'] = {'launches': layers, 'unit': 'per layer: wino4_input_kernel + conv_wino4_kernel (whole items + K pieces) + wino4_fixup_kernel',
                         'hbm_read_bytes_per_launch': rd, 'hbm_write_bytes_per_launch': wr, 'hbm_bytes_per_launch': rd + wr}
print(json.dumps(out, indent=1))
