"""Build-time guard for the gfx950 store-data hazard met in round 2 (profiles/r03_isa_notes.md).

A `buffer_store_dwordx3/x4` whose scalar offset is an SGPR, followed directly by a VALU write of one of its data
registers, stored part of the NEXT value on MI355X about once in eight launches.  LLVM's hazard recogniser pads this
pattern with a wait state only when the store has NO SGPR soffset (GCNHazardRecognizer::createsVALUHazard), so the code
base keeps per-store offsets in the voffset -- and this script proves it on the shipped library: it pulls every gfx950
code object out of libmydet_hip.so, disassembles it and fails if any 12/16-byte buffer store with an SGPR soffset is
immediately followed (no intervening instruction) by a VALU instruction that writes one of the store's data VGPRs.

    python tools/check_store_hazard.py [path/to/libmydet_hip.so]      exit code 1 and a listing when the pattern exists
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = '/opt/rocm/lib/llvm/bin'
MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'
STORE = re.compile(r'\bbuffer_store_dwordx[34]\s+v\[(\d+):(\d+)\],\s*(\S+),\s*s\[\d+:\d+\],\s*(\S+)')
VDST = re.compile(r'^\s*(v_\w+)\s+(v\[(\d+):(\d+)\]|v(\d+))')


def code_objects(lib, notes=False):
    """Disassembly text of every gfx950 code object bundled in `lib` (one bundle per translation unit)."""
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, 'fat.bin')
        subprocess.check_call([f'{LLVM}/llvm-objcopy', '--dump-section', f'.hip_fatbin={fat}', lib, os.path.join(tmp, 'copy.so')])
        blob = open(fat, 'rb').read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        for i, s in enumerate(starts):
            part = os.path.join(tmp, f'bundle{i}.bin')
            open(part, 'wb').write(blob[s:starts[i + 1] if i + 1 < len(starts) else len(blob)])
            co = os.path.join(tmp, f'dev{i}.co')
            r = subprocess.run([f'{LLVM}/clang-offload-bundler', '--type=o', '--unbundle', f'--input={part}',
                                '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', f'--output={co}'], capture_output=True)
            if r.returncode or not os.path.exists(co) or os.path.getsize(co) == 0:
                continue
            if notes:
                yield i, subprocess.run([f'{LLVM}/llvm-readelf', '--notes', co], capture_output=True, text=True, check=True).stdout
            else:
                yield i, subprocess.run([f'{LLVM}/llvm-objdump', '-d', co], capture_output=True, text=True, check=True).stdout


def scratch_users(lib):
    """{kernel name: bytes of scratch per lane} for every kernel of `lib` whose code object metadata asks for a private
    segment (register spills or a dynamically indexed local array): such a kernel runs its spills through memory, and round 5
    found a fixup kernel at 3 KB per lane (89 us for a 32 MB sum)."""
    out = {}
    for _, text in code_objects(lib, notes=True):
        name = None
        for line in text.splitlines():
            m = re.match(r'\s*\.name:\s+(\S+)', line)
            if m:
                name = m.group(1)
            m = re.match(r'\s*\.private_segment_fixed_size:\s+(\d+)', line)
            if m and name and int(m.group(1)) > 0:
                out[name] = int(m.group(1))
    return out


PACKED_F32 = re.compile(r'^v_pk_(fma|mul|add)_f32\b')


def packed_f32_users(lib):
    """{kernel name: count} of v_pk_fma/mul/add_f32 instructions in `lib`.  Round 5/6 finding (profiles/r06_pk_fma_finding.md):
    on MI355X a v_pk_fma_f32 of one wave returned wrong low halves in lanes 48-63 while ANOTHER wave of the SIMD issued dense
    bf16 MFMAs; every kernel of this library can run beside the split-bf16 convs (two batch lanes), so the library is built
    with -packed-fp32-ops off and must contain none."""
    out = {}
    for _, text in code_objects(lib):
        for fn, ins in instructions(text):
            if PACKED_F32.match(ins):
                out[fn] = out.get(fn, 0) + 1
    return out


def instructions(text):
    """(function, mnemonic line) for every instruction line of an llvm-objdump listing."""
    fn = '?'
    for line in text.splitlines():
        m = re.match(r'^[0-9a-f]+ <(.+)>:$', line)
        if m:
            fn = m.group(1)
            continue
        if '\t' in line and not line.startswith('Disassembly'):
            ins = line.split('//')[0].strip()
            if ins and not ins.endswith(':'):
                yield fn, ins


def violations(text):
    prev = None
    stores = sgpr_stores = 0
    bad = []
    for fn, ins in instructions(text):
        if prev is not None:
            m = VDST.match(ins)
            if m:
                lo, hi = (int(m.group(3)), int(m.group(4))) if m.group(3) else (int(m.group(5)), int(m.group(5)))
                if lo <= prev[2] and hi >= prev[1]:
                    bad.append((prev[0], prev[3], ins))
        prev = None
        s = STORE.search(ins)
        if s:
            stores += 1
            if re.fullmatch(r's\d+|m0|ttmp\d+', s.group(4)):
                sgpr_stores += 1
                prev = (fn, int(s.group(1)), int(s.group(2)), ins)
    return stores, sgpr_stores, bad


def main(lib):
    tot = sgpr = 0
    bad = []
    n = 0
    for _, text in code_objects(lib):
        n += 1
        a, b, c = violations(text)
        tot += a
        sgpr += b
        bad += c
    print(f'{n} gfx950 code objects, {tot} 12/16-byte buffer stores, {sgpr} of them with an SGPR soffset, {len(bad)} followed '
          'directly by a VALU write of their data registers')
    for fn, st, nx in bad[:20]:
        print(f'  {fn[:70]}\n      {st}\n      {nx}')
    pk = packed_f32_users(lib)
    print(f'{sum(pk.values())} packed-f32 VALU instructions (v_pk_fma/mul/add_f32) in {len(pk)} kernels')
    for fn, c in sorted(pk.items(), key=lambda kv: -kv[1])[:20]:
        print(f'  {c:5d}  {fn[:100]}')
    return 1 if bad or pk else 0


if __name__ == '__main__':
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.exit(main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, 'mydetection_amd', 'lib', 'libmydet_hip.so')))
