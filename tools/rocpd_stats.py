"""Per-kernel statistics from a rocprofv3 rocpd database (ROCm 7.2 writes <name>_results.db by default):
the same columns as the --stats CSV.   python tools/rocpd_stats.py <results.db> [--grid] > profiles/<name>.csv
--grid adds one line per (kernel, grid size) so layers of one kernel template can be told apart."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
by_grid = '--grid' in sys.argv
rows = db.execute('select name, duration, grid_x, grid_y, grid_z, workgroup_x from kernels').fetchall()
agg = {}
for name, dur, gx, gy, gz, wx in rows:
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\((ConvArgs|DwArgs|FuseArgs|PPArgs|DecodeArgs|StemArgs|WinoArgs)[^)]*\)$', '', name)
    key = (name, f'{gx // max(wx, 1)}x{gy}x{gz}') if by_grid else (name,)
    agg.setdefault(key, []).append(dur)
tot = sum(sum(v) for v in agg.values())
print('"Name",' + ('"Grid",' if by_grid else '') + '"Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
for key, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    cells = [f'"{key[0]}"'] + ([f'"{key[1]}"'] if by_grid else [])
    print(','.join(cells + [str(len(v)), str(sum(v)), f'{sum(v) / len(v):.1f}', f'{100 * sum(v) / tot:.2f}', str(min(v)), str(max(v))]))
