# Launches of the exact joint passes of a rocprofv3 kernel trace (csv) told apart by kernel name AND grid: several launches of one name
# (the border products and substitutions of the different levels) otherwise average into one line.  usage: kernels_by_grid.py <dir> [passes]
import collections
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 15
rows.sort(key=lambda r: int(r['Start_Timestamp']))
name = lambda r: r['Kernel_Name'].split('(')[0].replace('void ', '')
marks = [i for i, r in enumerate(rows) if name(r).endswith('k_status_clear')]
gathers = [i for i, r in enumerate(rows) if name(r).endswith('k_sep_gather')]
starts = [max(m for m in marks if m < g) for g in gathers][-(n + 1):]
per = collections.defaultdict(list)
for r in rows[starts[0]:starts[-1]]:
    if not name(r).startswith('sl::'):
        continue
    key = (name(r), int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1), int(r['Grid_Size_Y']), int(r['Grid_Size_Z']))
    per[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
np_ = len(starts) - 1
print(f"{'kernel':40s} {'workgroups (x, y, z)':>24s} {'launches/pass':>13s} {'median us':>10s} {'us/pass':>9s}")
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    v = sorted(v)
    print(f"{k[0][:40]:40s} {str(k[1:]):>24s} {len(v) / np_:13.1f} {v[len(v) // 2]:10.2f} {sum(v) / np_:9.1f}")
