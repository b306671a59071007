# Per-kernel summary of the EXACT joint passes from a rocprofv3 kernel trace (csv): the passes are delimited by k_sep_gather (one launch
# per pass); kernels of the last `n` complete passes only (the builds before them launch some of the same kernels on growing systems).
# usage: trace_exact.py <dir> [n passes]
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
marks = [i for i, r in enumerate(rows) if 'k_status_clear' in name(r)]
gathers = [i for i, r in enumerate(rows) if 'k_sep_gather' in name(r)]
if len(gathers) < n + 1:
    print("not enough passes in the trace:", len(gathers)); sys.exit(1)
# a pass starts at the k_status_clear before its gather and ends before the next pass's k_status_clear
starts = [max(m for m in marks if m < g) for g in gathers]
sel = starts[-(n + 1):]
lo, hi = sel[0], sel[-1]
per = collections.defaultdict(lambda: [0, 0.0])
for r in rows[lo:hi]:
    if not name(r).startswith('sl::'):      # (runtime copies of the host code between the passes — pose read-backs — are not part of a pass)
        continue
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    per[name(r)][0] += 1
    per[name(r)][1] += d
wall = (int(rows[hi]['Start_Timestamp']) - int(rows[lo]['Start_Timestamp'])) / 1e3 / n
print(f"exact joint passes: {n} passes, {wall:.1f} us wall per pass (profiler attached)")
print(f"{'kernel':58s} {'launches/pass':>13s} {'avg us':>9s} {'us/pass':>9s}")
for k, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print(f"{k[:58]:58s} {c / n:13.1f} {t / c:9.2f} {t / n:9.1f}")
# by-k medians of the separator's dense steps (k_chol_step, un-batched) inside those passes
steps = collections.defaultdict(list)
for a, b in zip(sel[:-1], sel[1:]):
    ks = [r for r in rows[a:b] if name(r).endswith('k_chol_step')]
    for k, r in enumerate(ks):
        steps[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print("separator k_chol_step by block column (median us):")
print(" | ".join(f"{k} {sorted(v)[len(v) // 2]:.1f}" for k, v in sorted(steps.items())))
# the border products of a pass in launch order (the robots', the bands' second level's, the separator's leaves', lambda block's, top block's)
prods = collections.defaultdict(list)
for a, b in zip(sel[:-1], sel[1:]):
    ks = [r for r in rows[a:b] if name(r).endswith('k_border_syrk')]
    for k, r in enumerate(ks):
        prods[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print("k_border_syrk by position in the pass (median us; grid):")
grid = {}
for r in rows[sel[-2]:sel[-1]]:
    if name(r).endswith('k_border_syrk'):
        grid[len(grid)] = "x".join(str(r.get(k, "?")) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
print(" | ".join(f"{k} {sorted(v)[len(v) // 2]:.1f} ({grid.get(k, '?')})" for k, v in sorted(prods.items())))
