# Summarise a rocprofv3 kernel trace: per-kernel totals, and k_chol_step durations per grid size (= per block column k).
import csv, glob, sys, collections
files = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)
rows = []
for f in files:
    rows += list(csv.DictReader(open(f)))
tot = collections.defaultdict(lambda: [0, 0.0])
steps = collections.defaultdict(list)
for r in rows:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    n = r['Kernel_Name'].split('(')[0]
    tot[n][0] += 1; tot[n][1] += d
    if 'k_chol_step' in n or 'k_chol_a' in n or 'k_chol_b' in n:
        steps[n.split('::')[-1], int(r['Grid_Size_X']) // 256 if 'Grid_Size_X' in r else int(r['Grid_Size']) // 256].append(d)
for n, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"{t:12.1f} us {c:7d} x {t/c:9.2f} us  {n}")
print("k_chol_step: workgroups -> median us")
for g in sorted(steps, reverse=True):
    v = sorted(steps[g]); print(g[0][-1], g[1], round(v[len(v)//2], 1), end=' | ')
print()
