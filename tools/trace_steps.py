# Summarise a rocprofv3 kernel trace: per-kernel totals, and k_chol_step durations per block column k
# (launches in start order, position modulo T).  usage: trace_steps.py <dir> [T]
import csv, glob, sys, collections
files = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)
T = int(sys.argv[2]) if len(sys.argv) > 2 else 59
rows = []
for f in files:
    rows += list(csv.DictReader(open(f)))
tot = collections.defaultdict(lambda: [0, 0.0])
st = []
for r in rows:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    n = r['Kernel_Name'].split('(')[0]
    tot[n][0] += 1; tot[n][1] += d
    if 'k_chol_step' in n:
        st.append((int(r['Start_Timestamp']), d))
for n, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"{t:12.1f} us {c:7d} x {t/c:9.2f} us  {n}")
st.sort()
NIT = int(sys.argv[3]) if len(sys.argv) > 3 else 20
st = st[-NIT * T:]      # the last NIT full-size factorizations (the build phase before them has growing T)
steps = collections.defaultdict(list)
for i, (_, d) in enumerate(st):
    steps[i % T].append(d)
print("k_chol_step: k -> median us")
s = 0.0
for k in sorted(steps):
    v = sorted(steps[k]); m = v[len(v) // 2]; s += m
    print(k, round(m, 1), end=' | ')
print("\nsum of medians", round(s, 1), "us")
