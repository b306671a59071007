# per-k median duration of k_chol_step_batched from a rocprofv3 kernel trace (csv): usage trace_batched.py <dir> [T]
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
T = int(sys.argv[2]) if len(sys.argv) > 2 else 59
tot = collections.defaultdict(lambda: [0, 0.0])
st = []
for r in rows:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    n = r['Kernel_Name'].split('(')[0]
    tot[n][0] += 1; tot[n][1] += d
    if 'k_chol_step_batched' in n: st.append((int(r['Start_Timestamp']), d))
for n, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"{t:12.1f} us {c:7d} x {t/c:9.2f} us  {n}")
st.sort(); st = st[-20 * T:]
steps = collections.defaultdict(list)
for i, (_, d) in enumerate(st): steps[i % T].append(d)
s = 0
for k in sorted(steps):
    v = sorted(steps[k]); m = v[len(v)//2]; s += m
    print(k, round(m, 1), end=' | ')
print("\nsum", round(s, 1))
