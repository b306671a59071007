# Idle time inside one replayed pass from a rocprofv3 kernel trace: wall time of a pass (first batched step of pass i -> first of
# pass i+1), union of kernel-busy time, and the largest idle gaps with the kernels on either side.
#   usage: pass_gaps.py <trace dir> [steps per pass = 118]
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
per = int(sys.argv[2]) if len(sys.argv) > 2 else 118
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('sl::', '').replace('void ', '')) for r in rows)
steps = [i for i, e in enumerate(ev) if 'k_chol_step_batched' in e[2]]
passes = [steps[i:i + per] for i in range(0, len(steps) - per + 1, per)]
a, b = passes[-4], passes[-3]
seg = ev[a[0]:b[0]]
wall = (ev[b[0]][0] - ev[a[0]][0]) / 1e3
busy, last, gaps = 0.0, seg[0][0], []
prev = None
excl = collections.defaultdict(lambda: [0, 0.0, 0.0])      # per kernel: launches, time it extends the busy front by, own duration
for st, en, n in seg:
    if st > last and prev is not None:
        gaps.append(((st - last) / 1e3, prev, n))
    s2 = max(st, last)
    excl[n][0] += 1
    excl[n][2] += (en - st) / 1e3
    if en > s2:
        busy += (en - s2) / 1e3
        excl[n][1] += (en - s2) / 1e3
        last = en
        prev = n
print(f"pass wall {wall:.1f} us, kernel-busy (union) {busy:.1f} us, idle {wall - busy:.1f} us in {len(gaps)} gaps, {len(seg)} kernels")
print("kernel                                     launches   front us   own us")
for n, (c, f, o) in sorted(excl.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"{n[:42]:42s} {c:8d} {f:10.1f} {o:8.1f}")
agg = collections.defaultdict(lambda: [0, 0.0])
for g, p, n in gaps:
    agg[(p, n)][0] += 1
    agg[(p, n)][1] += g
for (p, n), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{t:8.1f} us in {c:3d} gaps  after {p[:40]:40s} before {n[:40]}")
