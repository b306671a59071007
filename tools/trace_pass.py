# timeline of distributed passes with the batched factorisation from a rocprofv3 kernel trace: where the time outside
# k_chol_step_batched goes.  usage: trace_pass.py <dir> [T]
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
T = int(sys.argv[2]) if len(sys.argv) > 2 else 59
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('sl::', '').replace('void ', '')) for r in rows)
steps = [i for i, e in enumerate(ev) if 'k_chol_step_batched' in e[2]]
passes = [steps[i:i + T] for i in range(0, len(steps) - T + 1, T)][-12:-2]
res = collections.defaultdict(list)
for a, b in zip(passes[:-1], passes[1:]):
    s0, e0 = ev[a[0]][0], ev[a[-1]][1]            # batched steps of this pass
    nxt = ev[b[0]][0]                              # first batched step of the next pass
    res['batched steps (first start -> last end)'].append((e0 - s0) / 1e3)
    res['between passes (last step end -> next first step start)'].append((nxt - e0) / 1e3)
    inter = [e for e in ev[a[-1] + 1:b[0]]]
    busy = collections.defaultdict(float)
    for st, en, n in inter: busy[n] += (en - st) / 1e3
    for n, v in busy.items(): res['  sum of kernel time in between: ' + n].append(v)
    # union busy time in between
    t, last = 0, e0
    for st, en, n in sorted(inter):
        st = max(st, last)
        if en > st: t += en - st; last = en
    res['  union of kernel-busy time in between'].append(t / 1e3)
for k, v in res.items():
    v = sorted(v); print(f"{v[len(v)//2]:10.1f} us  {k}")
