# Per-launch medians of k_chol_step_batched inside the exact joint passes of a rocprofv3 kernel trace (csv), launch sequence by launch
# sequence: the launches of a pass are grouped by the queue they ran on (the bands' segments run in overlapping sequences on the batch's
# streams; the bands' second level and the separator's leaves follow on the pass's own stream) and numbered in start order.
# usage: trace_batched.py <dir> [passes]
import collections
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
n = int(sys.argv[2]) if len(sys.argv) > 2 and int(sys.argv[2]) < 40 else 15
rows.sort(key=lambda r: int(r['Start_Timestamp']))
name = lambda r: r['Kernel_Name'].split('(')[0].replace('void ', '')
dur = lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
tot = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    tot[name(r)][0] += 1
    tot[name(r)][1] += dur(r)
for k, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"{t:12.1f} us {c:7d} x {t / c:9.2f} us  {k}")
marks = [i for i, r in enumerate(rows) if name(r).endswith('k_status_clear')]
gathers = [i for i, r in enumerate(rows) if name(r).endswith('k_sep_gather')]
if len(gathers) < n + 1:
    print("not enough exact joint passes in the trace:", len(gathers))
    sys.exit(0)
starts = [max(m for m in marks if m < g) for g in gathers][-(n + 1):]
qkey = 'Queue_Id' if 'Queue_Id' in rows[0] else ('Stream_Id' if 'Stream_Id' in rows[0] else None)
per = collections.defaultdict(lambda: collections.defaultdict(list))      # sequence -> position -> durations
shape = None
for a, b in zip(starts[:-1], starts[1:]):
    seqs = collections.OrderedDict()
    gather_at = next(i for i in range(a, b) if name(rows[i]).endswith('k_sep_gather'))
    for i in range(a, b):
        r = rows[i]
        if not name(r).endswith('k_chol_step_batched'):
            continue
        q = (r[qkey] if qkey else '0', i > gather_at)      # (behind the gather: the separator's leaves)
        seqs.setdefault(q, []).append(dur(r))
    lens = tuple(len(v) for v in seqs.values())
    shape = shape or lens
    if lens != shape:
        continue
    for s_, v in enumerate(seqs.values()):
        for k, d in enumerate(v):
            per[s_][k].append(d)
print(f"k_chol_step_batched by launch sequence over {n} exact joint passes (median us per position; sequences of one pass: {shape}):")
for s_ in sorted(per):
    med = [sorted(v)[len(v) // 2] for _, v in sorted(per[s_].items())]
    print(f"sequence {s_} ({len(med)} launches, sum {sum(med):.1f} us): " + " | ".join(f"{k} {m:.1f}" for k, m in enumerate(med)))
