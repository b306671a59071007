# Sums rocprofv3 --pmc counter rows per kernel name: usage  pmc_summary.py <dir> <COUNTER>
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
cnt = sys.argv[2]
tot = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    if r['Counter_Name'] != cnt:
        continue
    n = r['Kernel_Name'].split('(')[0]
    tot[n][0] += 1
    tot[n][1] += float(r['Counter_Value'])
for n, (c, v) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"{cnt} {n[:50]:50s} dispatches {c:6d}  sum {v:16.1f}  per dispatch {v / c:14.2f}")
