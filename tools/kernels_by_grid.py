import csv, glob, sys, collections
rows=[]
for f in glob.glob(sys.argv[1]+'/**/*kernel_trace.csv', recursive=True): rows+=list(csv.DictReader(open(f)))
per=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name']
    if 'k_border_syrk' in n or 'k_border_apply' in n or 'k_chol_bwd_chain' in n:
        key=(n.split('(')[0], r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'])
        per[key].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(per.items()):
    v=sorted(v); print(k, len(v), 'median', v[len(v)//2])
