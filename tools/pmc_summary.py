import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if 'gemm_kernel' not in r['Kernel_Name']: continue
    k = r['Kernel_Name'].split('gemm_kernel')[1][:30]
    agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} n={len(v)} mean={sum(v)/len(v):.4e}")
