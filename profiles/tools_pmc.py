import csv, sys, glob, collections, os
d = sys.argv[1]
f = max(glob.glob(d + '/*/*counter_collection.csv'), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in agg.items():
    if not any(x in k for x in ('exx', 'mfma_gemm', 'greens', 'prop_fused', 'gj_big', 'chol_linv', 'reortho')): continue
    print(k)
    for c, v in cs.items():
        print("    %-32s n=%4d mean=%.4g" % (c, len(v), sum(v) / len(v)))
