import csv, sys, glob, collections
d = sys.argv[1]
f = glob.glob(d + '/*/*counter_collection.csv')[0]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in agg.items():
    if not any(x in k for x in ('exx', 'mfma_gemm', 'greens')): continue
    print(k)
    for c, v in cs.items():
        print("    %-32s n=%4d mean=%.4g" % (c, len(v), sum(v) / len(v)))
