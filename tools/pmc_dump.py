#!/usr/bin/env python3
"""Per-kernel means of the counters of a rocprofv3 --pmc run: python tools/pmc_dump.py <dir> [name filters...]"""
import collections
import csv
import glob
import os
import sys

f = max(glob.glob(sys.argv[1] + '/*/*counter_collection.csv'), key=os.path.getmtime)
filters = sys.argv[2:] or ['']
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k in agg:
    if any(x in k for x in filters):
        print(k)
        for c, v in sorted(agg[k].items()):
            print("   %-34s %.4g" % (c, sum(v) / len(v)))
