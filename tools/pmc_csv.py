#!/usr/bin/env python3
"""Per-kernel means of a rocprofv3 counter_collection.csv:  python tools/pmc_csv.py <file>"""
import collections
import csv
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    agg[r['Kernel_Name'][:90]][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(agg):
    for c, v in agg[k].items():
        print("%-92s %-12s n=%5d mean=%.6g" % (k, c, len(v), sum(v) / len(v)))
