#!/usr/bin/env python3
"""Per-kernel means of the counters in a rocprofv3 (rocpd sqlite) results database.
   python tools/pmc_db.py <dir or .db> [kernel-substring ...]"""
import collections
import glob
import os
import sqlite3
import sys


def main():
    path = sys.argv[1]
    if os.path.isdir(path):
        path = max(glob.glob(os.path.join(path, '**', '*.db'), recursive=True), key=os.path.getmtime)
    want = sys.argv[2:] or ['prop_fused', 'exx', 'mfma_gemm', 'greens', 'gj_big', 'vbias', 'vhs_ueg', 'energy_ueg']
    db = sqlite3.connect(path)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for name, cname, val, d in db.execute("select kernel_name, counter_name, value, duration from counters_collection"):
        agg[name[:70]][cname].append(val)
        dur[name[:70]].append(d)
    for k in sorted(agg):
        if not any(x in k for x in want):
            continue
        print("%s   (avg duration %.1f us while profiled)" % (k, sum(dur[k]) / len(dur[k]) / 1e3))
        for c, v in sorted(agg[k].items()):
            print("    %-28s n=%4d mean=%.5g" % (c, len(v), sum(v) / len(v)))


if __name__ == "__main__":
    main()
