import csv, sys, glob, os
f = max(glob.glob(sys.argv[1] + '/*/*kernel_stats.csv'), key=os.path.getmtime)
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 50
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms/step: %.3f" % (tot / 1e6 / steps))
for r in rows[:16]:
    print("%-72s calls=%5s avg_us=%8.2f us/step=%7.1f pct=%5.1f" % (r['Name'][:72], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e3 / steps, float(r['Percentage'])))
