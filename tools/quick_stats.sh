#!/bin/bash
# Kernel statistics of one bench.py run (rocprofv3 --kernel-trace --stats): bash tools/quick_stats.sh <tag> [bench args]
tag=${1:-q}; shift
out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/p -o p -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline "$@" > $out/bench.json 2> $out/bench.err
f=$(ls $out/p/*kernel_stats.csv $out/p/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp $f $out/kernel_stats.csv && rm -rf $out/p
python3 - $out/kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-90s %6s %10.2f" % (r['Name'][:90], r['Calls'], float(r['AverageNs']) / 1e3))
PY
tail -c 400 $out/bench.json
