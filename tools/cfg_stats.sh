#!/bin/bash
# rocprofv3 kernel statistics of one bench.py --config run:  bash tools/cfg_stats.sh <tag> <config> [extra bench args]
out=gpurun_out/$1; cfg=$2; shift; shift
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$cfg -o $cfg -- python3 bench.py --config $cfg --no-cpu-baseline "$@" > $out/$cfg.json 2> $out/$cfg.err
f=$(ls $out/prof_$cfg/*kernel_stats.csv $out/prof_$cfg/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp $f $out/${cfg}_kernel_stats.csv && python3 - $out/${cfg}_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-70s calls %6s avg %9.2f us  %5.1f %%" % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3, float(r['Percentage'])))
PY
