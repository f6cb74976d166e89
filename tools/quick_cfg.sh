#!/bin/bash
# Kernel statistics of one BASELINE configuration (rocprofv3 --kernel-trace --stats): bash tools/quick_cfg.sh <tag> <C1|C2|C4|C5sd|C5>
tag=${1:-q}; cfg=${2:-C2}
out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/p -o p -- python3 bench.py --config $cfg --no-cpu-baseline > $out/cfg_$cfg.json 2> $out/cfg_$cfg.err
f=$(ls $out/p/*kernel_stats.csv $out/p/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp $f $out/cfg_${cfg}_kernel_stats.csv && rm -rf $out/p
python3 - $out/cfg_${cfg}_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print("%-90s %6s %10.2f" % (r['Name'][:90], r['Calls'], float(r['AverageNs']) / 1e3))
PY
tail -c 300 $out/cfg_$cfg.json
