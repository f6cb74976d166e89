#!/bin/bash
# A/B runs of one BASELINE configuration in a tuning build on the GPU box:
#   bash tools/ab_cfg.sh <tag> <C2|C4|C5sd|...> <kernel-name regex> "VAR=val" ...
out=gpurun_out/$1; cfg=$2; pat=$3; shift; shift; shift
mkdir -p $out
make -C pauxy_amd/csrc -j32 TUNING=1 > $out/build.log 2>&1 || { tail -5 $out/build.log; exit 1; }
export AFQ_LIBRARY=$PWD/pauxy_amd/libafqmc_hip_tuning.so      # the product library is left alone
export TMPDIR=/tmp
i=0
for envs in "$@"; do
  i=$((i+1))
  ( export $envs; rocprofv3 --kernel-trace --stats --output-format csv -d $out/p$i -o p -- python3 bench.py --config $cfg --no-cpu-baseline > $out/run$i.json 2> $out/run$i.err )
  f=$(ls $out/p$i/*kernel_stats.csv $out/p$i/*/*kernel_stats.csv 2>/dev/null | head -1)
  echo "[$envs] $(grep -o '"ms_per_step": [0-9.]*' $out/run$i.json)"
  python3 - "$f" "$pat" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r['Name']):
        print("    %-64s calls %4s avg %9.1f us" % (r['Name'][:64], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
