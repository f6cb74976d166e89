#!/bin/bash
# HS-potential GEMM variants of the tuning build with the in-kernel loop stamps (AFQ_GEMM_TS): bash tools/vhs_variants.sh <tag> "VAR=val ..." ...
out=gpurun_out/$1; shift
mkdir -p $out
export AFQ_LIBRARY=$PWD/pauxy_amd/libafqmc_hip_tuning.so
export TMPDIR=/tmp
i=0
for envs in "$@"; do
  i=$((i+1))
  ( export $envs AFQ_GEMM_TS=1; rocprofv3 --kernel-trace --stats --output-format csv -d $out/p$i -o p -- python3 bench.py --steps 40 --warmup 10 --repeats 1 --no-cpu-baseline > $out/run$i.json 2> $out/run$i.err )
  f=$(ls $out/p$i/*kernel_stats.csv $out/p$i/*/*kernel_stats.csv 2>/dev/null | head -1)
  echo "[$envs] $(grep -o '"ms_per_step": [0-9.]*' $out/run$i.json | head -1)"
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'VhsProb' in r['Name']:
        print("    %-66s calls %4s avg %9.1f us" % (r['Name'][:66], r['Calls'], float(r['AverageNs']) / 1e3))
PY
  grep "GEMM_TS after VHS wg  0" $out/run$i.err | head -1
  grep -i "error\|Traceback" $out/run$i.err | head -3
  rm -rf $out/p$i
done
