#!/bin/bash
# HS-potential GEMM variants of the tuning build on the GPU box: kernel time (rocprofv3 --kernel-trace --stats) and HBM-side
# fetch traffic (separate --pmc FETCH_SIZE pass) per variant.   bash tools/vhs_traffic.sh <tag> "VAR=val ..." ...
out=gpurun_out/$1; shift
mkdir -p $out
export AFQ_LIBRARY=$PWD/pauxy_amd/libafqmc_hip_tuning.so
export TMPDIR=/tmp
i=0
for envs in "$@"; do
  i=$((i+1))
  ( export $envs; rocprofv3 --kernel-trace --stats --output-format csv -d $out/p$i -o p -- python3 bench.py --steps 40 --warmup 10 --repeats 1 --no-cpu-baseline > $out/run$i.json 2> $out/run$i.err )
  f=$(ls $out/p$i/*kernel_stats.csv $out/p$i/*/*kernel_stats.csv 2>/dev/null | head -1)
  echo "[$envs] $(grep -o '"ms_per_step": [0-9.]*' $out/run$i.json | head -1)"
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'VhsProb' in r['Name']:
        print("    %-70s calls %4s avg %9.1f us" % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3))
PY
  rm -rf $out/p$i
  for ctr in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    ( export $envs; rocprofv3 --pmc $ctr --output-format csv -d $out/c$i -o pmc -- python3 bench.py --steps 10 --warmup 5 --repeats 1 --no-cpu-baseline > /dev/null 2> $out/pmc$i.err )
    f=$(ls $out/c$i/*counter_collection.csv $out/c$i/*/*counter_collection.csv 2>/dev/null | head -1)
    [ -n "$f" ] && python3 tools/pmc_csv.py $f | grep VhsProb | sed 's/^/    /'
    rm -rf $out/c$i
  done
done
