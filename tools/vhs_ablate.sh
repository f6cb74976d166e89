#!/bin/bash
# Timing ablations of the HS-potential GEMM (tuning build, AFQ_VHS_ABL bits: 1 no MFMAs, 2 no refill, 4 no fragment reads,
# 8 no barrier, 16 no stores; results are WRONG with any bit set): kernel time per variant.   bash tools/vhs_ablate.sh <tag> <abl> ...
out=gpurun_out/$1; shift
mkdir -p $out
export AFQ_LIBRARY=$PWD/pauxy_amd/libafqmc_hip_tuning.so
export TMPDIR=/tmp
for abl in "$@"; do
  ( export AFQ_VHS_ABL=$abl AFQ_GEMM_TS=1; rocprofv3 --kernel-trace --stats --output-format csv -d $out/p$abl -o p -- python3 bench.py --steps 40 --warmup 10 --repeats 1 --no-cpu-baseline > $out/run$abl.json 2> $out/run$abl.err )
  f=$(ls $out/p$abl/*kernel_stats.csv $out/p$abl/*/*kernel_stats.csv 2>/dev/null | head -1)
  python3 - "$f" "$abl" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'VhsProb' in r['Name']:
        print("abl %3s  %-60s calls %4s avg %9.1f us" % (sys.argv[2], r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3))
PY
  grep "GEMM_TS after VHS wg  0" $out/run$abl.err | head -1
  rm -rf $out/p$abl
done
