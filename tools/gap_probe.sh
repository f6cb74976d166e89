#!/bin/bash
# Kernel trace of bench.py in a tuning build for two settings (gap analysis): bash tools/gap_probe.sh <tag> "VAR=val" "VAR=val"
out=gpurun_out/$1; shift
mkdir -p $out; export TMPDIR=/tmp
export AFQ_LIBRARY=$PWD/pauxy_amd/libafqmc_hip_tuning.so
i=0
for envs in "$@"; do
  i=$((i+1))
  ( export $envs; rocprofv3 --kernel-trace --output-format csv -d $out/p$i -o p -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline > $out/run$i.json 2> $out/run$i.err )
  f=$(ls $out/p$i/*kernel_trace.csv $out/p$i/*/*kernel_trace.csv 2>/dev/null | head -1)
  python3 - "$f" > $out/trace$i.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last 2500 kernel launches: name, start, end (ns relative)
t0 = int(rows[0]['Start_Timestamp'])
for r in rows[-4000:]:
    print(r['Kernel_Name'][:40].replace(' ', '_'), int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0)
PY
  rm -rf $out/p$i
  echo "[$envs] $(grep -o '"ms_per_step": [0-9.]*' $out/run$i.json | head -1)"
done
