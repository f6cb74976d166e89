#!/bin/bash
# A/B of tuning-build variants on another BASELINE configuration (bench.py --config, launch trace): bash tools/ab_config.sh <tag> <config> "VAR=val ..." ...
out=gpurun_out/$1; cfg=$2; shift; shift
mkdir -p $out
export AFQ_LIBRARY=$PWD/pauxy_amd/libafqmc_hip_tuning.so
i=0
for envs in "$@"; do
  i=$((i+1))
  ( export $envs; python3 bench.py --config $cfg --no-cpu-baseline > $out/run$i.json 2> $out/run$i.err )
  python3 - "$out/run$i.json" "$envs" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("[%s] ms/step %.4f | %s" % (sys.argv[2], d["ms_per_step"], "  ".join("%s %.1f" % (r["launch"][:18], 1e3 * r["avg_ms"]) for r in d["roofline_all"][:8])))
except Exception as e:
    print("[%s] FAILED %r" % (sys.argv[2], e))
PY
done
