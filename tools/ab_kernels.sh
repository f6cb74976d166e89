#!/bin/bash
# A/B of tuning-build variants, no profiler: bench.py's own event timings of the hot kernels.
#   bash tools/ab_kernels.sh <tag> "VAR=val ..." ...
out=gpurun_out/$1; shift
mkdir -p $out
export AFQ_LIBRARY=$PWD/pauxy_amd/libafqmc_hip_tuning.so
i=0
for envs in "$@"; do
  i=$((i+1))
  ( export $envs; python3 bench.py --no-cpu-baseline > $out/run$i.json 2> $out/run$i.err )
  python3 - "$out/run$i.json" "$envs" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("[%s] ms/step %.4f | %s" % (sys.argv[2], d["ms_per_step"], "  ".join("%s %.1f" % (r["kernel"].split("<")[1].split(">")[0][:8] if "<" in r["kernel"] else r["kernel"][:8], 1e3 * r["avg_ms"]) for r in d["roofline_all"])))
PY
done
