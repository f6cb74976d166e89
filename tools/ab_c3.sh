#!/bin/bash
# A/B of tuning-build variants on the headline configuration: bash tools/ab_c3.sh <tag> "VAR=val ..." ...
# prints ms/step, the traced propagator / exchange-energy launch and the last block's energy (bit-equality of variants)
out=gpurun_out/$1; shift
mkdir -p $out
export AFQ_LIBRARY=$PWD/pauxy_amd/libafqmc_hip_tuning.so
i=0
for envs in "$@"; do
  i=$((i+1))
  ( export $envs; python3 bench.py --no-cpu-baseline > $out/run$i.json 2> $out/run$i.err )
  python3 - "$out/run$i.json" "$envs" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("[%s] ms/step %.4f (%s) | prop %.1f us | exx %.1f us | E %.17g" % (sys.argv[2], d["ms_per_step"],
          " ".join("%.4f" % (x / d["steps"]) for x in d["timed_regions_ms"]), 1e3 * r["kernel_ms"],
          1e3 * r["cholesky_energy"]["kernel_ms"], d.get("last_block_ETotal") or float("nan")))
except Exception as e:
    print("[%s] FAILED %r" % (sys.argv[2], e))
PY
done
