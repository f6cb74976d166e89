#!/bin/bash
# A/B runs of bench.py in a tuning build on the GPU box:  bash tools/ab_bench.sh <tag> "VAR=val ..." "VAR2=val ..." ...
out=gpurun_out/$1; shift
mkdir -p $out
make -C pauxy_amd/csrc -j32 TUNING=1 > $out/build.log 2>&1 || { tail -5 $out/build.log; exit 1; }
export AFQ_LIBRARY=$PWD/pauxy_amd/libafqmc_hip_tuning.so      # the product library is left alone
i=0
for envs in "$@"; do
  i=$((i+1))
  ( export $envs; python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline > $out/run$i.json 2> $out/run$i.err )
  python3 - <<PY
import json
try:
    d=json.load(open("$out/run$i.json"))
    print("[%s] %.1f k w-steps/s  %.4f ms/step | " % ("$envs", d["value"]/1e3, d["ms_per_step"]) + "  ".join("%s %.1f" % (r["kernel"].split("<")[-1].split(">")[0][:14] if "gemm" in r["kernel"] else r["kernel"][:10], r["avg_ms"]*1e3) for r in d["roofline_all"]))
except Exception as e:
    print("[%s] FAILED %r" % ("$envs", e)); print(open("$out/run$i.err").read()[-400:])
PY
done
