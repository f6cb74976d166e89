#!/bin/bash
# A/B runs of the C5 single-determinant configuration in a tuning build on the GPU box:  bash tools/ab_c5.sh <tag> "VAR=val" ...
out=gpurun_out/$1; shift
mkdir -p $out
make -C pauxy_amd/csrc -j32 TUNING=1 > $out/build.log 2>&1 || { tail -5 $out/build.log; exit 1; }
export AFQ_LIBRARY=$PWD/pauxy_amd/libafqmc_hip_tuning.so      # the product library is left alone
export TMPDIR=/tmp
i=0
for envs in "$@"; do
  i=$((i+1))
  ( export $envs; rocprofv3 --kernel-trace --stats --output-format csv -d $out/p$i -o c5 -- python3 bench.py --config C5sd --no-cpu-baseline > $out/run$i.json 2> $out/run$i.err )
  f=$(ls $out/p$i/*kernel_stats.csv $out/p$i/*/*kernel_stats.csv 2>/dev/null | head -1)
  echo "[$envs] $(cut -c1-140 $out/run$i.json | grep -o '"ms_per_step": [0-9.]*')"
  grep -E "TaylorProb|OneBodyProb" $f | cut -d, -f1-4 | cut -c1-150
done
