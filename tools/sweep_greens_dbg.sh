#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in 0 1 2 4 7; do
  export AFQ_GREENS_DBG=$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/gd_$v -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/gd_$v.log 2>&1
done
