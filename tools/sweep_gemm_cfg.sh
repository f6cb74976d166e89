#!/bin/bash
# A/B sweep of the work-group GEMM configurations (run on the GPU box via gpurun)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in 0 1 2 3 4; do
  export AFQ_VHS_CFG=$v; unset AFQ_FB_CFG; unset AFQ_FB_SPLIT
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sw_vhs$v -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/sw_vhs$v.log 2>&1
done
unset AFQ_VHS_CFG
for f in "1 8" "1 16" "2 8" "2 16" "3 4" "3 8" "0 8" "0 4"; do
  set -- $f
  export AFQ_FB_CFG=$1; export AFQ_FB_SPLIT=$2
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sw_fb$1_$2 -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/sw_fb$1_$2.log 2>&1
done
