#!/bin/bash
# A/B sweep of the HS-potential GEMM tile configurations (run on the GPU box via gpurun)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in 0 1 2 3 4 5 6 7 8 9 10; do
  export AFQ_VHS_CFG=$v
  rm -rf $R/gpurun_out/sw_vhs$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sw_vhs$v -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/sw_vhs$v.log 2>&1
  echo "cfg=$v $(grep VhsProb $R/gpurun_out/sw_vhs$v/*/*kernel_stats.csv | cut -d, -f2-4 | tail -1)"
done
