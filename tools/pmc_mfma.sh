#!/bin/bash
# Matrix-pipe busy cycles of the C3 kernels from the hardware counter (cross-check of bench.py's frac_issued):
#   bash tools/pmc_mfma.sh <tag>   -> gpurun_out/<tag>/bench_c3_pmc_<counter>.txt   (separate --pmc passes, no tracing)
out=gpurun_out/${1:-pmc_mfma}; mkdir -p $out
export TMPDIR=/tmp
for ctr in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  rocprofv3 --pmc $ctr --output-format csv -d $out/c3_$ctr -o pmc -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --repeats 1 > /dev/null 2> $out/c3_$ctr.err
  f=$(ls $out/c3_$ctr/*counter_collection.csv $out/c3_$ctr/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 tools/pmc_csv.py $f > $out/bench_c3_pmc_$ctr.txt
  rm -rf $out/c3_$ctr
done
grep -h "prop_fused\|VhsProb\|ForceBias\|ExxQProb" $out/bench_c3_pmc_*.txt | cut -c1-60,93-
