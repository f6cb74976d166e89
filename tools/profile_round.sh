#!/bin/bash
# Collects the profile evidence of a round on the GPU box (rocprofv3): kernel-trace statistics of bench.py and of every
# BASELINE configuration through bench.py --config, and the HBM-side traffic counters (separate --pmc passes, as
# MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass) for bench.py and the UEG configuration.
#   bash tools/profile_round.sh <tag>       -> gpurun_out/<tag>/...
out=gpurun_out/${1:-prof}
mkdir -p $out
export TMPDIR=/tmp
run_stats() {   # name, command...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/$name -o $name -- "$@" > $out/$name.json 2> $out/$name.err
  f=$(ls $out/$name/*kernel_stats.csv $out/$name/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $out/${name}_kernel_stats.csv
  f=$(ls $out/$name/*kernel_trace.csv $out/$name/*/*kernel_trace.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $out/${name}_kernel_trace.csv
}
run_pmc() {     # name, counter, command...
  local name=$1 ctr=$2; shift; shift
  rocprofv3 --pmc $ctr --output-format csv -d $out/${name}_$ctr -o pmc -- "$@" > /dev/null 2> $out/${name}_$ctr.err
  f=$(ls $out/${name}_$ctr/*counter_collection.csv $out/${name}_$ctr/*/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && python3 tools/pmc_csv.py $f > $out/${name}_pmc_$ctr.txt
}
run_stats bench_c3 python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline
for c in C1 C2 C4 C5sd C5; do run_stats cfg_$c python3 bench.py --config $c --no-cpu-baseline; done
for ctr in FETCH_SIZE WRITE_SIZE; do
  run_pmc bench_c3 $ctr python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline
  for c in C2 C4 C5; do run_pmc cfg_$c $ctr python3 bench.py --config $c --no-cpu-baseline --repeats 1; done
done
ls $out
