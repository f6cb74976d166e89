#!/bin/bash
# Repeats the self-launched multi-rank bench on ONE GPU (gloo bootstrap, peer-window communicator; auto = whole candidate
# chain) and counts the runs that print their line:  bash tools/multirank_repeat.sh <tag> <repeats>
out=gpurun_out/$1; n=${2:-10}; mkdir -p $out
: > $out/multirank_repeat.txt
for mode in ipc auto; do
  for r in 2 4 8; do
    ok=0
    for i in $(seq $n); do
      if AFQ_BENCH_BACKEND=gloo AFQ_BENCH_DEVICE_COMM=$mode timeout 300 python3 bench.py --gpus $r --walkers-per-gpu $((512/r)) --steps 20 --warmup 10 --no-cpu-baseline > $out/_mr.json 2> $out/_mr.err; then
        python3 - $out/_mr.json "$mode $r ranks run $i" >> $out/multirank_repeat.txt <<'PY' && ok=$((ok+1))
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
s = d["comm_stats_per_rank"]
assert s["error"] == [0] * d["n_gpus"] and s["overflow"] == [0] * d["n_gpus"]
print("%s: %.0f walker-steps/s, %s, events %d, walkers sent %s" % (sys.argv[2], d["value"], d["comm_probe"], s["events"][0], s["walkers_sent"]))
PY
      else echo "$mode $r ranks run $i: FAILED rc=$?" >> $out/multirank_repeat.txt; tail -5 $out/_mr.err >> $out/multirank_repeat.txt; fi
    done
    echo "$mode, $r ranks: $ok / $n runs printed their line with clean communicator flags" | tee -a $out/multirank_repeat.txt
  done
done
rm -f $out/_mr.json $out/_mr.err
