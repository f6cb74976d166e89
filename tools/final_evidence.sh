#!/bin/bash
# The bench lines and stress runs a round keeps under profiles/ (after tools/profile_round.sh + tools/publish_profiles.py, so
# that the lines quote the traffic of the same library):  bash tools/final_evidence.sh <tag>
out=gpurun_out/$1; mkdir -p $out
python3 bench.py > $out/bench_c3.json 2> $out/bench_c3.err
python3 bench.py --gpus 1 --scaling strong --no-cpu-baseline > $out/bench_c3_strong_n1.json 2> $out/bench_c3_strong_n1.err
python3 bench.py --open-shell --no-cpu-baseline > $out/bench_c3_open_shell.json 2> $out/bench_c3_open_shell.err
for c in C1 C2 C4 C5sd C5; do python3 bench.py --config $c > $out/cfg_$c.json 2> $out/cfg_$c.err; done
for n in 2 4 8; do
  AFQ_BENCH_BACKEND=gloo AFQ_BENCH_DEVICE_COMM=ipc python3 bench.py --gpus $n --walkers-per-gpu $((512/n)) --no-cpu-baseline > $out/bench_${n}ranks_one_gpu_ipc.json 2> $out/bench_${n}ranks.err
done
# the exchange under load (10 % of the walkers cross ranks per comb) and the 8-GPU configurations on two ranks (weak scaling)
AFQ_BENCH_BACKEND=gloo AFQ_BENCH_DEVICE_COMM=ipc python3 bench.py --gpus 2 --no-cpu-baseline --exchange-stress 0.1 > $out/bench_2ranks_one_gpu_ipc_stress.json 2> $out/bench_2ranks_stress.err
for c in C4 C5; do
  AFQ_BENCH_BACKEND=gloo AFQ_BENCH_DEVICE_COMM=ipc python3 bench.py --config $c --gpus 2 --no-cpu-baseline > $out/cfg_${c}_2ranks_one_gpu_ipc.json 2> $out/cfg_${c}_2ranks.err
done
make -C tools stress > /dev/null 2>&1
python3 tools/stress_inputs.py /tmp/afq_stress_c3.bin > /dev/null 2>&1
tools/stress --inputs /tmp/afq_stress_c3.bin --iters 2000 --parallel 8 --steps 100 --timeout 60 > $out/stress_2000_c3.log 2>&1
tail -1 $out/stress_2000_c3.log
# fresh-process runs of the other configurations (new kernels of the round: tiny Green's function, UEG fields, averaged-G fold)
: > $out/fresh_process_runs.txt
for c in C2 C1 C4 C5; do
  n=10; [ $c = C4 ] && n=5; [ $c = C5 ] && n=3
  ok=0
  for i in $(seq $n); do
    if timeout 300 python3 bench.py --config $c --no-cpu-baseline --repeats 1 > $out/_fresh.json 2> $out/_fresh.err; then
      ok=$((ok+1)); python3 -c "import json,sys; d=json.loads(open('$out/_fresh.json').read().strip().splitlines()[-1]); print('$c run $i: %.4f ms/step, E %.10g' % (d['ms_per_step'], d['last_block_ETotal'] or 0))" >> $out/fresh_process_runs.txt
    else echo "$c run $i: FAILED rc=$?" >> $out/fresh_process_runs.txt; tail -3 $out/_fresh.err >> $out/fresh_process_runs.txt; fi
  done
  echo "$c: $ok / $n fresh-process runs completed" | tee -a $out/fresh_process_runs.txt
done
rm -f $out/_fresh.json $out/_fresh.err
