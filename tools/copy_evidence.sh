#!/bin/bash
# Copies what tools/final_evidence.sh left under gpurun_out/<tag> into profiles/<round>_*:  bash tools/copy_evidence.sh <tag> <round>
src=gpurun_out/$1; r=$2
last() { tail -1 $1 > $2; }
last $src/bench_c3.json profiles/${r}_bench_c3.json
last $src/bench_c3_strong_n1.json profiles/${r}_bench_c3_strong_n1.json
for c in C1 C2 C4 C5sd C5; do last $src/cfg_$c.json profiles/${r}_cfg_$c.json; done
for n in 2 4 8; do last $src/bench_${n}ranks_one_gpu_ipc.json profiles/${r}_bench_${n}ranks_one_gpu_ipc.json; done
last $src/bench_c3_open_shell.json profiles/${r}_bench_c3_open_shell.json
last $src/bench_2ranks_one_gpu_ipc_stress.json profiles/${r}_bench_2ranks_one_gpu_ipc_stress.json
for c in C4 C5; do last $src/cfg_${c}_2ranks_one_gpu_ipc.json profiles/${r}_cfg_${c}_2ranks_one_gpu_ipc.json; done
cp $src/stress_2000_c3.log profiles/${r}_stress_2000_end_of_round.log
cp $src/fresh_process_runs.txt profiles/${r}_fresh_process_runs_c1_c2_c4_c5.txt
git status --short profiles | head -20
