#!/bin/bash
# On the GPU box: SQ counters of prop_fused_kernel with the 4x4x4 Taylor products and with the 16x16x4 ones
# (tuning build: the AFQ_* switches exist only with make TUNING=1).  Output under gpurun_out/$1.
out=gpurun_out/${1:-pmc_prop}
mkdir -p $out
make -C pauxy_amd/csrc -j32 TUNING=1 > $out/build.log 2>&1 || { tail -5 $out/build.log; exit 1; }
export AFQ_LIBRARY=$PWD/pauxy_amd/libafqmc_hip_tuning.so      # the product library is left alone
export TMPDIR=/tmp
for v in t4 t16; do
  if [ $v = t16 ]; then export AFQ_NO_T4=1; else unset AFQ_NO_T4; fi
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE \
     -d $out/$v -o pmc -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/$v.json 2> $out/$v.err
  python3 profiles/tools_pmc.py $out/$v > $out/$v.txt 2>&1
  grep -A9 "prop_fused" $out/$v.txt | head -12
done
