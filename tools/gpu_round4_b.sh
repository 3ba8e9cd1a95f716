#!/bin/bash
# round 4, session b: after the wavefront fences -- (1) stress of the shipped kernels and of the early-flag variants again (poison must
# be gone), (2) which class of pauses changes results (timing-ordered reads), (3) A/B timing: round-3 library, fenced library, early flag everywhere
mkdir -p gpurun_out
J=serl_amd/csrc/libserl_amd_jitter.so
for t in exp_ef3_jitter exp_ef1_jitter; do
  ( SERL_LIB=serl_amd/csrc/libserl_amd_$t.so timeout 300 python tests/tools/handover_stress.py h2000_v90 gpurun_out/stress_$t.npz 0 1 ) > gpurun_out/r04b_stress_$t.txt 2>&1
done
( SERL_LIB=$J timeout 900 python tools/jitter_classes.py h2000_v90 team teams team4 ) > gpurun_out/r04b_classes.txt 2>&1
for rep in 1 2; do
  for t in r03 "" exp_ef1; do
    L=serl_amd/csrc/libserl_amd${t:+_$t}.so
    SERL_LIB=$L timeout 200 python tools/ab.py 150 1023 >> gpurun_out/r04b_ab.txt 2>> gpurun_out/r04b_err.txt
  done
done
tail -n 8 gpurun_out/r04b_stress_*.txt; cat gpurun_out/r04b_classes.txt | head -30; cat gpurun_out/r04b_ab.txt
