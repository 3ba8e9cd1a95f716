#!/bin/bash
# round 4, session a: hand-over stress (poison + jitter) of the shipped team kernels and of the early-flag variants that gave
# first-launch NaNs in round 3 (EARLY_FLAG=1: every wavefront; =3: only the wavefront of the handed-over chain)
mkdir -p gpurun_out
J=serl_amd/csrc/libserl_amd_jitter.so
( SERL_LIB=$J timeout 600 python tests/tools/handover_stress.py h2000_v90 gpurun_out/stress_nominal.npz 0 1 2 3 ) > gpurun_out/r04a_stress_nominal.txt 2>&1
for t in exp_ef3_jitter exp_ef1_jitter; do
  ( SERL_LIB=serl_amd/csrc/libserl_amd_$t.so timeout 600 python tests/tools/handover_stress.py h2000_v90 gpurun_out/stress_$t.npz 0 1 2 3 ) > gpurun_out/r04a_stress_$t.txt 2>&1
done
tail -n 30 gpurun_out/r04a_stress_*.txt
