#!/bin/bash
# round 4, session c: derivative stores behind B1 -- jitter classes again, then the new GPU tests (stress of all five variants vs the oracle,
# first launch of a process for all variants)
mkdir -p gpurun_out
J=serl_amd/csrc/libserl_amd_jitter.so
( SERL_LIB=$J timeout 900 python tools/jitter_classes.py h2000_v90 team teams team2 team4 ) > gpurun_out/r04c_classes.txt 2>&1
grep -v "^{" gpurun_out/r04c_classes.txt | tail -18
timeout 2400 python -m pytest tests/test_gpu_rollout.py -x -q -m gpu -k "handover or first_launch" > gpurun_out/r04c_pytest.txt 2>&1
tail -30 gpurun_out/r04c_pytest.txt
