#!/bin/bash
# round 4, session j: the specialised forward for H = 72 / 96 in the kernels; the LDS-actor team kernel without the streaming code (191 VGPRs, 61 KB)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04j
mkdir -p $O
cd $R
for a in serl10 td3; do
  AB_ACTORS=$a SERL_PROFILE=1 timeout 90 python tools/ab.py 30 384 >> $O/ab_streamed.txt 2>> $O/err.txt
  AB_ACTORS=$a SERL_SPLIT_ACTOR=1 timeout 90 python tools/ab.py 30 >> $O/ab_streamed.txt 2>> $O/err.txt
  AB_ACTORS=$a SERL_LIB=serl_amd/csrc/libserl_amd_r03.so timeout 90 python tools/ab.py 30 384 >> $O/ab_streamed.txt 2>> $O/err.txt
done
cut -c1-330 $O/ab_streamed.txt
for rep in 1 2; do
  for t in r03 ""; do
    L=serl_amd/csrc/libserl_amd${t:+_$t}.so
    SERL_LIB=$L timeout 200 python tools/ab.py 150 192 384 1023 >> $O/ab.txt 2>> $O/err.txt
  done
done
cat $O/ab.txt | sed 's/.*libserl_amd//' | cut -c1-330
timeout 1500 python -m pytest tests/test_gpu_rollout.py -x -q -m gpu --timeout=400 -k "streamed or handover or population_fitness or size_classes or full_size or env_configurations" > $O/pytest_j.txt 2>&1
tail -8 $O/pytest_j.txt
