#!/bin/bash
# round 4, session aa: team step top (registers + one LDS round trip) against the old one (_slowtop), balancer bias of the role beside the actor
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04aa
mkdir -p $O
cd $R
: > $O/ab.txt
for rep in 1 2; do
  for t in "" _slowtop $(sed 's/^/_/' tools/sweep_libs.txt); do
    SERL_LIB=$R/serl_amd/csrc/libserl_amd$t.so SERL_PROFILE=1 timeout 200 python tools/ab.py 150 >> $O/ab.txt 2>> $O/err.txt
  done
done
cut -c1-120 $O/ab.txt | sed 's/.*libserl_amd//'
timeout 1500 python -m pytest tests -x -q -m gpu --timeout=600 > $O/pytest_gpu.txt 2>&1
tail -n 5 $O/pytest_gpu.txt
