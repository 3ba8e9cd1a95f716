#!/bin/bash
# round 4, session y: the actor's previous layer through an LDS row (SERL_ACTOR_LDS_BCAST) against the readlane build, then the GPU suite
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04y
mkdir -p $O
cd $R
: > $O/ab.txt
for rep in 1 2; do
  for lib in "" _nobc; do
    L=$R/serl_amd/csrc/libserl_amd$lib.so
    SERL_LIB=$L SERL_PROFILE=1 timeout 200 python tools/ab.py 150 >> $O/ab.txt 2>> $O/err.txt
    SERL_LIB=$L AB_ACTORS=serl10 timeout 200 python tools/ab.py 30 >> $O/ab.txt 2>> $O/err.txt
    SERL_LIB=$L AB_ACTORS=td3 timeout 200 python tools/ab.py 30 >> $O/ab.txt 2>> $O/err.txt
  done
done
cut -c1-330 $O/ab.txt
timeout 1500 python -m pytest tests -x -q -m gpu --timeout=600 > $O/pytest_gpu.txt 2>&1
tail -n 5 $O/pytest_gpu.txt
