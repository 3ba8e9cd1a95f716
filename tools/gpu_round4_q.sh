#!/bin/bash
# round 4, session q: the role permutation / static priority of the one-episode kernels applied to two and four episodes per team
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04q
mkdir -p $O
cd $R
for rep in 1 2; do
  timeout 200 python tools/ab.py 384 1023 >> $O/ab.txt 2>> $O/err.txt
  for t in $(cat tools/sweep_libs.txt); do
    SERL_LIB=$R/serl_amd/csrc/libserl_amd_$t.so timeout 200 python tools/ab.py 384 1023 >> $O/ab.txt 2>> $O/err.txt
  done
done
cat $O/ab.txt | sed 's/.*libserl_amd_//' | cut -c1-200
