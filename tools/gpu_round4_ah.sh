#!/bin/bash
# round 4, session ah: stage time from a six-entry LDS table (_tv2) against the shipped choice
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04ah
mkdir -p $O
cd $R
: > $O/ab.txt
for rep in 1 2 3; do
  for t in "" _tv2; do
    L=$R/serl_amd/csrc/libserl_amd$t.so
    SERL_LIB=$L timeout 200 python tools/ab.py 150 1023 >> $O/ab.txt 2>> $O/err.txt
    SERL_LIB=$L AB_ACTORS=serl10 timeout 200 python tools/ab.py 30 384 >> $O/ab.txt 2>> $O/err.txt
  done
done
cut -c1-260 $O/ab.txt | sed 's/.*libserl_amd//'
