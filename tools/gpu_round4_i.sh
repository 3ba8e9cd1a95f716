#!/bin/bash
# round 4, session i: where the split actor's time goes (SERL_PROFILE), with / without issue priority, against the lone streaming actor
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04i
mkdir -p $O
cd $R
for a in serl10 td3; do
  AB_ACTORS=$a SERL_PROFILE=1 timeout 90 python tools/ab.py 30 >> $O/ab.txt 2>> $O/err.txt
  AB_ACTORS=$a SERL_PROFILE=1 SERL_LIB=serl_amd/csrc/libserl_amd_exp_noprio.so timeout 90 python tools/ab.py 30 >> $O/ab.txt 2>> $O/err.txt
  AB_ACTORS=$a SERL_PROFILE=1 SERL_SPLIT_ACTOR=0 timeout 90 python tools/ab.py 30 >> $O/ab.txt 2>> $O/err.txt
done
cat $O/ab.txt | cut -c1-600
