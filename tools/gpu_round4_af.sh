#!/bin/bash
# round 4, session af: the two best role maps of the sweep beside a STREAMING actor (development build, map through SERL_JITTER_SITES)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04af
mkdir -p $O
cd $R
: > $O/ab.txt
for rep in 1 2; do
  for m in 71543260 76420531 75246310; do
    for actors in serl10 td3; do
      echo -n "map $m $actors " >> $O/ab.txt
      SERL_LIB=$R/serl_amd/csrc/libserl_amd_devroles.so SERL_JITTER_SITES=0x$m AB_ACTORS=$actors timeout 200 python tools/ab.py 30 >> $O/ab.txt 2>> $O/err.txt
    done
  done
done
cut -c1-40,95-190 $O/ab.txt
