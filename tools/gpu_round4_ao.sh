#!/bin/bash
# round 4, session ao: a first look at role maps for the FOUR-per-team kernel (development build; the 15 best pairings of the one-per-team sweep + the identity), for the next round
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04ao
mkdir -p $O
cd $R
: > $O/ab.txt
for m in 76543210 76420531 75246310 75620431 76520431 74526310 71543260 75426310 76423510 74256310 76523410 76421530 74650321 76453210 75621430; do
  echo -n "map $m " >> $O/ab.txt
  SERL_LIB=$R/serl_amd/csrc/libserl_amd_devroles.so SERL_JITTER_SITES=0x$m timeout 30 python tools/ab.py 1023 >> $O/ab.txt 2>> $O/err.txt
done
cut -c1-14,60-150 $O/ab.txt
