#!/bin/bash
# round 4, session aj: lane-group team kernels -- the env step's books on ONE team wavefront (default: the last one; _bwN: wavefront N)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04aj
mkdir -p $O
cd $R
: > $O/ab.txt
for rep in 1 2; do
  for t in "" _bw0 _bw2 _bw4 _bw5; do
    L=$R/serl_amd/csrc/libserl_amd$t.so
    SERL_LIB=$L timeout 200 python tools/ab.py 384 1023 >> $O/ab.txt 2>> $O/err.txt
  done
done
SERL_LIB=$R/serl_amd/csrc/libserl_amd.so AB_ACTORS=serl10 timeout 200 python tools/ab.py 384 >> $O/ab.txt 2>> $O/err.txt
SERL_LIB=$R/serl_amd/csrc/libserl_amd_bw0.so AB_ACTORS=serl10 timeout 200 python tools/ab.py 384 >> $O/ab.txt 2>> $O/err.txt
cut -c1-230 $O/ab.txt | sed 's/.*libserl_amd//'
