#!/bin/bash
# A/B session on the GPU box: every library of tools/sweep_libs.txt through tools/ab.py (150 episodes x 2 001 steps), twice
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-sweep}
mkdir -p $O
cd $R
: > $O/ab.txt
for rep in $(seq ${REPS:-2}); do
  timeout ${TMO:-120} python tools/ab.py 150 >> $O/ab.txt 2>> $O/err.txt
  for t in $(cat tools/sweep_libs.txt); do
    SERL_LIB=$R/serl_amd/csrc/libserl_amd_$t.so timeout ${TMO:-120} python tools/ab.py 150 >> $O/ab.txt 2>> $O/err.txt
  done
done
cat $O/ab.txt | sed 's/.*libserl_amd_//' | cut -c1-110
