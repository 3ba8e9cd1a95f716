#!/bin/bash
# round 4, session ad: flags that were tuned on an older kernel, again (ODE combine split around the evaluation, static role priorities, wave 0 at priority)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04ad
mkdir -p $O
cd $R
: > $O/ab.txt
for rep in 1 2; do
  for t in "" _odesplit _prio00 _prio20 _prio22 _prio06 _w0prio; do
    SERL_LIB=$R/serl_amd/csrc/libserl_amd$t.so timeout 200 python tools/ab.py 150 >> $O/ab.txt 2>> $O/err.txt
  done
done
cut -c1-120 $O/ab.txt | sed 's/.*libserl_amd//'
