#!/bin/bash
# round 4, session k: role <-> wavefront permutations and static priorities on the re-balanced team (tools/sweeps/r04p_roles_prio.json); barrier profile
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04p
mkdir -p $O
cd $R
REPS=2 bash tools/sweep_run.sh r04p_sweep > $O/sweep_print.txt 2>&1
cat $O/sweep_print.txt
SERL_PROFILE=1 SERL_LIB=$R/serl_amd/csrc/libserl_amd_prof3.so timeout 300 python tools/ab.py 150 > $O/ab_prof3.txt 2>> $O/err.txt
grep -o '"phase_E150": \[[^]]*\]' $O/ab_prof3.txt
