#!/bin/bash
# round 4, session m: directed bias sweep on top of the adopted role permutation (tools/sweeps/r04m_bias_directed.json)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04m
mkdir -p $O
cd $R
REPS=1 bash tools/sweep_run.sh r04m_sweep > $O/sweep_print.txt 2>&1
sort -t: -k3 $O/sweep_print.txt | cut -c1-110
