#!/bin/bash
# round 4, session ai: static role priorities on the new role map; the speculative-smoothness GPU test
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04ai
mkdir -p $O
cd $R
: > $O/ab.txt
for rep in 1 2; do
  for t in "" _pm00 _pm01 _pm03 _pm06 _pm0a _pm22 _pm42 _pm12; do
    SERL_LIB=$R/serl_amd/csrc/libserl_amd$t.so timeout 200 python tools/ab.py 150 >> $O/ab.txt 2>> $O/err.txt
  done
done
cut -c1-120 $O/ab.txt | sed 's/.*libserl_amd//'
timeout 600 python -m pytest tests/test_gpu_rollout.py -x -q -m gpu --timeout=300 -k "speculative" > $O/pytest_spec.txt 2>&1
tail -n 3 $O/pytest_spec.txt
