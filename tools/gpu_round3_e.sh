#!/bin/bash
# round 3, session e: GPU tests of the tree with per-step invariants / precomputed look-up lanes / branch-free stores, then
# one / two / four episodes per team with the stores of the lane-group kernels branch-free too (libserl_amd_exp_us3all.so)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03e
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -n 5 $O/pytest.log
for rep in 1 2; do
  timeout 200 python tools/ab.py 150 384 1023 >> $O/ab.txt 2>> $O/err.txt
  SERL_LIB=$R/serl_amd/csrc/libserl_amd_exp_us3all.so timeout 200 python tools/ab.py 150 384 1023 >> $O/ab.txt 2>> $O/err.txt
done
cat $O/ab.txt | sed 's/.*libserl_amd_//' | cut -c1-260
