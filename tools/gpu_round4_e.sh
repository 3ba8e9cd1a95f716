#!/bin/bash
# round 4, session e: short sincos / tan / pow -- whole GPU suite, A/B against the round-3 library, mixed-sweep breakdown
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04e
mkdir -p $O
cd $R
for rep in 1 2; do
  for t in r03 ""; do
    L=serl_amd/csrc/libserl_amd${t:+_$t}.so
    SERL_LIB=$L timeout 200 python tools/ab.py 150 384 1023 >> $O/ab.txt 2>> $O/err.txt
  done
done
AB_ACTORS=serl10 SERL_LIB=serl_amd/csrc/libserl_amd_r03.so timeout 200 python tools/ab.py 30 384 >> $O/ab_serl10.txt 2>> $O/err.txt
AB_ACTORS=serl10 timeout 200 python tools/ab.py 30 384 >> $O/ab_serl10.txt 2>> $O/err.txt
cat $O/ab.txt $O/ab_serl10.txt
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1
tail -40 $O/pytest_gpu.txt
timeout 600 python tools/mixed_breakdown.py 256 $O/mixed_by_build.json > $O/mixed_breakdown.txt 2>&1
tail -3 $O/mixed_breakdown.txt
