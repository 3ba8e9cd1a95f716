#!/bin/bash
# first GPU session of round 3: GPU tests, default bench (reference Python timed in-run), strong-scaling line, 7-wave phase profile
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03a
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench_serl50.json 2> $O/bench_serl50.err
timeout 900 python bench.py --total-pop 512 --steps 3 --warmup 1 > $O/bench_total512.json 2> $O/bench_total512.err
SERL_PROFILE=1 SERL_LIB=$R/serl_amd/csrc/libserl_amd_prof.so timeout 300 python tools/ab.py 150 > $O/ab_prof1.txt 2>> $O/err.txt
SERL_PROFILE=1 SERL_LIB=$R/serl_amd/csrc/libserl_amd_prof2.so timeout 300 python tools/ab.py 150 > $O/ab_prof2.txt 2>> $O/err.txt
SERL_PROFILE=1 timeout 300 python tools/ab.py 150 > $O/ab.txt 2>> $O/err.txt
nproc > $O/nproc.txt; free -g >> $O/nproc.txt
tail -3 $O/pytest.log
