#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03c
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -n 15 $O/pytest.log
timeout 300 python bench.py --workload serl10 --pop 128 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_serl10_pop128.json 2> $O/bench_serl10_pop128.err; cut -c1-200 $O/bench_serl10_pop128.json
timeout 300 python bench.py --workload serl10 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_serl10.json 2> $O/bench_serl10.err; cut -c1-200 $O/bench_serl10.json
