#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03b
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -15 $O/pytest.log
timeout 300 python tools/bench_refill.py > $O/refill.json 2> $O/refill.err; tail -2 $O/refill.json $O/refill.err
timeout 300 python bench.py --workload serl10 --pop 128 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_serl10_pop128.json 2> $O/bench_serl10_pop128.err; cut -c1-400 $O/bench_serl10_pop128.json
timeout 300 python bench.py --pop 512 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_pop512.json 2> $O/bench_pop512.err; cut -c1-300 $O/bench_pop512.json
timeout 300 python bench.py --pop 341 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_pop341.json 2> $O/bench_pop341.err; cut -c1-300 $O/bench_pop341.json
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_serl50.json 2> $O/bench_serl50.err; cut -c1-300 $O/bench_serl50.json
