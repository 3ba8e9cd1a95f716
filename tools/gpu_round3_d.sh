#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03d
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -n 4 $O/pytest.log
timeout 900 python bench.py --workload mixed --total-pop 2048 --steps 2 --warmup 1 > $O/bench_mixed_total2048.json 2> $O/bench_mixed_total2048.err; cut -c1-250 $O/bench_mixed_total2048.json; tail -n 3 $O/bench_mixed_total2048.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 1 --total-pop 512 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_total512_rccl1.json 2> $O/bench_total512_rccl1.err; cut -c1-250 $O/bench_total512_rccl1.json
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-200 $O/bench_default.json
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -n 2 $O/smoke.log
