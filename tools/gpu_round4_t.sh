#!/bin/bash
# round 4, session v: chunked forward adopted (chunks of 8): parity tests that meet streamed actors, benches of the SERL10 shape
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04v
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_rollout.py -x -q -m gpu --timeout=400 -k "streamed or handover or population_fitness or size_classes or full_size or env_configurations or first_launch or rounding" > $O/pytest_v.txt 2>&1
tail -5 $O/pytest_v.txt
cd /tmp
timeout 600 python $R/bench.py --workload serl10 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_serl10.json 2> $O/bench_serl10.err
timeout 600 python $R/bench.py --workload serl10 --pop 128 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_serl10_pop128.json 2> $O/bench_serl10_pop128.err
for f in serl10 serl10_pop128; do python - $O/bench_$f.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], 'value %.3e ms/step %.2f kernel %.2f t_step_us %.2f' % (d['value'], d['ms_per_step'], d['kernel_ms'], d['t_step_us']))
PY
done
