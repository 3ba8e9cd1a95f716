#!/bin/bash
# round 4, session w: the env step's books moved from the team's first wavefront to the actor wavefront
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04w
mkdir -p $O
cd $R
SERL_PROFILE=1 timeout 300 python tools/ab.py 150 > $O/ab.txt 2>> $O/err.txt
cat $O/ab.txt $O/ab_serl10.txt $O/ab_td3.txt | cut -c1-700
timeout 1800 python -m pytest tests/test_gpu_rollout.py -x -q -m gpu --timeout=600 > $O/pytest_w.txt 2>&1
tail -5 $O/pytest_w.txt
cd /tmp
timeout 600 python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_serl50.json 2> $O/bench_serl50.err
timeout 600 python $R/bench.py --workload serl10 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_serl10.json 2> $O/bench_serl10.err
for f in serl50 serl10; do python - $O/bench_$f.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], 'value %.3e ms/step %.2f kernel %.2f t_step_us %.2f' % (d['value'], d['ms_per_step'], d['kernel_ms'], d['t_step_us']), d.get('parity'))
PY
done
