#!/bin/bash
# round 4, final tree: the whole GPU suite, then every file of the profile series (tools/profile_round.sh)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04fin
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu --timeout=600 > $O/pytest_gpu.txt 2>&1
tail -n 5 $O/pytest_gpu.txt
bash tools/profile_round.sh r04fin > $O/profile_round.log 2>&1
for f in serl50 total512 serl10 serl10_pop128 pop64 pop128 pop341 mixed mixed_total2048 rccl1; do python - $O/bench_$f.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(sys.argv[1].split('/')[-1], 'value %.4e ms/step %.2f kernel %.2f t_step_us %.2f' % (d['value'], d['ms_per_step'], d['kernel_ms'], d['t_step_us']), d.get('parity_vs_cpu_port'))
except Exception as ex:
    print(sys.argv[1], 'FAILED', ex)
PY
done
tail -n 3 $O/err.txt
