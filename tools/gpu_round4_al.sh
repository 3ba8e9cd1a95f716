#!/bin/bash
# round 4, session al: the bench line after the flag read moved in front of the all-gather (one collective per step on every rank)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04al
mkdir -p $O
cd /tmp
timeout 300 python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_serl50.json 2> $O/bench_serl50.err
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 $R/bench.py --gpus 1 --total-pop 512 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_total512_rccl1.json 2> $O/bench_total512_rccl1.err
for f in serl50 total512_rccl1; do python - $O/bench_$f.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(sys.argv[1].split('/')[-1], 'value %.4e ms/step %.2f kernel %.2f t_step_us %.2f' % (d['value'], d['ms_per_step'], d['kernel_ms'], d['t_step_us']), d.get('parity_vs_cpu_port'), (d.get('rccl') or {}).get('world_size'))
PY
done
tail -n 2 $O/*.err | cut -c1-200
