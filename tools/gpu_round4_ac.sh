#!/bin/bash
# round 4, session ac: the step top of the streamed-actor kernels (fast against the old one: _sslow), smoothness as a weighted norm in the bench lines
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04ac
mkdir -p $O
cd $R
: > $O/ab.txt
for rep in 1 2; do
  for t in "" _sslow; do
    L=$R/serl_amd/csrc/libserl_amd$t.so
    SERL_LIB=$L AB_ACTORS=serl10 timeout 200 python tools/ab.py 30 >> $O/ab.txt 2>> $O/err.txt
    SERL_LIB=$L AB_ACTORS=td3 timeout 200 python tools/ab.py 30 >> $O/ab.txt 2>> $O/err.txt
  done
done
SERL_PROFILE=1 timeout 200 python tools/ab.py 150 >> $O/ab.txt 2>> $O/err.txt
cut -c1-120 $O/ab.txt | sed 's/.*libserl_amd//'
cd /tmp
timeout 600 python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_serl50.json 2> $O/bench_serl50.err
timeout 900 python $R/bench.py --total-pop 512 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_total512.json 2> $O/bench_total512.err
for f in serl50 total512; do python - $O/bench_$f.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], 'value %.4e ms/step %.2f kernel %.2f t_step_us %.2f' % (d['value'], d['ms_per_step'], d['kernel_ms'], d['t_step_us']), d.get('parity_vs_cpu_port'))
PY
done
cd $R
timeout 900 python -m pytest tests -x -q -m gpu --timeout=600 -k "smooth or evalpop or metrics or population_fitness or eval_pop or generation" > $O/pytest_gpu.txt 2>&1
tail -n 5 $O/pytest_gpu.txt
