#!/bin/bash
# round 4, session z: int32 exponent reduction in tanh / expm1 (against the 64-bit form: _k64), streamed-actor kernels without the whole-row
# forms, branch-free row loads, fixed-point barrier credit; issue priority of the streaming actor (_sp0 / _sp2 / _sp3); GPU suite; SERL10 benches
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04z
mkdir -p $O
cd $R
: > $O/ab.txt
for rep in 1 2; do
  for lib in "" _k64; do
    L=$R/serl_amd/csrc/libserl_amd$lib.so
    SERL_LIB=$L SERL_PROFILE=1 timeout 200 python tools/ab.py 150 >> $O/ab.txt 2>> $O/err.txt
  done
  for lib in "" _k64 _sp0 _sp2 _sp3; do
    L=$R/serl_amd/csrc/libserl_amd$lib.so
    SERL_LIB=$L AB_ACTORS=serl10 timeout 200 python tools/ab.py 30 >> $O/ab.txt 2>> $O/err.txt
    SERL_LIB=$L AB_ACTORS=td3 timeout 200 python tools/ab.py 30 >> $O/ab.txt 2>> $O/err.txt
  done
done
cut -c1-200 $O/ab.txt | sed 's/.*libserl_amd//'
timeout 1500 python -m pytest tests -x -q -m gpu --timeout=600 > $O/pytest_gpu.txt 2>&1
tail -n 5 $O/pytest_gpu.txt
cd /tmp
timeout 600 python $R/bench.py --workload serl10 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_serl10.json 2> $O/bench_serl10.err
timeout 600 python $R/bench.py --workload serl10 --pop 128 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_serl10_pop128.json 2> $O/bench_serl10_pop128.err
for f in serl10 serl10_pop128; do python - $O/bench_$f.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], 'value %.3e ms/step %.2f kernel %.2f t_step_us %.2f' % (d['value'], d['ms_per_step'], d['kernel_ms'], d['t_step_us']), d.get('parity'))
PY
done
