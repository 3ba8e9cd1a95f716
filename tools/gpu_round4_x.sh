#!/bin/bash
# round 4, session x (re-entry after the container was replaced): the whole GPU suite on the tree of session w, the headline
# bench line with the reference's Python beside it, SERL10, rocprofv3 kernel stats and the SQ counter pass
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04x
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu --timeout=600 > $O/pytest_gpu.txt 2>&1
tail -n 5 $O/pytest_gpu.txt
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py"
timeout 600 $B --steps 10 --warmup 3 > $O/bench_serl50.json 2> $O/bench_serl50.err
timeout 600 $B --workload serl10 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_serl10.json 2> $O/bench_serl10.err
CMD="$B --steps 3 --warmup 1 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o r04x -- $CMD > $O/prof.log 2>&1
P1="$B --steps 1 --warmup 0 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES -d $O/pmc_sq -o ps -- $P1 > $O/pmc_sq.log 2>&1
cd $R
python tools/rocpd_summary.py $(ls $O/prof/*.db | head -1) "rocprofv3 --kernel-trace --stats -- $CMD" > $O/kernel_stats_serl50.md 2>> $O/err.txt
python tools/pmc_summary.py $O/pmc_sq > $O/pmc_sq.json 2>> $O/err.txt
SERL_PROFILE=1 timeout 300 python tools/ab.py 150 > $O/ab.txt 2>> $O/err.txt
rm -rf $O/prof $O/pmc_sq
for f in serl50 serl10; do python - $O/bench_$f.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], 'value %.3e ms/step %.2f kernel %.2f t_step_us %.2f' % (d['value'], d['ms_per_step'], d['kernel_ms'], d['t_step_us']), d.get('parity'))
PY
done
cat $O/pmc_sq.json | cut -c1-600
cut -c1-400 $O/ab.txt
