#!/bin/bash
# round 4, session d: whole GPU suite after the hand-over fixes; mixed-sweep breakdown and the split-grid queue launches; host-side tail at
# 1 536 episodes; instruction-cache counters of the team / team4 kernels
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04d
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1
tail -5 $O/pytest_gpu.txt
timeout 600 python tools/mixed_breakdown.py 256 $O/mixed_by_build.json > $O/mixed_breakdown.txt 2>&1
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py"
timeout 900 $B --steps 10 --warmup 3 > $O/bench_serl50.json 2> $O/bench_serl50.err
timeout 600 $B --workload mixed --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_mixed.json 2> $O/bench_mixed.err
timeout 900 $B --workload mixed --total-pop 2048 --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_mixed_total2048.json 2> $O/bench_mixed_total2048.err
timeout 900 $B --total-pop 512 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_total512.json 2> $O/bench_total512.err
rocprofv3 --list-avail 2>/dev/null | grep -i "icache\|ifetch\|SQ_WAIT_INST\|INST_CACHE" | head -40 > $O/avail_icache.txt
P1="$B --steps 1 --warmup 0 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH -d $O/pmc_ic -o ic -- $P1 > $O/pmc_ic.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH -d $O/pmc_ic4 -o ic4 -- $B --pop 341 --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_ic4.log 2>&1
cd $R
python tools/pmc_summary.py $O/pmc_ic > $O/pmc_icache_team.json 2>> $O/err.txt
python tools/pmc_summary.py $O/pmc_ic4 > $O/pmc_icache_team4.json 2>> $O/err.txt
rm -rf $O/pmc_ic $O/pmc_ic4
cat $O/mixed_breakdown.txt | tail -2; for f in bench_serl50 bench_mixed bench_mixed_total2048 bench_total512; do python - $O/$f.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], 'value %.3g' % d['value'], 'ms_per_step %.2f kernel_ms %.2f' % (d['ms_per_step'], d['kernel_ms']), 't_step_us %.2f' % d['t_step_us'], d.get('parity_vs_cpu_port'))
except Exception as e:
    print(sys.argv[1], 'unreadable', e)
PY
done
cat $O/pmc_icache_team.json $O/pmc_icache_team4.json $O/avail_icache.txt
