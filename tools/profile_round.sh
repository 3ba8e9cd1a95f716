#!/bin/bash
# Collect the round's measurements on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh <tag>        -> gpurun_out/<tag>/...
# bench lines of every workload, rocprofv3 --kernel-trace --stats of the default bench, separate --pmc passes
# (FETCH_SIZE / WRITE_SIZE / SQ issue counters, as MI355X_MICROARCH.md prescribes), the in-kernel cycle profile (two profiling
# builds: waves 0 - 3 and waves 4 - 6), the one-wavefront-per-SIMD instruction micro-benchmark and the DAG floors computed from it.
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
(cd $R && python serl_amd/build.py --source-hash) > $O/csrc_sha256.txt      # what these measurements belong to (bench.py quotes them only while it matches)
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py"
timeout 900 $B --steps 10 --warmup 3 > $O/bench_serl50.json 2> $O/bench_serl50.err
timeout 900 $B --total-pop 512 --steps 3 --warmup 1 > $O/bench_total512.json 2> $O/bench_total512.err                     # BASELINE config 4 on one GPU, parity vs the CPU restatement on all 1 536 episodes
timeout 600 $B --workload serl10 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_serl10.json 2> $O/bench_serl10.err
SERL_REMOTE_ACTOR=0 timeout 600 $B --workload serl10 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_serl10_samecu.json 2> $O/bench_serl10_samecu.err      # A/B: the streamed actor on the team's CU (round 5's kernel)
timeout 600 $B --workload serl10 --pop 128 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_serl10_pop128.json 2> $O/bench_serl10_pop128.err   # 384 episodes, H = 72: two per team, two actor wavefronts
timeout 600 $B --pop 64 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_pop64.json 2> $O/bench_pop64.err
timeout 600 $B --pop 128 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_pop128.json 2> $O/bench_pop128.err     # 384 episodes: two per team
timeout 600 $B --pop 341 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_pop341.json 2> $O/bench_pop341.err     # 1 023 episodes: four per team
timeout 600 $B --workload mixed --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_mixed.json 2> $O/bench_mixed.err                 # 768 episodes, three builds: ONE launch of one code object, variants placed by CU pair
timeout 600 $B --workload mixed --no-fused --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_mixed_nofused.json 2> $O/bench_mixed_nofused.err   # ... against a launch per build side by side
timeout 900 $B --workload mixed --total-pop 2048 --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_mixed_total2048.json 2> $O/bench_mixed_total2048.err   # BASELINE config 5 on one GPU: 6 144 episodes through the parts' work queues
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 $R/bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_rccl1.json 2> $O/bench_rccl1.err
(cd $R && timeout 300 python tools/bench_refill.py > $O/refill.json 2> $O/refill.err)
CMD="$B --steps 3 --warmup 1 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o ${TAG} -- $CMD > $O/prof.log 2>&1
P1="$B --steps 1 --warmup 0 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o pf -- $P1 > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o pw -- $P1 > $O/pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES -d $O/pmc_sq -o ps -- $P1 > $O/pmc_sq.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_serl10 -o ${TAG}s10r -- $B --workload serl10 --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_serl10.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_serl10p128 -o ${TAG}s10 -- $B --workload serl10 --pop 128 --steps 2 --warmup 1 --no-cpu-baseline > $O/prof_serl10p128.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_pop512 -o ${TAG}p512 -- $B --pop 512 --steps 2 --warmup 1 --no-cpu-baseline > $O/prof_pop512.log 2>&1
cd $R
python tools/rocpd_summary.py $(ls $O/prof/*.db | head -1) "rocprofv3 --kernel-trace --stats -- $CMD" > $O/kernel_stats_serl50.md 2>> $O/err.txt
python tools/rocpd_summary.py $(ls $O/prof_serl10/*.db | head -1) "rocprofv3 --kernel-trace --stats -- python bench.py --workload serl10 --steps 3 --warmup 1 --no-cpu-baseline" > $O/kernel_stats_serl10.md 2>> $O/err.txt
python tools/rocpd_summary.py $(ls $O/prof_serl10p128/*.db | head -1) "rocprofv3 --kernel-trace --stats -- python bench.py --workload serl10 --pop 128 --steps 2 --warmup 1 --no-cpu-baseline" > $O/kernel_stats_serl10_pop128.md 2>> $O/err.txt
python tools/rocpd_summary.py $(ls $O/prof_pop512/*.db | head -1) "rocprofv3 --kernel-trace --stats -- python bench.py --pop 512 --steps 2 --warmup 1 --no-cpu-baseline" > $O/kernel_stats_pop512.md 2>> $O/err.txt
python tools/pmc_summary.py $O/pmc_fetch > $O/pmc_fetch.json 2>> $O/err.txt
python tools/pmc_summary.py $O/pmc_write > $O/pmc_write.json 2>> $O/err.txt
python tools/pmc_summary.py $O/pmc_sq > $O/pmc_sq.json 2>> $O/err.txt
SERL_PROFILE=1 timeout 300 python tools/ab.py 150 192 > $O/ab.txt 2>> $O/err.txt
[ -f serl_amd/csrc/libserl_amd_prof.so ] && SERL_PROFILE=1 SERL_LIB=$R/serl_amd/csrc/libserl_amd_prof.so timeout 300 python tools/ab.py 150 > $O/ab_prof1.txt 2>> $O/err.txt
[ -f serl_amd/csrc/libserl_amd_prof2.so ] && SERL_PROFILE=1 SERL_LIB=$R/serl_amd/csrc/libserl_amd_prof2.so timeout 300 python tools/ab.py 150 > $O/ab_prof2.txt 2>> $O/err.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/valu_latency.hip -o /tmp/valu_latency 2>> $O/err.txt && timeout 120 /tmp/valu_latency > $O/valu_latency.json 2>> $O/err.txt
python tools/dag/critical_path.py nominal --latency $O/valu_latency.json > $O/critical_path.json 2>> $O/err.txt
rm -rf $O/prof $O/prof_serl10 $O/prof_serl10p128 $O/prof_pop512 $O/pmc_fetch $O/pmc_write $O/pmc_sq     # (the databases are large; their summaries stay)
ls -la $O
