#!/bin/bash
# round 4, session g: (1) one forward pass over two actor wavefronts (SERL10 / TD3 shapes, one episode per team): parity tests + timing;
# (2) random search of the balancer knobs on top of the short libm (tools/sweeps/r04g_balancer_random.json)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04g
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_rollout.py -x -q -m gpu -k "streamed or handover or first_launch or full_size or population_fitness" > $O/pytest_split.txt 2>&1
tail -15 $O/pytest_split.txt
for a in serl10 td3; do
  AB_ACTORS=$a SERL_LIB=serl_amd/csrc/libserl_amd_r03.so timeout 200 python tools/ab.py 30 >> $O/ab_split.txt 2>> $O/err.txt
  AB_ACTORS=$a timeout 200 python tools/ab.py 30 >> $O/ab_split.txt 2>> $O/err.txt
done
cat $O/ab_split.txt
REPS=1 bash tools/sweep_run.sh r04g_sweep > $O/sweep_print.txt 2>&1
sort -t: -k4 -n $O/sweep_print.txt | head -50
