#!/bin/bash
# round 4, session h: the split actor after the barrier-credit fix (a short guarded run first), new unit tests, A/B of the re-tuned balancer
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04h
mkdir -p $O
cd $R
AB_ACTORS=serl10 timeout 90 python tools/ab.py 30 > $O/ab_split.txt 2>> $O/err.txt
if [ $? -ne 0 ]; then echo "split actor run failed or hung: skipping everything that uses it"; tail -3 $O/err.txt; SKIP="not streamed and not first_launch and not population_fitness and not handover and not size_classes"; else
  AB_ACTORS=td3 timeout 90 python tools/ab.py 30 >> $O/ab_split.txt 2>> $O/err.txt
  AB_ACTORS=serl10 timeout 90 python tools/ab.py 384 >> $O/ab_split.txt 2>> $O/err.txt
  SKIP="streamed or handover or first_launch or population_fitness or size_classes or constant_division or short_libm or full_size"
fi
cat $O/ab_split.txt
timeout 1500 python -m pytest tests/test_gpu_rollout.py -x -q -m gpu --timeout=400 -k "$SKIP" > $O/pytest_h.txt 2>&1
tail -8 $O/pytest_h.txt
for rep in 1 2; do
  for t in r03 "" exp_h12 exp_h31 exp_h06 exp_hnotan; do
    L=serl_amd/csrc/libserl_amd${t:+_$t}.so
    SERL_LIB=$L timeout 200 python tools/ab.py 150 384 1023 >> $O/ab.txt 2>> $O/err.txt
  done
done
cat $O/ab.txt | sed 's/.*libserl_amd//' | cut -c1-200
