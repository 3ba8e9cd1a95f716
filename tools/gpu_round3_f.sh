#!/bin/bash
# round 3, session f: GPU tests of the final tree, A/B lines for one / two / four episodes per team, the streamed-actor teams
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03f
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -n 3 $O/pytest.log
timeout 200 python tools/ab.py 150 384 1023 > $O/ab.txt 2>> $O/err.txt
AB_ACTORS=serl10 timeout 200 python tools/ab.py 30 384 >> $O/ab.txt 2>> $O/err.txt
AB_ACTORS=td3 timeout 200 python tools/ab.py 30 >> $O/ab.txt 2>> $O/err.txt
cat $O/ab.txt | cut -c1-300
