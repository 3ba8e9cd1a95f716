#!/bin/bash
# regenerate every committed generated kernel source (serl_amd/csrc/gen): run after any change of tools/dag/codegen*.py
set -e
cd "$(dirname "$0")/.."
V="nominal ice cg_timed gust test"
python tools/dag/codegen.py $V
python tools/dag/codegen_team.py $V
python tools/dag/codegen_team.py $V --lane-groups        # gen/citation_<v>_teamg.inc: the partition of the two / four-episodes-per-team kernels
python tools/dag/codegen_team.py $V --waves=6 --suffix=6
python tools/dag/codegen_lane.py $V
