#!/usr/bin/env python3
"""Build a set of team-kernel variants for an A/B session on the GPU box (tools/exp_build.py per variant, in parallel):

  python tools/sweep_build.py <sweep.json>        sweep.json: {"tag": {"env": {...generator knobs...}, "flags": ["-D..."]}, ...}

writes gen/citation_nominal_team_<tag>.inc (git-ignored: *_exp*.inc) + serl_amd/csrc/libserl_amd_<tag>.so for every tag and
gpurun_out/sweep_libs.txt (one tag per line) for tools/sweep_run.sh."""
import json, os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = json.load(open(sys.argv[1]))
sys.path.insert(0, ROOT)
from serl_amd import build as B
B.build()


def one(item):
    tag, cfg = item
    env = dict(os.environ, **{k: str(v) for k, v in cfg.get('env', {}).items()})
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'dag', 'codegen_team.py'), 'nominal', '--suffix=_' + tag] + cfg.get('gen_args', []),
                       env=env, capture_output=True, text=True)
    if r.returncode:
        return tag, 'codegen failed: ' + r.stderr[-2000:]
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'exp_build.py'), tag] + cfg.get('flags', []), env=env, capture_output=True, text=True)
    return tag, (r.stdout.strip().splitlines()[-1] if r.returncode == 0 else 'build failed: ' + (r.stderr + r.stdout)[-2000:])


with ThreadPoolExecutor(max_workers=int(os.environ.get('SWEEP_JOBS', 8))) as ex:
    res = list(ex.map(one, spec.items()))
ok = [t for t, m in res if m.endswith('.so')]
for t, m in res:
    print(t, m)
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
open(os.path.join(ROOT, 'tools', 'sweep_libs.txt'), 'w').write('\n'.join(ok) + '\n')
