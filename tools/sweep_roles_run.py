#!/usr/bin/env python3
"""On the GPU box: every role map of tools/sweep_roles.py through tools/ab.py on libserl_amd_devroles.so (150 episodes x 2 001 steps), then the
best dozen twice more; the product library before and after.  Writes gpurun_out/<tag>/roles.json and prints the ranking."""
import json, os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'roles'
EPISODES = sys.argv[2] if len(sys.argv) > 2 else '150'      # 150: one episode per team; 384 / 1023: the lane-group kernels (build them with -DSERL_DEV_ROLE_MAP=1 too)
O = os.path.join(R, 'gpurun_out', tag)
os.makedirs(O, exist_ok=True)
maps = subprocess.run([sys.executable, os.path.join(R, 'tools', 'sweep_roles.py')], capture_output=True, text=True).stdout.split()


def run(lib, m=None):
    env = dict(os.environ, SERL_LIB=os.path.join(R, 'serl_amd', 'csrc', lib))
    if m is not None:
        env['SERL_JITTER_SITES'] = '0x' + m
    r = subprocess.run([sys.executable, os.path.join(R, 'tools', 'ab.py'), EPISODES], capture_output=True, text=True, env=env, cwd=R, timeout=120)
    try:
        return json.loads(r.stdout.strip().splitlines()[-1].split(' ', 1)[1])['loop_E' + EPISODES]
    except Exception:
        return None


res = {'product': [run('libserl_amd.so')], 'maps': {}}
for m in maps:
    res['maps'][m] = [run('libserl_amd_devroles.so', m)]
rank = sorted((v[0], m) for m, v in res['maps'].items() if v[0] is not None)
for _ in range(2):
    for _, m in rank[:12]:
        res['maps'][m].append(run('libserl_amd_devroles.so', m))
    res['maps'][maps[0]].append(run('libserl_amd_devroles.so', maps[0]))
res['product'].append(run('libserl_amd.so'))
json.dump(res, open(os.path.join(O, 'roles.json'), 'w'), indent=1)
print('product', res['product'], 'shipped map on the development build', res['maps'][maps[0]])
for t, m in rank[:15]:
    print(m, res['maps'][m])
print('worst', rank[-3:])
