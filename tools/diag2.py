import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import serl_amd
from serl_amd import refsignals
from oracle import rollout as R, dynamics
np.set_printoptions(precision=17, linewidth=200)
NET32 = dict(state_dim=7, action_dim=3, hidden=32, num_layers=3, activation='tanh')
w = np.load('tests/golden/actors.npz')['serl50']
moe = np.array([35], dtype=np.int32)
ref = refsignals.synthetic_reference_tables(150, 3, 80, seed=7)[106:107]
o = R.rollout(w, NET32, moe, ref, t_max=80, threads=1, traces=True)
eng = serl_amd.RolloutEngine(0)
spec = serl_amd.NetSpec(7, 3, 32, 3, 'tanh')
out = eng.rollout(torch.from_numpy(w), spec, moe, ref, t_max=80, traces=True)
S = out['states'].cpu().numpy()[0]; A = out['actions'].cpu().numpy()[0]
for k in (0, 1, 2, 3):
    print('k', k, 'gpu x', S[k]); print('     ora x', o['states'][0, k]); print('     diff', S[k] - o['states'][0, k]); print('     act diff', A[k] - o['actions'][0, k])
# open loop: oracle's action sequence through both dynamics
T = 3000
cmds = np.zeros((1, T, 10)); cmds[0, 1:, :3] = o['actions'][0, :T - 1]
st = eng.dynamics_open_loop(cmds).cpu().numpy()[0]
d = dynamics.CitationDynamics('h2000_v90')
so = np.array([d.step(cmds[0, k]) for k in range(T)])
print('oracle open loop == oracle closed loop states:', np.abs(so[1:T] - o['states'][0, :T - 1]).max())
for k in (1, 2, 3, 10, 100, 1000, 2999):
    print('open loop k', k, 'abs diff', np.abs(st[k] - so[k]))
