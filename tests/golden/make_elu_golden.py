#!/usr/bin/env python3
"""Golden vectors for ELU actors (base/core/mod_utils.py:14-18: activations['elu'] = nn.ELU()), which none of the shipped
checkpoints uses: the weights of shipped SERL50 actors are loaded into the REFERENCE'S OWN Actor built with
activation_actor='elu' and flown by its own Agent.evaluate (build container only) -> tests/golden/elu.npz

  obs_samples [64, 7], act_samples [n, 64, 3]   torch forward of the ELU actors on random observations
  ret [n, 4]                                    fitness, length, smoothness, steps of a 20 s episode, base reference
  actions_<i> [T, 3]                            env.last_u of every step
"""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refshim
os.chdir('/tmp')
refshim.install()
import make_golden as MG

ACTORS = (18, 0, 7)


def main():
    sds, h, _ = MG.load_pop('serl50')
    th, ph = MG.base_refs(20)
    env = refshim.make_env('nominal', 20)
    rng = np.random.default_rng(0)
    obs = np.concatenate([rng.normal(0, 0.05, (64, 3)), rng.normal(0, 0.1, (64, 4))], axis=1)
    out = dict(obs_samples=obs, actors=np.array(ACTORS))
    acts, ret = [], []
    for i in ACTORS:
        actor = refshim.make_actor(sds[i], h, 3, 'elu')
        acts.append(np.stack([actor.select_action(o) for o in obs]))
        ep = MG.run_ref(env, actor, th, ph)
        ret.append([ep.fitness, ep.length, ep.smoothness, len(ep.reward_lst)])
        out['actions_%d' % i] = np.asarray(ep.actions)
        print(i, ret[-1], flush=True)
    out['act_samples'], out['ret'] = np.stack(acts), np.array(ret)
    np.savez_compressed(os.path.join(HERE, 'elu.npz'), **out)


if __name__ == '__main__':
    main()
