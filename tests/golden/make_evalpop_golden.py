#!/usr/bin/env python3
"""Golden vectors of the `-eval_pop` path from the REFERENCE'S OWN base/evaluate.py (evaluate :59-120, validate_agent
:123-150, the eval_pop loop :236-256), base/evaluation_utils.py gen_refs (:23-55) and base/core/utils.py calc_nMAE /
calc_smoothness -- imported unmodified (build container only) -> tests/golden/evalpop.npz

base/evaluate.py parses its command line, builds its global env at import time and imports two packages that are not
installed (`toml`, `plotters`): it is imported here with sys.argv set to `-agent_name x`, stand-ins for those two
(never called) and refshim's gym / signals / _citation shims.  Then, in main()'s order with its seed 7:
gen_refs(theta), gen_refs(phi), + the base reference (:173-186); validate_agent for four shipped SERL50 actors on ONE
global env (the tracking error carried over reset() from episode to episode, as in the reference run).

  times/amps_theta/amps_phi   the reference tuples (random one first, base last), as SmoothedStepSequence parameters
  nmae, smoothness [R, pop]    per (reference, actor), recorded from evaluate()'s return values
  stats [pop, 4]               Stats(nmae, nmae_sd, sm, sm_sd) of validate_agent
  champion                     index the eval_pop loop keeps (first strict minimum of stats.nmae)
  err0 [pop*R, 3]              env.error at the moment each episode's reset() ran (what obs0 carried)
  data_<i>                     every 50th row of the last episode's data[T, 19] (ref3, u3, x12, reward) of actor i
"""
import os, sys, types, io, contextlib
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refshim
os.chdir('/tmp')
refshim.install()
import make_golden as MG

ACTORS = (18, 0, 7, 33)


def import_reference_evaluate():
    for name in ('toml', 'plotters', 'plotters.plot_utils'):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules['plotters.plot_utils'].plot = lambda *a, **k: (None, None)
    sys.modules['toml'].TomlNumpyEncoder = object
    argv = sys.argv
    sys.argv = ['evaluate.py', '-agent_name', 'x']
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            import evaluate as ev
    finally:
        sys.argv = argv
    return ev


def main():
    ev = import_reference_evaluate()
    from evaluation_utils import gen_refs
    import signals
    t_max = 80
    time_array = np.linspace(0., t_max, 6)
    np.random.seed(7)                                   # evaluate.py main(): np.random.seed(parameters.seed), -seed default 7
    theta_refs = gen_refs(t_max, time_array, 12.0, num_trails=1)
    phi_refs = gen_refs(t_max, time_array, 10.0, num_trails=1)
    theta_refs.append(signals.SmoothedStepSequence(time_array, [0, 12, 3, -4, -8, 2], smooth_width=t_max // 10))
    phi_refs.append(signals.SmoothedStepSequence(time_array, [2, -2, 2, 10, 2, -6], smooth_width=t_max // 10))
    user_eval_refs = list(zip(theta_refs, phi_refs))

    sds, h, act = MG.load_pop('serl50')
    pop = [types.SimpleNamespace(actor=refshim.make_actor(sds[i], h, 3, act)) for i in ACTORS]
    rec, err0 = [], []
    orig = ev.evaluate

    def recording_evaluate(actor, **kw):
        err0.append(np.array(ev.env.error, dtype=np.float64).copy() if ev.env.error is not None else np.zeros(3))
        data, nmae, sm = orig(actor, **kw)
        rec.append((nmae, sm))
        return data, nmae, sm
    ev.evaluate = recording_evaluate
    stats, datas = [], []
    nmae_min, champion = 500, None
    for i, agent in enumerate(pop):                     # evaluate.py:243-256
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            data, st = ev.validate_agent(agent, user_eval_refs, 1)
        stats.append([st.nmae, st.nmae_sd, st.sm, st.sm_sd])
        datas.append(data)
        if st.nmae < nmae_min:
            nmae_min, champion = st.nmae, i
        print(i, stats[-1], flush=True)
    R = len(user_eval_refs)
    rec = np.array(rec).reshape(len(pop), R, 2)
    out = dict(actors=np.array(ACTORS), times=np.array([np.asarray(r.times, float) for r in theta_refs]),
               times_phi=np.array([np.asarray(r.times, float) for r in phi_refs]),
               amps_theta=np.array([np.asarray(r.amps, float) for r in theta_refs]),
               amps_phi=np.array([np.asarray(r.amps, float) for r in phi_refs]),
               nmae=rec[:, :, 0].T.copy(), smoothness=rec[:, :, 1].T.copy(), stats=np.array(stats), champion=np.array(champion),
               err0=np.array(err0))
    for i, d in enumerate(datas):
        out['data_%d' % i] = d[::50]
        out['data_len_%d' % i] = np.array(len(d))
    np.savez_compressed(os.path.join(HERE, 'evalpop.npz'), **out)
    print('champion', champion, 'nmae', out['nmae'])


if __name__ == '__main__':
    main()
