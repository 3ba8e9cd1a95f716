"""Shims that let the reference's OWN, UNMODIFIED Python (envs/phlabenv.py CitationEnv, the SWIG
wrappers envs/<build>/citation.py with their fault code, base/core/agent.py Agent.evaluate,
base/core/genetic_agent.py Actor) run under this container's Python 3.10 / torch 2.10:

  * `gym`      -- minimal stand-in (Env, Wrapper, spaces.Box, wrappers); gym 0.17 is not installed
  * `signals`  -- oracle.signals (un-vendored dependency)
  * `envs.<build>._citation` -- ctypes-backed module around the reference's own shared object
                  (the cp38 SWIG extension cannot be imported by 3.10; its raw C entry points can)

Used only by tests/golden/make_golden.py (golden-vector generation in the build container) and by
the tests marked `needs_reference`.  Nothing here is imported by the product.
"""
import sys, types, os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from oracle import refso as _refso
# a directory (/root/reference, $SERL_REFERENCE) or the archive staged by oracle/build.py stage_ref() (zipimport)
REF = _refso.REF


def install(use_oracle_dynamics=False):
    """Install the shims.  use_oracle_dynamics=True swaps the reference .so for the C restatement
    (lets the reference Python run where /root/reference/envs/*.so cannot be loaded)."""
    sys.dont_write_bytecode = True
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from oracle import signals as osig
    # ---- gym ----
    gym = types.ModuleType('gym')

    class Env:
        def seed(self, seed=None):
            return [seed]

    class Wrapper(Env):
        def __init__(self, env):
            self.env = env

    class Box:
        def __init__(self, low, high, shape=None, dtype=np.float64):
            self.low = np.asarray(low, dtype=dtype)
            self.high = np.asarray(high, dtype=dtype)
            self.shape = self.low.shape
            self.dtype = dtype

    def make(*a, **k):
        raise RuntimeError('gym.make is not available in the shim')

    spaces = types.ModuleType('gym.spaces')
    spaces.Box = Box
    gym.Env, gym.Wrapper, gym.make, gym.spaces = Env, Wrapper, make, spaces
    gym.wrappers = types.ModuleType('gym.wrappers')
    sys.modules.update({'gym': gym, 'gym.spaces': spaces, 'gym.wrappers': gym.wrappers})
    # ---- signals ----
    sig = types.ModuleType('signals')
    for n in ('BaseSignal', 'Const', 'SmoothedStepSequence'):
        setattr(sig, n, getattr(osig, n))
    st = types.ModuleType('signals.stochastic_signals')
    st.RandomizedCosineStepSequence = osig.RandomizedCosineStepSequence
    sig.stochastic_signals = st
    sys.modules.update({'signals': sig, 'signals.stochastic_signals': st})
    # ---- low-level _citation modules ----
    for p in (REF, os.path.join(REF, 'base')):
        if p not in sys.path:
            sys.path.insert(0, p)
    from oracle import refso, dynamics
    import importlib
    for build in ('h2000_v90', 'h2000_v150', 'h10000_v90', 'be', 'jr', 'sa', 'se', 'ice', 'noise', 'cg',
                  'cg_for', 'cg_timed', 'gust', 'test'):
        name = 'envs.%s._citation' % build
        if name in sys.modules:
            continue
        m = types.ModuleType(name)
        m._sim = None
        m._build = build

        def _get(m=m):
            if m._sim is None:
                m._sim = dynamics.CitationDynamics(m._build) if use_oracle_dynamics else refso.RefCitation(m._build)
            return m._sim

        m.initialize = lambda m=m, _get=_get: _get().initialize()
        m.step = lambda cmd, m=m, _get=_get: _get().step(np.asarray(cmd, dtype=np.float64))
        m.terminate = lambda m=m: None
        sys.modules[name] = m
        pkg = importlib.import_module('envs.%s' % build)
        pkg._citation = m
    sys.path[:] = [str(p) for p in sys.path]      # (the reference's envs/__init__.py appends pathlib objects)


def make_env(mode='nominal', t_max=80):
    """The reference's CitationEnv in attitude configuration, eval mode (base/evaluate.py:55-56)."""
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        from envs.phlabenv import CitationEnv
        env = CitationEnv(configuration='attitude', mode=mode)
        env.set_eval_mode(t_max=t_max)
    return env


def make_actor(state_dict, hidden, num_layers, activation):
    """The reference's torch Actor (base/core/genetic_agent.py:69) holding a shipped state_dict."""
    import argparse, torch
    from core.genetic_agent import Actor
    args = argparse.Namespace(hidden_size=hidden, num_layers=num_layers, activation_actor=activation,
                              state_dim=7, action_dim=3, device=torch.device('cpu'))
    a = Actor(args)
    a.load_state_dict(state_dict)
    a.eval()
    return a


def reference_evaluate(env, actor, user_refs=None, smooth_fitness=False):
    """Call the reference's own Agent.evaluate (base/core/agent.py:63-138) on a minimal `self`."""
    import argparse
    from core.agent import Agent
    fake = argparse.Namespace()
    fake.args = argparse.Namespace(smooth_fitness=smooth_fitness, noise_sd=0.0, noise_clip=0.0)
    if user_refs is not None:
        class _Env:  # forwards reset() with the user refs, everything else untouched
            def __init__(self, e):
                self.__dict__['_e'] = e

            def reset(self):
                return self._e.reset(user_refs=user_refs)

            def __getattr__(self, k):
                return getattr(self._e, k)
        fake.env = _Env(env)
    else:
        fake.env = env
    agent = argparse.Namespace(actor=actor)
    return Agent.evaluate(fake, agent, False, False)
