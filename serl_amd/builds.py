"""Dynamics builds of the reference (`envs/<build>/`), their tables and the env-name grammar.

Data files (serl_amd/data/citation_<build>.npz) hold what the reference binary holds: `.rodata` as f64
(aero tables rtConstP, rtConstB, literal pool), the post-initialize() images of rtX / rtDW and the
table3 parameters; extracted by tools/lift/gen_models.py.

Mode grammar follows envs/phlabenv.py:99-172 (`PHlab_<config>_<mode>`, envs/config.py:16-25).
"""
import json, os
import numpy as np

DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data')
CODE_IDS = {'nominal': 0, 'ice': 1, 'cg_timed': 2, 'gust': 3, 'test': 4}
INF = float('inf')

# mode -> (build directory, actuator-fault row {elev_gain, elev_clip, ail_clip, jam_on, jam, 0,0,0})
NOMINAL_ROW = (1.0, INF, INF, 0.0, 0.0, 0.0, 0.0, 0.0)
MODES = {
    'nominal': ('h2000_v90', NOMINAL_ROW),
    'h2000-v90': ('h2000_v90', NOMINAL_ROW),
    'high-q': ('h2000_v150', NOMINAL_ROW), 'h2000-v150': ('h2000_v150', NOMINAL_ROW),
    'low-q': ('h10000_v90', NOMINAL_ROW), 'h10000-v90': ('h10000_v90', NOMINAL_ROW),
    'be': ('h2000_v90', (0.3, INF, INF, 0.0, 0.0, 0.0, 0.0, 0.0)),                      # envs/be/citation.py:71-75
    'jr': ('h2000_v90', (1.0, INF, INF, 1.0, 15 * 3.14159 / 180, 0.0, 0.0, 0.0)),       # envs/jr/citation.py:71-75
    'sa': ('h2000_v90', (1.0, INF, float(np.deg2rad(1)), 0.0, 0.0, 0.0, 0.0, 0.0)),     # envs/sa/citation.py:73-79
    'se': ('h2000_v90', (1.0, float(np.deg2rad(2.5)), INF, 0.0, 0.0, 0.0, 0.0, 0.0)),   # envs/se/citation.py:73-79
    'ice': ('ice', NOMINAL_ROW),
    'cg': ('cg', NOMINAL_ROW), 'cg-aft': ('cg', NOMINAL_ROW),
    'cg-for': ('cg_for', NOMINAL_ROW),
    'cg-shift': ('cg_timed', NOMINAL_ROW), 'cg-timed': ('cg_timed', NOMINAL_ROW),
    'gust': ('gust', NOMINAL_ROW),
    'noise': ('h2000_v90', NOMINAL_ROW),                # the nominal binary behind the sensor model of envs/noise/citation.py
    'test': ('test', NOMINAL_ROW),
}
# modes whose SWIG wrapper adds the sensor model to what step() returns (envs/noise/citation.py:71-82,
# envs/gust/citation.py:72-86); the evaluator feeds the kernel a pre-drawn table (sensor_noise_table)
SENSOR_NOISE_MODES = ('noise', 'gust')


def mode_key(mode):
    """Key of MODES for a mode / env name, following the if-chain of envs/phlabenv.py:99-172 in its order: the first
    branch matches 'nominal' exactly or any name CONTAINING 'h2000-v90'; the last one any name containing 'test';
    every other branch compares the lower-cased name for equality.  Env names split like envs/config.py:16-25
    ('PHlab_attitude_<mode>'; two tokens = empty mode, which the reference rejects)."""
    m = mode
    if m.lower().startswith('phlab_'):
        toks = m.lower().split('_')
        m = toks[2] if len(toks) == 3 else ''
    if 'incremental' in m.lower():          # envs/phlabenv.py:99: the nominal build (rate control: env_config below)
        return 'nominal'
    if m == 'nominal' or 'h2000-v90' in m.lower():
        return 'nominal'
    m = m.lower()
    if m not in MODES and 'test' in m:
        return 'test'
    return m


ENV_ATTITUDE, ENV_SYMMETRIC, ENV_FULL = 0, 1, 2       # serl_rollout_desc.env_config (include/serl_amd.h)


def env_config(name):
    """(env_config, incremental) of an env name `PHlab_<configuration>_<mode>` (envs/config.py:16-25) the way
    CitationEnv reads it (envs/phlabenv.py:86-97: 'symmetric' / 'attitude' as substrings of the configuration, anything
    else = full state; :175: 'incremental' as a substring of the mode).  A bare mode ('nominal', 'be', 'incremental' ...)
    means the attitude configuration."""
    n = name.lower()
    cfg, mode = 'attitude', n
    if n.startswith('phlab_') or n.startswith('ph'):
        toks = n.split('_')
        if len(toks) == 3:
            cfg, mode = toks[1], toks[2]
        else:
            cfg, mode = toks[-1], ''
    c = ENV_SYMMETRIC if 'symmetric' in cfg else (ENV_ATTITUDE if 'attitude' in cfg else ENV_FULL)
    return c, 'incremental' in mode


def env_dims(config, incremental=False):
    """(state_dim, action_dim) of a configuration: observation = [error (A), observed states, last_u (A, incremental)]
    (envs/phlabenv.py:213-220)"""
    A = 1 if config == ENV_SYMMETRIC else 3
    nx = {ENV_ATTITUDE: 4, ENV_SYMMETRIC: 1, ENV_FULL: 10}[config]
    return A + nx + (A if incremental else 0), A


def has_sensor_noise(mode):
    return mode_key(mode) in SENSOR_NOISE_MODES


def sensor_noise_table(n_steps, rng=np.random):
    """f64 [n_steps + 1, 7]: the addends of the reference's sensor model for one episode -- entry 0 for the step reset()
    takes, entry k + 1 for env step k -- for the channels p q r | alpha | beta | phi theta, drawn in the wrapper's
    order (randn(3), randn(1), randn(1), randn(2) per call of step(); one randn(n, 7) block of the same legacy
    generator yields the same stream) and combined with the wrapper's own expressions."""
    z = rng.randn(n_steps + 1, 7)
    t = np.empty_like(z)
    t[:, 0:3] = 3.0 * 10**(-5) + 6.3 * 10**(-4) * z[:, 0:3]
    t[:, 3] = 4.0 * 10**(-10) * z[:, 3]
    t[:, 4] = 1.8 * 10**(-3) + 2.7 * 10**(-4) * z[:, 4]
    t[:, 5:7] = 4.0 * 10**(-3) + 3.2 * 10**(-5) * z[:, 5:7]
    return t

def sensor_terms(z):
    """the wrapper's expressions (envs/noise/citation.py:71-82) applied to standard-normal draws z [.., 7]"""
    t = np.empty_like(z)
    t[..., 0:3] = 3.0 * 10**(-5) + 6.3 * 10**(-4) * z[..., 0:3]
    t[..., 3] = 4.0 * 10**(-10) * z[..., 3]
    t[..., 4] = 1.8 * 10**(-3) + 2.7 * 10**(-4) * z[..., 4]
    t[..., 5:7] = 4.0 * 10**(-3) + 3.2 * 10**(-5) * z[..., 5:7]
    return t


def draw_episode_noise(n_steps, action, sensor, rng=np.random, n_actions=3):
    """The np.random draws of ONE sequential reference episode, pre-drawn in the reference's interleaved order:
    reset() -> step(): sensor model 7 draws (if the mode has one); then per env step: exploration noise randn(n_actions)
    (base/core/agent.py:91 `randn(action.shape[0])`, if enabled) followed by the sensor model's 7 draws inside env.step().
    -> (z_action [n_steps, n_actions] standard normals or None, sensor table [n_steps + 1, 7] or None, resync)
    where resync(n) rewinds the generator and re-draws exactly what an episode of n steps consumes, so that a seeded run
    leaves np.random where the reference's run leaves it (an episode that ends early draws less)."""
    per = (n_actions if action else 0) + (7 if sensor else 0)
    head = 7 if sensor else 0
    if per == 0:
        return None, None, (lambda n: None)
    state = rng.get_state()
    z = rng.randn(head + n_steps * per)
    body = z[head:].reshape(n_steps, per)
    za = body[:, :n_actions].copy() if action else None
    sn = None
    if sensor:
        sn = sensor_terms(np.concatenate([z[:7][None], body[:, per - 7:]], 0))

    def resync(n):
        rng.set_state(state)
        rng.randn(head + int(n) * per)
    return za, sn, resync


_index = None
_cache = {}


def index():
    global _index
    if _index is None:
        _index = json.load(open(os.path.join(DATA_DIR, 'builds.json')))
    return _index


def load(build):
    """-> (dict of numpy arrays ro,x0,dw0,t3,dt,ro_base,nB ; index entry {data, code, nB})"""
    ent = index()[build]
    key = ent['data']
    if key not in _cache:
        z = np.load(os.path.join(DATA_DIR, 'citation_%s.npz' % key))
        _cache[key] = {k: np.ascontiguousarray(z[k]) for k in z.files}
    return _cache[key], ent


def resolve_mode(mode):
    """'nominal' | 'be' | 'PHlab_attitude_ice' ... -> (build, fault_row)"""
    m = mode_key(mode)
    if m not in MODES:
        raise ValueError('unknown PH-LAB mode %r (known: %s)' % (mode, ', '.join(sorted(MODES))))
    return MODES[m]
