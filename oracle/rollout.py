"""ctypes binding of oracle/rollout_ref.c (episode + GA evaluate loop).  TEST INFRASTRUCTURE."""
import ctypes
import numpy as np
from . import dynamics as _dyn

_D = ctypes.POINTER(ctypes.c_double)
_F = ctypes.POINTER(ctypes.c_float)
_I = ctypes.POINTER(ctypes.c_int32)
ACT = {'tanh': 0, 'elu': 1, 'relu': 2}


class BuildDesc(ctypes.Structure):
    _fields_ = [('code', ctypes.c_int32), ('n_ro', ctypes.c_int32), ('ro_base', ctypes.c_uint64),
                ('ro', _D), ('t3', _D), ('x0', _D), ('dw0', _D), ('dt', ctypes.c_double)]


class RolloutDesc(ctypes.Structure):
    _fields_ = [('state_dim', ctypes.c_int32), ('action_dim', ctypes.c_int32), ('hidden', ctypes.c_int32),
                ('num_layers', ctypes.c_int32), ('activation', ctypes.c_int32), ('n_members', ctypes.c_int32),
                ('weights', _F), ('weight_stride', ctypes.c_int64),
                ('n_episodes', ctypes.c_int32), ('build_slot', ctypes.c_int32),
                ('member_of_episode', _I), ('faults', _D), ('ref', _D), ('ref_stride', ctypes.c_int64),
                ('err0', _D), ('action_noise', _D), ('noise_row', _I), ('sensor_noise', _D), ('sensor_row', _I), ('tick0', _I), ('t_max', ctypes.c_double),
                ('max_steps', ctypes.c_int32), ('lanes_per_wave', ctypes.c_int32),
                ('concurrent_episodes', ctypes.c_int32), ('pad_', ctypes.c_int32),
                ('fitness', _D), ('length_steps', _I), ('length_t', _D), ('cost_steps', _I),
                ('actions', _D), ('states', _D), ('rewards', _D), ('transitions', _F),
                ('ref_spec', ctypes.c_void_p), ('ref_spec_stride', ctypes.c_int64),
                ('env_config', ctypes.c_int32), ('incremental', ctypes.c_int32)]


def _n_steps_for(t_max, dt=0.01):
    """steps of a full-length episode: first k with the accumulated t_k >= t_max, inclusive (envs/phlabenv.py:391-399,473)"""
    acc, k = 0.0, 0
    while True:
        k += 1
        if acc >= t_max:
            return k
        acc += dt


def param_count(S, H, L, A):
    return H * S + H + L * (H * H + 3 * H) + A * H + A


def pack_state_dict(sd):
    """Flatten a reference Actor state_dict (keys net.0.weight ... in module order) to f32 [P]."""
    return np.concatenate([np.asarray(v, dtype=np.float32).reshape(-1) for v in sd.values()])


FAULT_ROWS = {  # envs/{be,jr,sa,se}/citation.py:71-79
    'nominal': [1.0, np.inf, np.inf, 0.0, 0.0, 0, 0, 0],
    'be': [0.3, np.inf, np.inf, 0.0, 0.0, 0, 0, 0],
    'jr': [1.0, np.inf, np.inf, 1.0, 15 * 3.14159 / 180, 0, 0, 0],
    'sa': [1.0, np.inf, float(np.deg2rad(1)), 0.0, 0.0, 0, 0, 0],
    'se': [1.0, float(np.deg2rad(2.5)), np.inf, 0.0, 0.0, 0, 0, 0],
}


def make_build_desc(build):
    data, ent = _dyn.load_build_data(build)
    bd = BuildDesc(code=_dyn.CODE_IDS[ent['code']], n_ro=len(data['ro']), ro_base=int(np.asarray(data['ro_base']).reshape(-1)[0]),
                   ro=data['ro'].ctypes.data_as(_D), t3=data['t3'].ctypes.data_as(_D),
                   x0=data['x0'].ctypes.data_as(_D), dw0=data['dw0'].ctypes.data_as(_D), dt=float(np.asarray(data['dt']).reshape(-1)[0]))
    bd._keep = data
    return bd


def activation(act, x):
    """The actor's activation (include/serl_amd.h: det_tanhf / det_expm1f_neg / LeakyReLU) of the C restatement on an f32 array"""
    L = _dyn.lib()
    L.serl_oracle_act.argtypes = [ctypes.c_int, _F, ctypes.c_int, _F]
    L.serl_oracle_act.restype = None
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty_like(x)
    L.serl_oracle_act(ACT[act] if isinstance(act, str) else int(act), x.ctypes.data_as(_F), len(x), y.ctypes.data_as(_F))
    return y


def rollout(weights, net, member_of_episode, ref, *, build='h2000_v90', faults=None, err0=None,
            tick0=None, action_noise=None, noise_row=None, sensor_noise=None, sensor_row=None, t_max=80.0, traces=False, transitions=False, threads=1,
            env_config=0, incremental=False, short_libm=False):
    """Run episodes on the CPU oracle (short_libm: the flavour that shares sin / cos / tan / pow with the kernels -- oracle/citation_rt.h).

    weights [n_members, P] f32 (packed state_dict order); net = dict(state_dim, action_dim, hidden,
    num_layers, activation); ref [n_ep, T, 3] or [T, 3] f64 radians; faults: list of fault names or
    [n_ep, 8] rows; env_config 0 / 1 / 2 = attitude / symmetric / full, incremental = rate control
    (envs/phlabenv.py:84-97,174-176).  Returns dict of numpy arrays."""
    L = _dyn.lib(short_libm)
    L.serl_oracle_rollout.argtypes = [ctypes.POINTER(BuildDesc), ctypes.POINTER(RolloutDesc), ctypes.c_int]
    L.serl_oracle_rollout.restype = ctypes.c_int
    weights = np.ascontiguousarray(weights, dtype=np.float32)
    moe = np.ascontiguousarray(member_of_episode, dtype=np.int32)
    n_ep = len(moe)
    spec = None
    if isinstance(ref, np.ndarray) and ref.dtype.names:         # serl_ref_spec rows: the reference is generated, not read
        spec = np.ascontiguousarray(ref)
        assert spec.dtype.itemsize == 288 and len(spec) in (1, n_ep)
        T = _n_steps_for(t_max)
        ref, shared = np.zeros((1, 3)), True
    else:
        ref = np.ascontiguousarray(ref, dtype=np.float64)
        shared = ref.ndim == 2
        T = ref.shape[-2]
    P = param_count(net['state_dim'], net['hidden'], net['num_layers'], net['action_dim'])
    assert weights.shape[1] >= P
    out = dict(fitness=np.zeros(n_ep), length_steps=np.zeros(n_ep, np.int32), length_t=np.zeros(n_ep),
               cost_steps=np.zeros(n_ep, np.int32))
    d = RolloutDesc(state_dim=net['state_dim'], action_dim=net['action_dim'], hidden=net['hidden'],
                    num_layers=net['num_layers'], activation=ACT[net['activation']] if isinstance(net['activation'], str) else net['activation'],
                    n_members=weights.shape[0], weights=weights.ctypes.data_as(_F), weight_stride=weights.shape[1],
                    n_episodes=n_ep, build_slot=0, member_of_episode=moe.ctypes.data_as(_I),
                    ref=ref.ctypes.data_as(_D), ref_stride=0 if shared else T * 3, t_max=float(t_max),
                    max_steps=T, lanes_per_wave=0, env_config=int(env_config), incremental=int(bool(incremental)))
    keep = [weights, moe, ref]
    if spec is not None:
        d.ref = None
        d.ref_spec, d.ref_spec_stride = spec.ctypes.data, (0 if len(spec) == 1 else 1)
        keep.append(spec)
    if faults is not None:
        if len(faults) and isinstance(faults[0], str):
            faults = [FAULT_ROWS[f] for f in faults]
        fr = np.ascontiguousarray(faults, dtype=np.float64).reshape(n_ep, 8)
        d.faults = fr.ctypes.data_as(_D); keep.append(fr)
    if err0 is not None:
        e0 = np.ascontiguousarray(err0, dtype=np.float64).reshape(n_ep, 3)
        d.err0 = e0.ctypes.data_as(_D); keep.append(e0)
    if tick0 is not None:
        tk = np.ascontiguousarray(tick0, dtype=np.int32).reshape(n_ep)
        d.tick0 = tk.ctypes.data_as(_I); keep.append(tk)
    if sensor_noise is not None:
        sn = np.ascontiguousarray(sensor_noise, dtype=np.float64).reshape(-1, T + 1, 7)
        d.sensor_noise = sn.ctypes.data_as(_D); keep.append(sn)
        if sensor_row is not None:
            sr = np.ascontiguousarray(sensor_row, dtype=np.int32).reshape(n_ep)
            assert sr.max() < sn.shape[0]
            d.sensor_row = sr.ctypes.data_as(_I); keep.append(sr)
        else:
            assert sn.shape[0] == n_ep
    if action_noise is not None:
        an = np.ascontiguousarray(action_noise, dtype=np.float64).reshape(-1, T, 3)
        d.action_noise = an.ctypes.data_as(_D); keep.append(an)
        if noise_row is not None:
            nr = np.ascontiguousarray(noise_row, dtype=np.int32).reshape(n_ep)
            assert nr.max() < an.shape[0]
            d.noise_row = nr.ctypes.data_as(_I); keep.append(nr)
        else:
            assert an.shape[0] == n_ep
    d.fitness = out['fitness'].ctypes.data_as(_D)
    d.length_steps = out['length_steps'].ctypes.data_as(_I)
    d.length_t = out['length_t'].ctypes.data_as(_D)
    d.cost_steps = out['cost_steps'].ctypes.data_as(_I)
    if traces:
        out['actions'] = np.zeros((n_ep, T, 3)); out['states'] = np.zeros((n_ep, T, 12)); out['rewards'] = np.zeros((n_ep, T))
        d.actions = out['actions'].ctypes.data_as(_D); d.states = out['states'].ctypes.data_as(_D)
        d.rewards = out['rewards'].ctypes.data_as(_D)
    if transitions:
        out['transitions'] = np.zeros((n_ep, T, 2 * net['state_dim'] + net['action_dim'] + 3), np.float32)
        d.transitions = out['transitions'].ctypes.data_as(_F)
    bd = make_build_desc(build)
    rc = L.serl_oracle_rollout(ctypes.byref(bd), ctypes.byref(d), int(threads))
    if rc:
        raise RuntimeError('serl_oracle_rollout failed: %d' % rc)
    return out
