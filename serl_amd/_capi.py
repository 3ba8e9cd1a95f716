"""ctypes loader of the C-ABI HIP extension (include/serl_amd.h).  Fails loudly when the library is
missing: the product has no CPU path."""
import ctypes, os

_D = ctypes.POINTER(ctypes.c_double)
_F = ctypes.POINTER(ctypes.c_float)
_I = ctypes.POINTER(ctypes.c_int32)
VP = ctypes.c_void_p
LIB_PATH = os.environ.get('SERL_LIB') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc', 'libserl_amd.so')

EXPORTS = ['serl_abi_version', 'serl_last_error', 'serl_param_count', 'serl_ctx_create', 'serl_ctx_destroy',
           'serl_ctx_load_build', 'serl_rollout', 'serl_rollout_multi', 'serl_dyn_open_loop', 'serl_debug_profile', 'serl_debug_mixed_placement', 'serl_last_rollout_ms', 'serl_last_rollout_info', 'serl_ga_clone', 'serl_ga_crossover',
           'serl_ga_mutate', 'serl_ga_scaled_perturb', 'serl_abi_layout', 'serl_ga_sensitivity', 'serl_ga_novelty',
           'serl_replay_scatter', 'serl_env_state_dim', 'serl_env_action_dim',
           'serl_smoothness', 'serl_smoothness_work_size', 'serl_ga_distill', 'serl_host_sample_slots']


class BuildDesc(ctypes.Structure):
    _fields_ = [('code', ctypes.c_int32), ('n_ro', ctypes.c_int32), ('ro_base', ctypes.c_uint64),
                ('ro', VP), ('t3', VP), ('x0', VP), ('dw0', VP), ('dt', ctypes.c_double)]


class RolloutDesc(ctypes.Structure):
    _fields_ = [('state_dim', ctypes.c_int32), ('action_dim', ctypes.c_int32), ('hidden', ctypes.c_int32),
                ('num_layers', ctypes.c_int32), ('activation', ctypes.c_int32), ('n_members', ctypes.c_int32),
                ('weights', VP), ('weight_stride', ctypes.c_int64),
                ('n_episodes', ctypes.c_int32), ('build_slot', ctypes.c_int32),
                ('member_of_episode', VP), ('faults', VP), ('ref', VP), ('ref_stride', ctypes.c_int64),
                ('err0', VP), ('action_noise', VP), ('noise_row', VP), ('sensor_noise', VP), ('sensor_row', VP), ('tick0', VP), ('t_max', ctypes.c_double),
                ('max_steps', ctypes.c_int32), ('lanes_per_wave', ctypes.c_int32),
                ('concurrent_episodes', ctypes.c_int32), ('kernel_hint', ctypes.c_int32),
                ('fitness', VP), ('length_steps', VP), ('length_t', VP), ('cost_steps', VP),
                ('actions', VP), ('states', VP), ('rewards', VP), ('transitions', VP),
                ('ref_spec', VP), ('ref_spec_stride', ctypes.c_int64),
                ('env_config', ctypes.c_int32), ('incremental', ctypes.c_int32)]


class FaultRow(ctypes.Structure):
    _fields_ = [(n, ctypes.c_double) for n in ('elev_gain', 'elev_clip', 'ail_clip', 'rudder_jam_on', 'rudder_jam', 'pad0', 'pad1', 'pad2')]


class RefSpec(ctypes.Structure):
    _fields_ = [('n_theta', ctypes.c_int32), ('n_phi', ctypes.c_int32), ('w_theta', ctypes.c_double), ('w_phi', ctypes.c_double),
                ('trim_deg', ctypes.c_double), ('t_theta', ctypes.c_double * 8), ('a_theta', ctypes.c_double * 8),
                ('t_phi', ctypes.c_double * 8), ('a_phi', ctypes.c_double * 8)]


class ReplayJob(ctypes.Structure):
    _fields_ = [('ring', VP), ('capacity', ctypes.c_int32), ('position', ctypes.c_int32), ('episode', ctypes.c_int32),
                ('length', ctypes.c_int32), ('cost_only', ctypes.c_int32), ('skip', ctypes.c_int32)]


# serl_rollout_desc.kernel_hint (enum serl_kernel_hint)
KERNEL_HINTS = {None: 0, 'auto': 0, 'team': 1, 'wave': 2, 'half': 3, 'team2': 4, 'team4': 5}
ABI_VERSION = 8
# serl_last_rollout_info out[0] (enum serl_kernel_family)
FAMILIES = {0: None, 1: 'team', 2: 'teams', 3: 'teams2', 4: 'teamx', 5: 'team2', 6: 'team2s', 7: 'team4', 8: 'team4_mixed', 9: 'half', 10: 'wave', 11: 'wavex', 12: 'lane', 13: 'teamr'}


def expected_layout():
    """What serl_abi_layout() must return for these hand-written mirrors to be right."""
    return ([ctypes.sizeof(RolloutDesc)] + [getattr(RolloutDesc, n).offset for n, _ in RolloutDesc._fields_] +
            [ctypes.sizeof(BuildDesc), ctypes.sizeof(FaultRow), ctypes.sizeof(RefSpec), ctypes.sizeof(ReplayJob)])


_lib = None


class ExtensionMissing(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ExtensionMissing(
            'serl_amd HIP extension not built: %s is missing. Run `python -c "import __graft_entry__ as g; '
            'g.build()"` (or `python serl_amd/build.py`). There is no CPU fallback.' % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    L.serl_abi_version.restype = ctypes.c_int
    L.serl_last_error.restype = ctypes.c_char_p
    L.serl_param_count.argtypes = [ctypes.c_int] * 4
    L.serl_ctx_create.argtypes = [ctypes.c_int, ctypes.POINTER(VP)]
    L.serl_ctx_destroy.argtypes = [VP]
    L.serl_ctx_load_build.argtypes = [VP, ctypes.c_int, ctypes.POINTER(BuildDesc)]
    L.serl_rollout.argtypes = [VP, ctypes.POINTER(RolloutDesc), VP]
    L.serl_rollout_multi.argtypes = [VP, ctypes.c_int32, ctypes.POINTER(RolloutDesc), VP]
    L.serl_dyn_open_loop.argtypes = [VP, ctypes.c_int, ctypes.c_int32, ctypes.c_int32, VP, VP, ctypes.c_int32, ctypes.c_int32, VP]
    L.serl_debug_profile.argtypes = [VP, ctypes.POINTER(ctypes.c_ulonglong)]
    L.serl_debug_mixed_placement.argtypes = [VP, ctypes.POINTER(ctypes.c_int32)]
    L.serl_last_rollout_ms.argtypes = [VP, ctypes.POINTER(ctypes.c_float)]
    L.serl_last_rollout_info.argtypes = [VP, ctypes.POINTER(ctypes.c_int32)]
    L.serl_ga_clone.argtypes = [VP, VP, ctypes.c_int64, ctypes.c_int32, VP, VP, ctypes.c_int32, VP]
    L.serl_ga_crossover.argtypes = [VP, VP, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, VP, ctypes.c_int32, VP]
    L.serl_ga_mutate.argtypes = [VP, VP, ctypes.c_int64, ctypes.c_int32, VP, VP, VP, VP, ctypes.c_int32, VP]
    L.serl_ga_scaled_perturb.argtypes = [VP, VP, ctypes.c_int64, ctypes.c_int32, VP, VP, ctypes.c_int32, VP, VP, VP]
    L.serl_abi_layout.argtypes = [VP, ctypes.c_int32]
    i32 = ctypes.c_int32
    L.serl_ga_sensitivity.argtypes = [VP, VP, ctypes.c_int64, i32, i32, i32, i32, i32, VP, i32, VP, i32, VP, VP]
    L.serl_ga_novelty.argtypes = [VP, VP, ctypes.c_int64, i32, i32, i32, i32, i32, VP, i32, VP, VP, i32, VP, VP]
    L.serl_replay_scatter.argtypes = [VP, VP, ctypes.c_int64, VP, i32, VP]
    L.serl_smoothness.argtypes = [VP, VP, ctypes.c_int64, VP, i32, i32, ctypes.c_double, VP, VP, VP]
    L.serl_smoothness_work_size.argtypes = [i32, i32]
    L.serl_ga_distill.argtypes = [VP, VP, ctypes.c_int64, i32, i32, i32, i32, i32, i32, VP, VP, VP, i32, VP, i32, VP, VP, ctypes.c_float, VP]
    L.serl_host_sample_slots.argtypes = [VP, ctypes.c_longlong, i32, i32, i32, VP, i32]
    for f in EXPORTS:
        if f not in ('serl_last_error',):
            getattr(L, f).restype = ctypes.c_longlong if f == 'serl_host_sample_slots' else ctypes.c_int
    if L.serl_abi_version() != ABI_VERSION:
        raise RuntimeError('serl_amd: ABI version mismatch (library %d, binding %d): rebuild with `python serl_amd/build.py`'
                           % (L.serl_abi_version(), ABI_VERSION))
    want = expected_layout()
    got = (ctypes.c_int32 * len(want))()
    n = L.serl_abi_layout(got, len(want))
    if n != len(want) or list(got) != want:
        raise RuntimeError('serl_amd: struct layout of the ctypes mirrors differs from the library (serl_abi_layout): '
                           'library %s, binding %s' % (list(got)[:n], want))
    _lib = L
    return L


E_UNSUPPORTED = -3      # enum serl_status SERL_E_UNSUPPORTED


def check(rc, what):
    if rc != 0:
        raise RuntimeError('%s failed (%d): %s' % (what, rc, lib().serl_last_error().decode()))
