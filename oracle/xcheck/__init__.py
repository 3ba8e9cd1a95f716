"""TEST INFRASTRUCTURE -- GPU kernels compiled from the LIFTED model source (oracle/gen), loaded from
oracle/_build/libserl_xcheck.so.  A second, independently derived GPU implementation of the dynamics (the product
ships what tools/dag generates from the DAG); tests/test_gpu_rollout.py compares the two bit for bit.  Only tests/
may import this module."""
import ctypes, os, subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(os.path.dirname(HERE), '_build', 'libserl_xcheck.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
UNITS = ['xcheck_capi.hip', 'xcheck_nominal.hip', 'xcheck_ice.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-Wno-unused-value',
         '-I', os.path.join(ROOT, 'serl_amd', 'csrc'), '-I', HERE]


def _deps():
    out = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(('.hip', '.inc'))]
    out += [os.path.join(ROOT, 'oracle', 'gen', 'citation_%s.inc' % v) for v in ('nominal', 'ice')]
    out += [os.path.join(ROOT, 'serl_amd', 'csrc', f) for f in ('citation_dev.h', 'citation_leaves.h', 'citation_step_dev.h', 'citation_libm.h',
                                                                'rollout_device.h', 'rollout_variant.inc')]
    out.append(os.path.join(ROOT, 'include', 'serl_amd.h'))
    return out


def build(force=False):
    """hipcc cross-compiles for gfx950 without a GPU (build container); the .so travels to the GPU box with the snapshot."""
    if not force and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in _deps()):
        return LIB
    objdir = os.path.join(os.path.dirname(LIB), 'xcheck_obj')
    os.makedirs(objdir, exist_ok=True)

    def cc(unit):
        obj = os.path.join(objdir, unit.replace('.hip', '.o'))
        r = subprocess.run([HIPCC] + FLAGS + ['-c', os.path.join(HERE, unit), '-o', obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed for %s:\n%s' % (unit, r.stderr[-4000:]))
        return obj

    with ThreadPoolExecutor(max_workers=len(UNITS)) as ex:
        objs = list(ex.map(cc, UNITS))
    r = subprocess.run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('link failed:\n' + r.stderr[-4000:])
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            raise RuntimeError('%s not built (python -c "import __graft_entry__ as g; g.build()")' % LIB)
        L = ctypes.CDLL(LIB)
        VP, i32 = ctypes.c_void_p, ctypes.c_int32
        L.serl_xcheck_last_error.restype = ctypes.c_char_p
        L.serl_xcheck_rollout.argtypes = [ctypes.c_int, VP, ctypes.c_int, ctypes.c_double, VP, ctypes.c_int, VP]
        L.serl_xcheck_dyn_open_loop.argtypes = [ctypes.c_int, VP, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int, VP, VP,
                                               ctypes.c_int, VP]
        L.serl_xcheck_div_const.argtypes = [VP, ctypes.c_int, ctypes.c_double, ctypes.c_double, VP, VP, VP]
        L.serl_xcheck_libm.argtypes = [ctypes.c_int, VP, ctypes.c_int, ctypes.c_double, VP, VP, VP]
        L.serl_xcheck_act.argtypes = [ctypes.c_int, VP, ctypes.c_int, VP, VP]
        _lib = L
    return _lib


def div_const(x, c, rc):
    """(x / c by the product's reciprocal + fma correction, x / c by the IEEE division) on the GPU, as numpy arrays"""
    import torch
    xt = torch.as_tensor(x, dtype=torch.float64).cuda().contiguous()
    fast, ieee = torch.empty_like(xt), torch.empty_like(xt)
    rc_ = lib().serl_xcheck_div_const(xt.data_ptr(), xt.numel(), float(c), float(rc), fast.data_ptr(), ieee.data_ptr(), None)
    assert rc_ == 0
    torch.cuda.synchronize()
    return fast.cpu().numpy(), ieee.cpu().numpy()


def libm(kind, x, c=0.0):
    """citw_sincos (kind 0) / citw_tan (1) / citw_pow(x, c) (2) / citw_atan (3) of serl_amd/csrc/citation_libm.h on the GPU"""
    import torch
    xt = torch.as_tensor(x, dtype=torch.float64).cuda().contiguous()
    o0, o1 = torch.empty_like(xt), torch.empty_like(xt)
    rc_ = lib().serl_xcheck_libm(int(kind), xt.data_ptr(), xt.numel(), float(c), o0.data_ptr(), o1.data_ptr(), None)
    assert rc_ == 0
    torch.cuda.synchronize()
    return o0.cpu().numpy(), o1.cpu().numpy()


def activation(act, x):
    """serl_act (0 tanh / 1 ELU / 2 LeakyReLU) of serl_amd/csrc/rollout_device.h on the GPU, f32 array in, f32 array out"""
    import numpy as np, torch
    xt = torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32)).cuda().contiguous()
    y = torch.empty_like(xt)
    rc_ = lib().serl_xcheck_act(int(act), xt.data_ptr(), xt.numel(), y.data_ptr(), None)
    assert rc_ == 0
    torch.cuda.synchronize()
    return y.cpu().numpy()


def rollout(weights, spec, member_of_episode, ref, *, build='h2000_v90', t_max=80.0, lanes_per_wave=8):
    """The attitude task on the lifted-code lane kernels: same inputs as serl_amd.RolloutEngine.rollout (table references),
    returns dict of device tensors (fitness, length_steps, length_t, cost_steps)."""
    import numpy as np, torch
    from serl_amd import _capi, builds
    from serl_amd.actor import pad_rows
    dev = torch.device('cuda', 0)
    data, ent = builds.load(build)
    code = builds.CODE_IDS[ent['code']]
    blob = torch.from_numpy(np.concatenate([np.asarray(data[k], np.float64).reshape(-1) for k in ('ro', 't3', 'x0', 'dw0')])).to(dev)
    assert blob.numel() == len(data['ro']) + 46 + 19 + 31
    w = torch.as_tensor(weights, dtype=torch.float32).to(dev)
    if w.shape[1] % 4 or w.stride(0) % 4 or not w.is_contiguous():
        w = pad_rows(w)
    moe = torch.as_tensor(np.asarray(member_of_episode), dtype=torch.int32).to(dev).contiguous()
    E = moe.numel()
    ref_t = torch.as_tensor(ref, dtype=torch.float64).to(dev).contiguous()
    shared = ref_t.dim() == 2
    T = ref_t.shape[-2]
    out = dict(fitness=torch.zeros(E, dtype=torch.float64, device=dev), length_steps=torch.zeros(E, dtype=torch.int32, device=dev),
               length_t=torch.zeros(E, dtype=torch.float64, device=dev), cost_steps=torch.zeros(E, dtype=torch.int32, device=dev))
    d = _capi.RolloutDesc(state_dim=spec.state_dim, action_dim=spec.action_dim, hidden=spec.hidden, num_layers=spec.num_layers,
                          activation=spec.activation_id, n_members=w.shape[0], weights=w.data_ptr(),
                          weight_stride=w.stride(0) if w.shape[0] > 1 else w.shape[1], n_episodes=E, build_slot=0,
                          member_of_episode=moe.data_ptr(), ref=ref_t.data_ptr(), ref_stride=0 if shared else T * 3,
                          t_max=float(t_max), max_steps=T, lanes_per_wave=int(lanes_per_wave),
                          fitness=out['fitness'].data_ptr(), length_steps=out['length_steps'].data_ptr(),
                          length_t=out['length_t'].data_ptr(), cost_steps=out['cost_steps'].data_ptr())
    stream = torch.cuda.current_stream(dev).cuda_stream
    rc = lib().serl_xcheck_rollout(code, blob.data_ptr(), len(data['ro']), float(np.asarray(data['dt']).reshape(-1)[0]),
                                   ctypes.byref(d), int(lanes_per_wave), ctypes.c_void_p(stream))
    if rc != 0:
        raise RuntimeError('serl_xcheck_rollout failed (%d): %s' % (rc, lib().serl_xcheck_last_error().decode()))
    torch.cuda.synchronize(dev)
    return out
