"""ctypes binding of the C restatement of the reference dynamics library.  TEST INFRASTRUCTURE.

Mirrors the SWIG surface of envs/<build>/citation.py (``initialize()``, ``step(cmd[10]) -> x[12]``,
``terminate()``; reference envs/h2000_v90/citation.py:65-72) as an object instead of a module-level
singleton.
"""
import ctypes, json, os
import numpy as np
from . import build as _build

_D = ctypes.POINTER(ctypes.c_double)
DATA_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'serl_amd', 'data')
CODE_IDS = {'nominal': 0, 'ice': 1, 'cg_timed': 2, 'gust': 3, 'test': 4}
_libs = {}


def lib(short_libm=False):
    """short_libm: the flavour whose sin / cos / tan / pow are the product's short bodies compiled for the CPU (oracle/citation_rt.h)"""
    if short_libm not in _libs:
        L = ctypes.CDLL(_build.build(short_libm=bool(short_libm)))
        L.cit_instance_size.restype = ctypes.c_int
        L.cit_reset.argtypes = [ctypes.c_void_p, ctypes.c_int, _D, _D, _D, _D, ctypes.c_double]
        L.cit_reset.restype = None
        L.cit_step.argtypes = [ctypes.c_void_p, _D, _D]
        L.cit_step.restype = ctypes.c_int
        L.cit_step_n.argtypes = [ctypes.c_void_p, _D, _D, ctypes.c_int]
        L.cit_step_n.restype = ctypes.c_int
        for f in ('cit_B', 'cit_X', 'cit_DW'):
            getattr(L, f).argtypes = [ctypes.c_void_p]
            getattr(L, f).restype = _D
        _libs[short_libm] = L
    return _libs[short_libm]


_index = None


def build_index():
    global _index
    if _index is None:
        _index = json.load(open(os.path.join(DATA_DIR, 'builds.json')))
    return _index


_data_cache = {}


def load_build_data(build):
    """Tables and post-initialize() state images of one reference build directory name."""
    ent = build_index()[build]
    if ent['data'] not in _data_cache:
        z = np.load(os.path.join(DATA_DIR, 'citation_%s.npz' % ent['data']))
        _data_cache[ent['data']] = {k: np.ascontiguousarray(z[k]) for k in z.files}
    return _data_cache[ent['data']], ent


class CitationDynamics:
    """One independent simulator instance (the reference allows one per loaded library image)."""

    def __init__(self, build='h2000_v90', short_libm=False):
        self.L = lib(short_libm)
        self.data, self.ent = load_build_data(build)
        self.code = CODE_IDS[self.ent['code']]
        self.buf = ctypes.create_string_buffer(self.L.cit_instance_size())
        self._out = np.zeros(12)
        self.initialize()

    def initialize(self):
        d = self.data
        self.L.cit_reset(self.buf, self.code, d['ro'].ctypes.data_as(_D), d['t3'].ctypes.data_as(_D),
                         d['x0'].ctypes.data_as(_D), d['dw0'].ctypes.data_as(_D), float(np.asarray(d['dt']).reshape(-1)[0]))

    def step(self, cmd):
        cmd = np.ascontiguousarray(cmd, dtype=np.float64)
        assert cmd.shape == (10,)
        rc = self.L.cit_step(self.buf, cmd.ctypes.data_as(_D), self._out.ctypes.data_as(_D))
        assert rc == 0
        return self._out.copy()

    def terminate(self):
        pass

    @property
    def X(self):
        return np.ctypeslib.as_array(self.L.cit_X(self.buf), (19,))

    @property
    def B(self):
        return np.ctypeslib.as_array(self.L.cit_B(self.buf), (int(self.ent['nB']),))
