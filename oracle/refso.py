"""ctypes binding of the REFERENCE's own prebuilt dynamics library.  TEST INFRASTRUCTURE, and only
usable where /root/reference exists (this build container, not the GPU box).

The cp38 SWIG module cannot be imported by Python 3.10, but ``ctypes.CDLL`` resolves its CPython
symbols against the running interpreter and the raw C entry points ``initialize`` /
``step(double*, double*)`` work.  State is file-scope inside the library, so every
independent instance needs its own copy of the file.
"""
import ctypes, os, shutil, tempfile
import numpy as np

REF = os.environ.get('SERL_REFERENCE', '/root/reference')
_D = ctypes.POINTER(ctypes.c_double)


def available():
    return os.path.isdir(os.path.join(REF, 'envs', 'h2000_v90'))


def so_path(build):
    return os.path.join(REF, 'envs', build, '_citation.cpython-38-x86_64-linux-gnu.so')


class RefCitation:
    def __init__(self, build='h2000_v90', private_copy=True):
        src = so_path(build)
        if private_copy:
            self._tmp = tempfile.mkdtemp(prefix='refso_')
            dst = os.path.join(self._tmp, '_citation_%s_%d.so' % (build, id(self)))
            shutil.copy(src, dst)
            src = dst
        self.lib = ctypes.CDLL(src)
        self.lib.initialize.restype = None
        self.lib.step.restype = None
        self.lib.step.argtypes = [_D, _D]
        self._out = np.zeros(12)
        nb = 648 if build in ('gust', 'test') else 647
        self.X = np.ctypeslib.as_array((ctypes.c_double * 19).in_dll(self.lib, 'rtX'))
        self.B = np.ctypeslib.as_array((ctypes.c_double * nb).in_dll(self.lib, 'rtB'))
        self.initialize()

    def initialize(self):
        self.lib.initialize()

    def step(self, cmd):
        cmd = np.ascontiguousarray(cmd, dtype=np.float64)
        self.lib.step(cmd.ctypes.data_as(_D), self._out.ctypes.data_as(_D))
        return self._out.copy()

    def terminate(self):
        pass
