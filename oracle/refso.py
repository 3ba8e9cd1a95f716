"""ctypes binding of the REFERENCE's own prebuilt dynamics library.  TEST INFRASTRUCTURE, and usable where
/root/reference exists (this build container) or where oracle/build.py stage_ref() left its archive (oracle/_ref/).

The cp38 SWIG module cannot be imported by Python 3.10, but ``ctypes.CDLL`` resolves its CPython
symbols against the running interpreter and the raw C entry points ``initialize`` /
``step(double*, double*)`` work.  State is file-scope inside the library, so every
independent instance needs its own copy of the file.
"""
import ctypes, os, shutil, tempfile
import numpy as np

_D = ctypes.POINTER(ctypes.c_double)
_ARCHIVE = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_ref', 'reference_path.zip')


def reference_root():
    """Where the reference lives: $SERL_REFERENCE, /root/reference (build container), or the archive oracle/build.py
    stage_ref() packed into the git-ignored oracle/_ref/ (the GPU box).  None if none of them exists."""
    for p in (os.environ.get('SERL_REFERENCE'), '/root/reference'):
        if p and (os.path.isdir(os.path.join(p, 'envs', 'h2000_v90')) or (os.path.isfile(p) and p.endswith('.zip'))):
            return p
    return _ARCHIVE if os.path.isfile(_ARCHIVE) else None


REF = reference_root() or '/root/reference'


def available():
    return reference_root() is not None


_extracted = {}          # build -> path of its library extracted from the staged archive (once per process, removed at exit)


def so_path(build):
    rel = 'envs/%s/_citation.cpython-38-x86_64-linux-gnu.so' % build
    if os.path.isfile(REF):          # staged archive: shared objects cannot be loaded from inside a zip
        if build not in _extracted:
            import atexit, zipfile
            d = tempfile.mkdtemp(prefix='refso_zip_')
            atexit.register(shutil.rmtree, d, True)
            with zipfile.ZipFile(REF) as z:
                _extracted[build] = z.extract(rel, d)
        return _extracted[build]
    return os.path.join(REF, rel)


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
