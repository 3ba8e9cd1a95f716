"""Build the C restatement (oracle/_build/libcitation_oracle.so).  Test infrastructure."""
import os, subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, '_build', 'libcitation_oracle.so')
LIB_SHORT = os.path.join(HERE, '_build', 'libcitation_oracle_shortlibm.so')      # sin / cos / tan / pow = the CPU build of the product's citation_libm.h (citation_rt.h)


def build(force=False, short_libm=False):
    """Both flavours are built together; returns the glibc one (pinned to the reference binary) or, short_libm=True, the one that
    shares its sin / cos / tan / pow with the kernels (GPU-vs-oracle comparisons at zero tolerance)."""
    srcs = [os.path.join(HERE, f) for f in ('citation_ref.c', 'rollout_ref.c', 'citation_rt.h', 'citation_step.h',
                                            '../include/serl_amd.h', '../serl_amd/csrc/citation_libm.h', 'Makefile')]
    srcs += [os.path.join(HERE, 'gen', f) for f in sorted(os.listdir(os.path.join(HERE, 'gen')))]
    if force or not all(os.path.exists(l) and all(os.path.getmtime(l) >= os.path.getmtime(s) for s in srcs) for l in (LIB, LIB_SHORT)):
        subprocess.run(['make', '-C', HERE, '-s'] + (['-B'] if force else []), check=True)
    return LIB_SHORT if short_libm else LIB


REF_ARCHIVE = os.path.join(HERE, '_ref', 'reference_path.zip')
REF_BUILDS = ('h2000_v90', 'h2000_v150', 'h10000_v90', 'be', 'jr', 'sa', 'se', 'ice', 'noise', 'cg', 'cg_for', 'cg_timed',
              'gust', 'test')


def stage_ref(reference='/root/reference', force=False):
    """Pack the minimal reference tree of the hot path -- the reference's own Python (base/core/*.py, base/parameters.py,
    envs/phlabenv.py, the SWIG wrappers envs/<build>/citation.py) and its prebuilt dynamics libraries -- into ONE archive,
    oracle/_ref/reference_path.zip.  oracle/_ref/ is git-ignored (nothing of the reference enters the history) but not
    gpurun-ignored, so the archive travels to the GPU box, where bench.py's cpu_baseline leg times the reference's OWN
    unmodified Agent.evaluate on that box's host cores (tests/tools/time_reference.py; python modules are imported from the
    archive with zipimport, the shared objects are extracted to a temp dir per process).  Returns the path, or None when
    neither the reference nor a staged archive exists."""
    import zipfile
    if not os.path.isdir(os.path.join(reference, 'envs', 'h2000_v90')):
        return REF_ARCHIVE if os.path.exists(REF_ARCHIVE) else None
    if os.path.exists(REF_ARCHIVE) and not force:
        return REF_ARCHIVE
    os.makedirs(os.path.dirname(REF_ARCHIVE), exist_ok=True)
    files = ['base/parameters.py', 'envs/__init__.py', 'envs/phlabenv.py', 'envs/config.py']
    files += ['base/core/' + f for f in sorted(os.listdir(os.path.join(reference, 'base', 'core'))) if f.endswith('.py')]
    for b in REF_BUILDS:
        files += ['envs/%s/%s' % (b, f) for f in ('__init__.py', 'citation.py', '_citation.cpython-38-x86_64-linux-gnu.so')]
    tmp = REF_ARCHIVE + '.tmp'
    with zipfile.ZipFile(tmp, 'w', zipfile.ZIP_DEFLATED) as z:
        for f in files:
            if os.path.exists(os.path.join(reference, f)):          # (some build directories are namespace packages: no __init__.py)
                z.write(os.path.join(reference, f), f)
            elif not f.endswith('__init__.py'):
                raise FileNotFoundError(os.path.join(reference, f))
        for b in REF_BUILDS:                                       # directory entries: zipimport needs them for namespace packages
            if 'envs/%s/__init__.py' % b not in z.namelist():
                z.writestr('envs/%s/' % b, '')
    os.replace(tmp, REF_ARCHIVE)
    return REF_ARCHIVE


def build_xcheck(force=False):
    """GPU kernels around the lifted model (oracle/xcheck/): the checker's second implementation on the device."""
    from oracle import xcheck
    return xcheck.build(force)


if __name__ == '__main__':
    print(build(force=True))
    print(build_xcheck(force=True))
