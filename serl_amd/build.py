"""Compile the HIP extension for gfx950, in-tree (serl_amd/csrc/libserl_amd.so).

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container; the built .so is
git-ignored but travels to the GPU box with the repo snapshot.
"""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(CSRC, 'libserl_amd.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
UNITS = ['serl_capi.hip', 'serl_ga.hip', 'serl_metrics.hip', 'serl_distill.hip', 'rollout_nominal.hip', 'rollout_ice.hip', 'rollout_cg_timed.hip', 'rollout_gust.hip', 'rollout_test.hip', 'rollout_wave_nominal.hip', 'rollout_wave_ice.hip',
         'rollout_wave_cg_timed.hip', 'rollout_wave_gust.hip', 'rollout_wave_test.hip', 'rollout_team_nominal.hip',
         'rollout_team_ice.hip', 'rollout_team_cg_timed.hip', 'rollout_team_gust.hip', 'rollout_team_test.hip',
         'rollout_team2_nominal.hip', 'rollout_team2_ice.hip', 'rollout_team2_cg_timed.hip', 'rollout_team2_gust.hip', 'rollout_team2_test.hip',
         'rollout_team4_mixed.hip', 'rollout_team4_nominal.hip', 'rollout_team4_ice.hip', 'rollout_team4_cg_timed.hip', 'rollout_team4_gust.hip', 'rollout_team4_test.hip',
         'rollout_teams2_nominal.hip', 'rollout_teams2_ice.hip', 'rollout_teams2_cg_timed.hip', 'rollout_teams2_gust.hip', 'rollout_teams2_test.hip',
         'rollout_teamr_nominal.hip', 'rollout_teamr_ice.hip', 'rollout_teamr_cg_timed.hip', 'rollout_teamr_gust.hip', 'rollout_teamr_test.hip',
         'rollout_team2s_nominal.hip', 'rollout_team2s_ice.hip', 'rollout_team2s_cg_timed.hip', 'rollout_team2s_gust.hip', 'rollout_team2s_test.hip',
         'rollout_half_nominal.hip', 'rollout_half_ice.hip', 'rollout_half_cg_timed.hip', 'rollout_half_gust.hip', 'rollout_half_test.hip']
# -ffp-contract=off: the IEEE-754 operation order of the reference binary is part of the contract (no FMA fusion).
# -disable-machine-licm: the model evaluation is inlined into the ODE5 stage loop; hoisting its ~110 f64 literals
# out of the loop (2 SGPRs each) makes them spill -- rematerialising them at use is cheaper.
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-Wno-unused-value',
         '-DCITW_EVAL_INLINE=__forceinline__', '-mllvm', '-disable-machine-licm']


def _deps():
    out = []
    for root, _, files in os.walk(CSRC):
        for f in files:
            if f.endswith(('.hip', '.h', '.inc')):
                out.append(os.path.join(root, f))
    out.append(os.path.join(os.path.dirname(HERE), 'include', 'serl_amd.h'))
    return out


def source_hash():
    """SHA-256 over the kernel sources (serl_amd/csrc/**.hip|.h|.inc, include/serl_amd.h, the build flags): what a measurement of the library belongs to.
    tools/profile_round.sh records it next to the counters it collects, tools/distill_profiles.py stores it in profiles/pmc_current.json and
    floors_current.json, and bench.py quotes those numbers only while it still matches the tree it runs from."""
    import hashlib
    h = hashlib.sha256(' '.join(FLAGS).encode())
    for f in sorted(_deps()):
        if os.sep + 'build' + os.sep in f or '_role_isa' in f or '_exp' in os.path.basename(f):
            continue
        h.update(os.path.relpath(f, os.path.dirname(HERE)).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()


def build(force=False, verbose=False, extra_flags=(), lib=None, tag=''):
    """extra_flags / lib / tag build a variant next to the product library (e.g. the phase-profiling build:
    extra_flags=['-DCITW_PROFILE'], lib='libserl_amd_prof.so', tag='_prof')."""
    LIB = os.path.join(CSRC, lib) if lib else globals()['LIB']
    if not force and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in _deps()):
        return LIB
    objdir = os.path.join(CSRC, 'build')
    os.makedirs(objdir, exist_ok=True)

    def cc(unit):
        obj = os.path.join(objdir, unit.replace('.hip', tag + '.o'))
        cmd = [HIPCC] + FLAGS + list(extra_flags) + ['-c', os.path.join(CSRC, unit), '-o', obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed for %s:\n%s' % (unit, r.stderr[-4000:]))
        if verbose and r.stderr:
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(UNITS), max(4, 4 * (os.cpu_count() or 8)))) as ex:      # (hipcc spends most of its time waiting on its own sub-processes)
        objs = list(ex.map(cc, UNITS))
    r = subprocess.run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs,
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('link failed:\n' + r.stderr[-4000:])
    return LIB


JITTER_LIB = 'libserl_amd_jitter.so'
TEAM_UNITS = [u for u in UNITS if u.startswith('rollout_team')]


def build_variant(lib, tag, extra_flags, units, force=False):
    """A library next to the product: `units` recompiled with `extra_flags` (objects build/<unit><tag>.o), every other object the
    product's own."""
    build()
    path = os.path.join(CSRC, lib)
    if not force and os.path.exists(path) and all(os.path.getmtime(path) >= os.path.getmtime(d) for d in _deps() + [LIB]):
        return path
    objdir = os.path.join(CSRC, 'build')

    def cc(unit):
        obj = os.path.join(objdir, unit.replace('.hip', tag + '.o'))
        r = subprocess.run([HIPCC] + FLAGS + list(extra_flags) + ['-c', os.path.join(CSRC, unit), '-o', obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed for %s:\n%s' % (unit, r.stderr[-4000:]))
        return obj
    with ThreadPoolExecutor(max_workers=min(len(units), max(4, 4 * (os.cpu_count() or 8)))) as ex:
        mine = list(ex.map(cc, units))
    objs = mine + [os.path.join(objdir, u.replace('.hip', '.o')) for u in UNITS if u not in units]
    r = subprocess.run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', path] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('link failed:\n' + r.stderr[-4000:])
    return path


def build_jitter(force=False):
    """TEST-ONLY library csrc/libserl_amd_jitter.so: the team kernel families compiled with -DCITW_POISON=1 -DCITW_JITTER=1
    (citation_wave.h: LDS blackboards start as slot-naming signalling NaNs; seeded pseudo-random pauses around every hand-over
    flag and barrier, seed from SERL_JITTER_SEED).  tests/test_gpu_rollout.py runs it against the oracle; the product never loads it."""
    return build_variant(JITTER_LIB, '_jitter', ['-DCITW_POISON=1', '-DCITW_JITTER=1'], TEAM_UNITS, force)


if __name__ == '__main__':
    if '--source-hash' in sys.argv:
        print(source_hash())
        sys.exit(0)
    print(build(force='--force' in sys.argv, verbose=True))
    if '--jitter' in sys.argv:
        print(build_jitter(force='--force' in sys.argv))
