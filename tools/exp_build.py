#!/usr/bin/env python3
"""Experiment builds of the team kernel: python tools/exp_build.py <tag> [extra hipcc flags ...]
Recompiles ONLY serl_amd/csrc/rollout_team_nominal.hip with -DCITW_TEAM_INC="gen/citation_nominal_team_<tag>.inc" (a file
written by `tools/dag/codegen_team.py nominal --suffix=_<tag>`) and links it with the product's other objects into
serl_amd/csrc/libserl_amd_<tag>.so (select it with SERL_LIB=...; tools/ab.py prints the per-step times)."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from serl_amd import build as B

tag = sys.argv[1]
extra = sys.argv[2:]
B.build()                                   # the product objects
objdir = os.path.join(B.CSRC, 'build')
units = os.environ.get('EXP_UNITS', 'rollout_team_nominal.hip').split(',')      # e.g. EXP_UNITS=rollout_team_nominal.hip,rollout_team4_nominal.hip
objs = [os.path.join(objdir, u.replace('.hip', '.o')) for u in B.UNITS if u not in units]
inc = 'gen/citation_nominal_team_%s.inc' % tag
flags = list(B.FLAGS) + extra
if os.environ.get('EXP_DROP_LICM'):            # A/B: let the machine LICM hoist the model's f64 literals out of the stage loop
    i = flags.index('-disable-machine-licm')
    del flags[i - 1:i + 1]
if os.path.exists(os.path.join(B.CSRC, inc)):
    flags.append('-DCITW_TEAM_INC="%s"' % inc)
mine = []
for u in units:
    obj = os.path.join(objdir, u.replace('.hip', '_%s.o' % tag))
    r = subprocess.run([B.HIPCC] + flags + ['-c', os.path.join(B.CSRC, u), '-o', obj], capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stderr[-3000:])
    mine.append(obj)
lib = os.path.join(B.CSRC, 'libserl_amd_%s.so' % tag)
r = subprocess.run([B.HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + mine + objs, capture_output=True, text=True)
if r.returncode:
    sys.exit(r.stderr[-3000:])
print(lib)
