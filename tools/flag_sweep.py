#!/usr/bin/env python3
"""Compiler-flag variants of the one-episode team kernel (bit-identical by construction: no flag here touches floating-point semantics):
    python tools/flag_sweep.py tools/sweeps/<name>.json        {"tag": [extra hipcc flags ...], ...}
builds serl_amd/csrc/libserl_amd_<tag>.so for every tag (tools/exp_build.py: only rollout_team_nominal.hip is recompiled), reports the kernel's
static instruction count / registers / spills, and writes tools/sweep_libs.txt for tools/sweep_run.sh (A/B through tools/ab.py on the GPU box)."""
import json, os, re, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from serl_amd import build as B
B.build()
spec = json.load(open(sys.argv[1]))
LL = '/opt/rocm/lib/llvm/bin'


def meta(tag):
    obj = os.path.join(B.CSRC, 'build', 'rollout_team_nominal_%s.o' % tag)
    fb, elf = '/tmp/fs_%s.bin' % tag, '/tmp/fs_%s.elf' % tag
    subprocess.run([LL + '/llvm-objcopy', '--dump-section', '.hip_fatbin=' + fb, obj], check=True)
    subprocess.run([LL + '/clang-offload-bundler', '--unbundle', '--type=o', '--input=' + fb, '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--output=' + elf], check=True)
    notes = subprocess.run([LL + '/llvm-readelf', '--notes', elf], capture_output=True, text=True).stdout
    out = {}
    for blk in re.findall(r'- \.agpr_count.*?(?=\n\s+- \.agpr_count|\Z)', notes, re.S):
        if 'serl_rollout_team_kernel_nominal' in blk:
            for k in ('vgpr_count', 'vgpr_spill_count', 'sgpr_spill_count'):
                out[k] = int(re.search(r'\.%s:\s+(\d+)' % k, blk).group(1))
    dis = subprocess.run([LL + '/llvm-objdump', '-d', elf], capture_output=True, text=True).stdout
    m = re.search(r'<_Z32serl_rollout_team_kernel_nominal11RolloutArgs>:\n(.*?)(?=\n[0-9a-f]+ <|\Z)', dis, re.S)
    out['instructions'] = len(re.findall(r'^\s+\S+.*//', m.group(1), re.M)) if m else None
    os.remove(fb); os.remove(elf)
    return out


def one(item):
    tag, flags = item
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'exp_build.py'), tag] + flags, capture_output=True, text=True)
    if r.returncode:
        return tag, 'FAILED ' + (r.stderr + r.stdout)[-400:].replace('\n', ' | ')
    return tag, meta(tag)


with ThreadPoolExecutor(max_workers=int(os.environ.get('SWEEP_JOBS', 6))) as ex:
    res = list(ex.map(one, spec.items()))
ok = []
for t, m in res:
    print(t, ' '.join(spec[t]), m)
    if isinstance(m, dict):
        ok.append(t)
open(os.path.join(ROOT, 'tools', 'sweep_libs.txt'), 'w').write('\n'.join(ok) + '\n')
