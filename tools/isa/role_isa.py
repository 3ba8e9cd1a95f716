#!/usr/bin/env python3
"""Static instruction mix of every wavefront ROLE of the generated team evaluation (one kernel per role, the evaluation inlined into a
six-stage loop like rollout_team.inc does):  python tools/isa/role_isa.py [team .inc] [extra hipcc flags ...]
-> one line per role: instructions of the loop body by category (what `s_mov_b32` pairs, `v_mov_b32` coefficients, selects,
hazard nops ... cost next to the f64 arithmetic); --json for a machine-readable line."""
import os, re, subprocess, sys, collections, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from serl_amd import build as B

args = [a for a in sys.argv[1:] if a != '--json']
inc = next((a for a in args if a.endswith('.inc')), 'gen/citation_nominal_team.inc')
flags = [a for a in args if not a.endswith('.inc')]
K = 7
src = '''
#define CITW_SEARCH_BATCH 1
#define CITW_MAX_WAVES 1
#define CITW_M_ROWS 8
#define CITW_OUT2_ROWS 1
#define CITW_INV_SLOTS 8
#include "citation_wave.h"
#include "rollout_device.h"
#include "gen/citation_nominal_wave.inc"
#include "%s"
''' % inc
for r in range(K):
    src += '''
extern "C" __global__ void __launch_bounds__(512) role%d(double *out, unsigned steps, double t0)
{
  double acc = 0.0, xl = out[threadIdx.x & 63];
  CitwKRegs kr;                                 // the role's f64 literals in registers, like rollout_team.inc loads them (ROLE_ISA_NOK=1: literals)
  for (int j = 0; j < citw_nominal_team_NKLIT; ++j) kr.k[j] = g_x[j];
  for (unsigned fseq = 0; fseq < steps; ++fseq) {
#pragma nounroll
    for (int st = 0; st < 6; ++st) {
      acc += citw_nominal_team_eval_w%d(st, t0 + st, fseq, fseq, xl, %s, kr);
      xl = g_f[0][st][threadIdx.x & 15] + acc;
    }
  }
  out[threadIdx.x] = acc;
}
''' % (r, r, 'false' if os.environ.get('ROLE_ISA_NOK') else 'true')
path = os.path.join(B.CSRC, '_role_isa.hip')
open(path, 'w').write(src)
try:
    r = subprocess.run([B.HIPCC] + B.FLAGS + flags + ['--cuda-device-only', '-S', path, '-o', '/tmp/_role_isa.s'], capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stderr[-3000:])
finally:
    os.remove(path)
text = open('/tmp/_role_isa.s').read()


def category(o):
    if o.startswith(('v_add_f64', 'v_mul_f64', 'v_fma_f64', 'v_fmac_f64', 'v_div_', 'v_rcp_f64', 'v_rsq_f64', 'v_sqrt_f64', 'v_ldexp_f64', 'v_frexp',
                     'v_rndne_f64', 'v_trunc_f64', 'v_floor_f64', 'v_min_f64', 'v_max_f64', 'v_cvt')):
        return 'f64'
    if o.startswith(('s_mov_b32', 's_movk', 's_mov_b64')): return 's_mov'
    if o.startswith('v_mov'): return 'v_mov'
    if o.startswith('v_cndmask'): return 'cndmask'
    if o.startswith('v_cmp'): return 'v_cmp'
    if o.startswith('s_nop'): return 's_nop'
    if o.startswith('s_waitcnt'): return 'waitcnt'
    if o.startswith('ds_'): return 'lds'
    if o.startswith(('s_cbranch', 's_branch', 's_and_saveexec', 's_or_saveexec', 's_andn2_saveexec')): return 'branch'
    if o.startswith('s_load') or o.startswith('s_buffer'): return 's_load'
    if o.startswith('s_'): return 'salu'
    if o.startswith(('v_readlane', 'v_writelane', 'v_readfirstlane')): return 'lanex'
    if o.startswith('v_'): return 'valu'
    return 'other'


res = {}
for r in range(K):
    m = re.search(r'^role%d:.*?^\.Lfunc_end\d+:' % r, text, re.S | re.M)
    ops = [l.split()[0] for l in m.group(0).splitlines() if l.startswith('\t') and not l.strip().startswith((';', '.'))]
    c = collections.Counter(category(o) for o in ops)
    res[r] = dict(total=len(ops), **c)
    if '--json' not in sys.argv:
        print('role %d: %5d  ' % (r, len(ops)) + '  '.join('%s %d' % kv for kv in c.most_common()))
tot = collections.Counter()
for r in res:
    tot.update(res[r])
if '--json' in sys.argv:
    print(json.dumps(dict(per_role=res, total=dict(tot))))
else:
    print('all    : %5d  ' % tot['total'] + '  '.join('%s %d' % kv for kv in tot.most_common() if kv[0] != 'total'))
