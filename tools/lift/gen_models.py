#!/usr/bin/env python3
"""Generate the C restatement of the PH-LAB Citation dynamics for every reference build.

Usage:  python tools/lift/gen_models.py [--ref /root/reference] [build ...]

For each build directory ``envs/<build>/`` of the reference (14 directories, 8 distinct
binaries) this script
  1. disassembles the shared object (GNU objdump) and lifts ``step`` (model outputs + Derivative
     block update; the ODE5 driver is cut out and hand-written in ``citation_rt.h``),
     ``rt_GetLookupIndex``, ``rt_Lookup``, ``rt_Lookup2D_Normal``, ``rt_powd_snf``, ``matmultiply``
     and the ``ac_atmos`` / ``ac_axes`` S-function ``mdlOutputs`` bodies to C  (x86lift.py);
  2. probes the live library through ctypes for what ``initialize()`` sets up: the S-function
     wiring (child index -> kind, input/output block-signal offsets, mode parameter), the table3
     parameters, and the post-``initialize()`` images of ``rtX`` / ``rtDW``;
  3. dumps ``.rodata`` (aero tables ``rtConstP`` + ``rtConstB`` + literal pool) as f64.

Outputs (committed):
  <outdir>/citation_<key>.inc     generated model code (one per *code-distinct* build)
  <datadir>/citation_<build>.npz  ro[] f64, x0[19], dw0[31], meta   (one per data-distinct build)
"""
import sys, os, re, subprocess, hashlib, ctypes, struct, json, argparse
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import x86lift as XL

BUILDS = ['h2000_v90', 'h2000_v150', 'h10000_v90', 'cg', 'cg_for', 'cg_timed', 'ice', 'gust', 'test',
          'be', 'jr', 'sa', 'se', 'noise']


def so_path(ref, build):
    return os.path.join(ref, 'envs', build, '_citation.cpython-38-x86_64-linux-gnu.so')


def elf_info(so):
    syms = {}
    out = subprocess.run(['nm', '-S', so], capture_output=True, text=True, check=True).stdout
    for line in out.splitlines():
        p = line.split()
        if len(p) == 4:
            syms[p[3]] = (int(p[0], 16), int(p[1], 16))
        elif len(p) == 3:
            syms[p[2]] = (int(p[0], 16), 0)
    secs = {}
    out = subprocess.run(['readelf', '-SW', so], capture_output=True, text=True, check=True).stdout
    for m in re.finditer(r'\]\s+(\.\S+)\s+\S+\s+([0-9a-f]+)\s+([0-9a-f]+)\s+([0-9a-f]+)', out):
        secs[m.group(1)] = (int(m.group(2), 16), int(m.group(3), 16), int(m.group(4), 16))
    got = {}
    out = subprocess.run(['readelf', '-rW', so], capture_output=True, text=True, check=True).stdout
    for line in out.splitlines():
        p = line.split()
        if len(p) >= 5 and p[2] == 'R_X86_64_GLOB_DAT':
            got[int(p[0], 16)] = p[4]
    return syms, secs, got


def probe(so, syms, step_addr, mdl_addrs):
    """What initialize() leaves behind (S-function wiring, parameter arrays, initial state images)."""
    lib = ctypes.CDLL(so)
    D = ctypes.POINTER(ctypes.c_double)
    lib.initialize.restype = None
    lib.initialize()
    base = ctypes.cast(lib.step, ctypes.c_void_p).value - step_addr
    A = lambda s: base + syms[s][0]
    u64 = lambda a: ctypes.c_uint64.from_address(a).value
    f64 = lambda a: ctypes.c_double.from_address(a).value
    rtB, rtDW = A('rtB'), A('rtDW')
    arr = u64(A('rtM_'))
    kinds = {mdl_addrs[0]: 'atmos', mdl_addrs[1]: 'axes', mdl_addrs[2]: 'table3'}
    sf = []
    k = 0
    while True:
        S = u64(arr + 8 * k)
        fn = u64(S + 0x310) - base if S else None
        if fn not in kinds:
            break
        kind = kinds[fn]
        U = u64(u64(S + 0xe8) + 0x10)
        Y = u64(u64(S + 0xf0) + 0x10)
        ent = dict(kind=kind, y=Y - rtB)
        if kind == 'table3':
            ent['u'] = [u64(U + 8 * j) - rtB for j in range(3)]
            prm = u64(S + 0x100)
            P = []
            for j in range(4):
                mx = u64(prm + 8 * j)
                m, n = int(f64(mx)), int(f64(mx + 8))
                P.append(dict(m=m, n=n, v=[f64(mx + 0x10 + 8 * i) for i in range(m * n)]))
            ent['P'] = P
            ent['rwork'] = u64(S + 0x190) - rtDW
            ent['iwork'] = u64(S + 0x188) - rtDW
        else:
            ent['u'] = U - rtB
            if kind == 'axes':
                ent['mode'] = int(f64(u64(u64(S + 0x100)) + 0x10))
        sf.append(ent)
        k += 1
    x0 = np.ctypeslib.as_array((ctypes.c_double * 19).in_dll(lib, 'rtX')).copy()
    nb = syms['rtB'][1] // 8
    b0 = np.ctypeslib.as_array((ctypes.c_double * nb).in_dll(lib, 'rtB')).copy()
    dw0 = np.ctypeslib.as_array((ctypes.c_double * 31).in_dll(lib, 'rtDW')).copy()
    y0 = np.ctypeslib.as_array((ctypes.c_double * 12).in_dll(lib, 'rtY')).copy()
    m = A('rtM_')
    dt = f64(m + 0xba20)  # placeholder, validated by caller against the lifted field map
    return dict(sfun=sf, x0=x0, b0=b0, dw0=dw0, y0=y0), lib


def find_funcs(funcs):
    by = {}
    mdl = []
    for (a, n), inss in funcs.items():
        if n == 'mdlOutputs':
            mdl.append(a)
        else:
            by[n] = a
    return by, mdl


def lift_build(ref, build, prefix):
    so = so_path(ref, build)
    syms, secs, got = elf_info(so)
    funcs = XL.disassemble(so)
    by, mdl = find_funcs(funcs)
    assert len(mdl) == 3, mdl
    ro_lo, _, ro_sz = secs['.rodata']
    raw = open(so, 'rb').read()
    ro = np.frombuffer(raw[secs['.rodata'][1]:secs['.rodata'][1] + (ro_sz // 8) * 8], dtype='<f8').copy()
    info, lib = probe(so, syms, by['step'], mdl)
    M = syms['rtM_'][0]
    nB = syms['rtB'][1]

    # ---- model-struct (rtM_) field map, discovered from the step code ----------------------------
    step_ins = funcs[(by['step'], 'step')]
    # ODE5 region: from the first load of rtM_+0xa8 after the Derivative-bank copy loop to the
    # "jmp <out-copy>" that follows the clockTick increments.  Located structurally:
    tick_adds = [i for i in step_ins if i.mn == 'addl' and i.ripabs and M <= i.ripabs < M + 0x10000]
    assert len(tick_adds) == 2, tick_adds
    tick0_off = tick_adds[0].ripabs - M
    jmp_after = next(i for i in step_ins if i.addr > tick_adds[1].addr and i.mn == 'jmp')
    outcopy = jmp_after.target
    # major-step entry block (computes solverStopTime); it is the target of the first `je`
    first_je = next(i for i in step_ins if i.mn == 'je')
    sts_load = next(i for i in step_ins if i.mn == 'mov' and i.ripabs and M <= i.ripabs < M + 0x10000)
    simts_off = sts_load.ripabs - M
    blk = [i for i in step_ins if first_je.target <= i.addr < first_je.target + 0x30]
    step_sz = next(i for i in blk if i.mn == 'mulsd' and i.ripabs).ripabs - M
    stop_off = next(i for i in blk if i.mn == 'movsd' and i.ripabs).ripabs - M
    assert next(i for i in blk if i.mn == 'mov' and i.ripabs).ripabs - M == tick0_off
    # t pointer fields: the two pointer loads in the first block
    head = [i for i in step_ins if i.addr < first_je.target and i.ripabs and M <= i.ripabs < M + 0x10000]
    # pattern at function entry: mov M+simts -> eax ; ... mov M+tpp -> rax ; mov (rax),rax ; movsd (rax),xmm0;
    #                            mov M+tptr -> rax ; movsd xmm0,(rax)
    ent = [i for i in step_ins[:24] if i.ripabs and M <= i.ripabs < M + 0x10000]
    tpp_off = ent[1].ripabs - M
    tptr_off = ent[2].ripabs - M
    # the ODE5 driver starts at the first reference to M+tpp after the entry block
    ode_start = next(i for i in step_ins if i.addr > ent[2].addr and i.ripabs == M + tpp_off
                     and i.addr > first_je.target)
    # ... but the load of solverStopTime etc. precede the first derivative call: cut exactly there
    cut_addr = ode_start.addr

    rules = dict(
        abs_regions=[('RO', ro_lo, ro_lo + ro_sz), ('M', M, M + 0x10000),
                     ('X', syms['rtX'][0], syms['rtX'][0] + 0x98), ('B', syms['rtB'][0], syms['rtB'][0] + nB),
                     ('DW', syms['rtDW'][0], syms['rtDW'][0] + 0xf8), ('Y', syms['rtY'][0], syms['rtY'][0] + 0x60),
                     ('RTINF', syms['rtInf'][0], syms['rtInf'][0] + 8),
                     ('RTMINF', syms['rtMinusInf'][0], syms['rtMinusInf'][0] + 8),
                     ('RTNAN', syms['rtNaN'][0], syms['rtNaN'][0] + 8)],
        got_syms={'rtX': ('X', 0), 'rtB': ('B', 0), 'rtDW': ('DW', 0), 'rtY': ('Y', 0),
                  'rtConstP': ('RO', syms['rtConstP'][0]), 'rtConstB': ('RO', syms['rtConstB'][0]),
                  'rtInf': ('RTINF', 0), 'rtMinusInf': ('RTMINF', 0), 'rtNaN': ('RTNAN', 0)},
        ptr_loads={('M', 0): ('MCHILD', 0), ('M', tptr_off): ('TPTR', 0), ('M', tpp_off): ('TPP', 0),
                   ('TPP', 0): ('TPTR', 0),
                   ('SIMS', 0xe8): ('INP', 0), ('INP', 0x10): ('SU', 0),
                   ('SIMS', 0xf0): ('OUTP', 0), ('OUTP', 0x10): ('SY', 0),
                   ('SIMS', 0x100): ('PRM', 0), ('PRM', 0): ('MX0', 0)},
        ptr_loads_any={('MCHILD', None): 'SFUN'},
        cuts={cut_addr: ('goto', outcopy)},
        calls={},
    )
    P = prefix
    rules['calls'] = {
        by['rt_GetLookupIndex']: (P + 'rt_GetLookupIndex', [('rdi', 'p'), ('rsi', 'i'), ('xmm0', 'f')], 'i'),
        by['rt_Lookup']: (P + 'rt_Lookup', [('rdi', 'p'), ('rsi', 'i'), ('xmm0', 'f'), ('rdx', 'p')], 'f'),
        by['rt_Lookup2D_Normal']: (P + 'rt_Lookup2D_Normal', [('rdi', 'p'), ('rsi', 'i'), ('rdx', 'p'),
                                   ('rcx', 'i'), ('r8', 'p'), ('xmm0', 'f'), ('xmm1', 'f')], 'f'),
        by['rt_powd_snf']: (P + 'rt_powd_snf', [('xmm0', 'f'), ('xmm1', 'f')], 'f'),
        by['matmultiply']: (P + 'matmultiply', [('rdi', 'p'), ('rsi', 'p'), ('rdx', 'p')], 'void'),
        by['rtIsNaN']: ('LIFT_ISNAN', [('xmm0', 'f')], 'i'),
        by['rtIsInf']: ('LIFT_ISINF', [('xmm0', 'f')], 'i'),
        'sincos@plt': ('LIFT_SINCOS', [('xmm0', 'f'), ('rdi', 'p'), ('rsi', 'p')], 'void'),
        'pow@plt': ('LIFT_POW', [('xmm0', 'f'), ('xmm1', 'f')], 'f'),
        'exp@plt': ('LIFT_EXP', [('xmm0', 'f')], 'f'),
        'log10@plt': ('LIFT_LOG10', [('xmm0', 'f')], 'f'),
        'sqrt@plt': ('LIFT_SQRT', [('xmm0', 'f')], 'f'),
        'sin@plt': ('LIFT_SIN', [('xmm0', 'f')], 'f'),
        'cos@plt': ('LIFT_COS', [('xmm0', 'f')], 'f'),
        'tan@plt': ('LIFT_TAN', [('xmm0', 'f')], 'f'),
        'atan@plt': ('LIFT_ATAN', [('xmm0', 'f')], 'f'),
        'atan2@plt': ('LIFT_ATAN2', [('xmm0', 'f'), ('xmm1', 'f')], 'f'),
        'asin@plt': ('LIFT_ASIN', [('xmm0', 'f')], 'f'),
        'acos@plt': ('LIFT_ACOS', [('xmm0', 'f')], 'f'),
        'log@plt': ('LIFT_LOG', [('xmm0', 'f')], 'f'),
        'floor@plt': ('LIFT_FLOOR', [('xmm0', 'f')], 'f'),
    }
    # lifted leaf functions take the read-only pool first
    for k in ('rt_GetLookupIndex', 'rt_Lookup', 'rt_Lookup2D_Normal', 'rt_powd_snf', 'matmultiply'):
        cn, args, ret = rules['calls'][by[k]]
        rules['calls'][by[k]] = (cn, [('ro', 'ctx')] + args, ret)

    der_ins = funcs[(by['citation_to_python_derivatives'], 'citation_to_python_derivatives')]
    xdot_off = next(i for i in der_ins if i.ripabs and M <= i.ripabs < M + 0x10000).ripabs - M
    rules['ptr_loads'][('M', xdot_off)] = ('XDOT', 0)
    L = XL.Lifter(so, funcs, got, ro_lo, ro_sz, rules)
    S = XL.Spec
    pieces = []

    def lift(name, addr, spec):
        em = XL.Emitter(L, funcs[(addr, name)], spec, name)
        pieces.append(em.emit())

    ROP = 'const double *ro'
    lift('rt_GetLookupIndex', by['rt_GetLookupIndex'],
         S(P + 'rt_GetLookupIndex', [('rdi', 'p'), ('rsi', 'i'), ('xmm0', 'f')], 'i', extra_params=ROP))
    lift('rt_Lookup', by['rt_Lookup'],
         S(P + 'rt_Lookup', [('rdi', 'p'), ('rsi', 'i'), ('xmm0', 'f'), ('rdx', 'p')], 'f', extra_params=ROP))
    lift('rt_Lookup2D_Normal', by['rt_Lookup2D_Normal'],
         S(P + 'rt_Lookup2D_Normal', [('rdi', 'p'), ('rsi', 'i'), ('rdx', 'p'), ('rcx', 'i'), ('r8', 'p'),
                                      ('xmm0', 'f'), ('xmm1', 'f')], 'f', extra_params=ROP))
    lift('rt_powd_snf', by['rt_powd_snf'], S(P + 'rt_powd_snf', [('xmm0', 'f'), ('xmm1', 'f')], 'f',
                                              extra_params=ROP))
    lift('matmultiply', by['matmultiply'], S(P + 'matmultiply', [('rdi', 'p'), ('rsi', 'p'), ('rdx', 'p')],
                                              'void', extra_params=ROP))
    def lift_as(fname, name, addr, spec):
        em = XL.Emitter(L, funcs[(addr, name)], spec, fname)
        pieces.append(em.emit())
        print('   %s: %d loops unrolled at lift time' % (fname, em.n_unrolled))
        for a, n in em.idx_vectors:   # the hand-written GPU index search relies on strict monotonicity
            v = ro[(a - ro_lo) // 8:(a - ro_lo) // 8 + n]
            assert n >= 2 and np.all(np.diff(v) > 0), ('breakpoint vector not strictly increasing', hex(a), n, v)

    lift_as('ac_atmos', 'mdlOutputs', mdl[0], S(P + 'ac_atmos', [], 'void', entry={'rdi': ('SIMS', 0)},
                                 extra_params=ROP + ', const double *su, double *sy'))
    lift_as('ac_axes', 'mdlOutputs', mdl[1], S(P + 'ac_axes', [], 'void', entry={'rdi': ('SIMS', 0)},
                                 extra_params=ROP + ', const double *su, double *sy, int mode'))
    lift('citation_to_python_derivatives', by['citation_to_python_derivatives'],
         S(P + 'derivatives', [], 'void', extra_params='CitCtx *c, double *xdot',
           ret_expr='const double *ro = c->ro; (void)ro;'))
    lift_as('model', 'step', by['step'], S(P + 'model', [], 'void', entry={'rdi': ('CMD', 0), 'rsi': ('OUT', 0)},
                               extra_params='CitCtx *c, const double *cmd, double *out',
                               ret_expr='const double *ro = c->ro;'))

    # ---- S-function dispatch macros (what initialize() wired up) --------------------------------
    sf_lines = []
    for k, e in enumerate(info['sfun']):
        if e['kind'] == 'atmos':
            sf_lines.append('#define SFUN_CALL_%d() %sac_atmos(ro, &B_D(0x%x), &B_D(0x%x))' % (k, P, e['u'], e['y']))
        elif e['kind'] == 'axes':
            sf_lines.append('#define SFUN_CALL_%d() %sac_axes(ro, &B_D(0x%x), &B_D(0x%x), %d)' %
                            (k, P, e['u'], e['y'], e['mode']))
        else:
            sf_lines.append('#define SFUN_CALL_%d() cit_table3(c, &B_D(0x%x), &B_D(0x%x), &B_D(0x%x), &B_D(0x%x))'
                            % (k, e['u'][0], e['u'][1], e['u'][2], e['y']))
    t3 = next(e for e in info['sfun'] if e['kind'] == 'table3')
    assert t3['rwork'] == 26 * 8 and t3['iwork'] == 29 * 8, t3
    hdr = ['/* GENERATED by tools/lift/gen_models.py from envs/%s -- do not edit.' % build,
           ' * Restatement of the reference dynamics library (no source exists; SURVEY.md section 2.1):',
           ' * one C statement per x86-64 instruction, IEEE-754 operation order preserved. */',
           '#define RO_BASE 0x%xULL' % ro_lo,
           '#define M_I32_0x%x (c->major)' % simts_off,
           '#define M_I32_0x%x (c->tick)' % tick0_off,
           '#define M_D_0x%x (c->stop_time)' % stop_off,
           '#define M_D_0x%x (c->dt)' % step_sz,
           '#define CIT_NB %d' % (nB // 8),
           '#define RO_USED_LO 0x%xULL' % (L.ro_min & ~7),
           '#define RO_USED_HI 0x%xULL' % (ro_lo + (ro_sz // 8) * 8)] + sf_lines
    hdr.append('enum { %sRO_BASE_W = 0x%x / 8, %sRO_LO_W = 0x%x / 8, %sRO_HI_W = 0x%x / 8 };  /* f64 word range of .rodata the model reads */'
               % (P, ro_lo, P, L.ro_min & ~7, P, ro_lo + (ro_sz // 8) * 8))
    body = '\n'.join(hdr) + '\n\n' + '\n'.join(pieces)
    body += '\n#undef RO_BASE\n' + ''.join('#undef SFUN_CALL_%d\n' % k for k in range(len(info['sfun'])))
    body += ''.join('#undef %s\n' % s for s in ('RO_USED_LO', 'RO_USED_HI', 'M_I32_0x%x' % simts_off, 'M_I32_0x%x' % tick0_off,
                                               'M_D_0x%x' % stop_off, 'M_D_0x%x' % step_sz, 'CIT_NB'))
    meta = dict(build=build, ro_base=ro_lo, ro_count=len(ro), nB=nB // 8, constp=syms['rtConstP'][0],
                constb=syms['rtConstB'][0], sfun=info['sfun'], cut=cut_addr, outcopy=outcopy,
                dt=float(struct.unpack('<d', bytes((ctypes.c_char * 8).from_address(
                    ctypes.cast(lib.step, ctypes.c_void_p).value - by['step'] + M + step_sz)))[0]))
    return body, ro, info, meta


CODE_NAMES = {'h2000_v90': 'nominal'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref', default='/root/reference')
    ap.add_argument('builds', nargs='*')
    a = ap.parse_args()
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    outdirs = [os.path.join(root, 'oracle', 'gen')]      # the lifted source is the checker's; the product ships what tools/dag generates from it
    datadir = os.path.join(root, 'serl_amd', 'data')
    for d in outdirs + [datadir]:
        os.makedirs(d, exist_ok=True)
    index = {}
    by_md5, by_code = {}, {}
    for b in (a.builds or BUILDS):
        md5 = hashlib.md5(open(so_path(a.ref, b), 'rb').read()).hexdigest()
        if md5 in by_md5:
            index[b] = dict(index[by_md5[md5]])
            continue
        by_md5[md5] = b
        body, ro, info, meta = lift_build(a.ref, b, 'cit_@KEY@_')
        lines = body.splitlines()
        h = hashlib.sha1('\n'.join(lines[1:]).encode()).hexdigest()
        if h not in by_code:
            key = CODE_NAMES.get(b, b)
            by_code[h] = key
            text = body.replace('@KEY@', key)
            for d in outdirs:
                open(os.path.join(d, 'citation_%s.inc' % key), 'w').write(text)
            print('code variant %-10s from envs/%s: %d lines' % (key, b, len(lines)))
        key = by_code[h]
        t3e = next(e for e in info['sfun'] if e['kind'] == 'table3')
        t3 = np.array(sum([p['v'] for p in t3e['P']], []), dtype=np.float64)
        assert t3.shape == (46,) and [(p['m'], p['n']) for p in t3e['P']] == [(1, 3), (4, 1), (1, 3), (4, 9)]
        np.savez_compressed(os.path.join(datadir, 'citation_%s.npz' % b), ro=ro, x0=info['x0'], dw0=info['dw0'],
                            y0=info['y0'], t3=t3, dt=np.float64(meta['dt']), nB=np.int64(meta['nB']),
                            ro_base=np.int64(meta['ro_base']))
        index[b] = dict(data=b, code=key, nB=meta['nB'])
        print('build %-10s -> data %s, code %s' % (b, b, key))
    if a.builds:   # partial run: keep the other entries
        old = json.load(open(os.path.join(datadir, 'builds.json')))
        old.update(index)
        index = old
    json.dump(index, open(os.path.join(datadir, 'builds.json'), 'w'), indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
