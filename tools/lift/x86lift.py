"""x86-64 (scalar SSE2) -> portable C lifter for the PH-LAB Citation dynamics library.

The SERL reference ships its aircraft model only as a compiled Simulink/Embedded-Coder
shared object (``envs/<build>/_citation.cpython-38-x86_64-linux-gnu.so``).
There is no source to read, so the algorithm is *restated* by lifting the machine code of the
handful of functions on the hot path (``step``, ``rt_Lookup*``, the S-function ``mdlOutputs``
bodies ...) into plain C that keeps the exact IEEE-754 operation order.  The output is C with
``goto`` labels -- one statement per instruction -- which both gcc (CPU oracle) and hipcc (gfx950
device code) optimise back into SSA; all memory references are resolved *statically* to named
regions (model state ``X``, block signals ``B``, read-only tables ``RO``, the C stack ...), so no
emulated address space survives into the generated code.

This is build tooling: it needs the reference binary (``/root/reference``) and GNU objdump and is
run by ``tools/lift/gen_models.py``; its products are committed.
"""
from __future__ import annotations
import re, subprocess, collections

GPR64 = ['rax', 'rbx', 'rcx', 'rdx', 'rsi', 'rdi', 'rbp', 'rsp'] + ['r%d' % i for i in range(8, 16)]
_SUB = {}
for r in ['ax', 'bx', 'cx', 'dx', 'si', 'di', 'bp', 'sp']:
    _SUB['r' + r] = ('r' + r, 64); _SUB['e' + r] = ('r' + r, 32); _SUB[r] = ('r' + r, 16)
for r in 'abcd':
    _SUB[r + 'l'] = ('r' + r + 'x', 8)
_SUB['sil'] = ('rsi', 8); _SUB['dil'] = ('rdi', 8)
for i in range(8, 16):
    _SUB['r%d' % i] = ('r%d' % i, 64); _SUB['r%dd' % i] = ('r%d' % i, 32)
    _SUB['r%dw' % i] = ('r%d' % i, 16); _SUB['r%db' % i] = ('r%d' % i, 8)

NOPS = {'nop', 'nopl', 'nopw', 'endbr64', 'cs', 'data16', 'xchg'}  # xchg %ax,%ax only (checked)


class Ins:
    __slots__ = ('addr', 'mn', 'ops', 'raw', 'target', 'ripabs', 'callname')

    def __repr__(self):
        return '%x: %s %s' % (self.addr, self.mn, self.raw)


def split_ops(s):
    out, depth, cur = [], 0, ''
    for ch in s:
        if ch == '(':
            depth += 1
        elif ch == ')':
            depth -= 1
        if ch == ',' and depth == 0:
            out.append(cur.strip()); cur = ''
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def disassemble(so_path):
    txt = subprocess.run(['objdump', '-d', '--no-show-raw-insn', so_path], check=True,
                         capture_output=True, text=True).stdout
    funcs = collections.OrderedDict()
    cur = None
    for line in txt.splitlines():
        m = re.match(r'^([0-9a-f]+) <(.+)>:$', line)
        if m:
            cur = []
            funcs[(int(m.group(1), 16), m.group(2))] = cur
            continue
        m = re.match(r'^\s*([0-9a-f]+):\t(\S+)\s*(.*)$', line)
        if not m or cur is None:
            continue
        ins = Ins()
        ins.addr = int(m.group(1), 16)
        ins.mn = m.group(2)
        rest = m.group(3)
        ins.ripabs = None; ins.target = None; ins.callname = None
        if '#' in rest:
            rest, cm = rest.split('#', 1)
            mm = re.match(r'\s*([0-9a-f]+)', cm)
            if mm:
                ins.ripabs = int(mm.group(1), 16)
        rest = rest.strip()
        if ins.mn in ('rep', 'repz'):
            parts = rest.split(None, 1)
            ins.mn = 'rep_' + parts[0]
            rest = parts[1] if len(parts) > 1 else ''
        if ins.mn in ('cs', 'data16'):
            ins.mn = 'nop'
            rest = ''
        mm = re.match(r'^([0-9a-f]+) <(.+)>$', rest)
        if mm and (ins.mn.startswith('j') or ins.mn == 'call'):
            ins.target = int(mm.group(1), 16)
            ins.callname = mm.group(2)
            ins.ops = []
        else:
            ins.ops = split_ops(rest)
        ins.raw = rest
        cur.append(ins)
    return funcs


# ---------------------------------------------------------------------------------------------
# abstract values: (kind, const)   kind = 'int' | <region name> | 'top'
# ---------------------------------------------------------------------------------------------
INT = ('int', None)
TOP = ('top', None)


def join(a, b):
    if a == b:
        return a
    if a is None:
        return b
    if b is None:
        return a
    if a[0] != b[0]:
        return TOP
    return (a[0], None)


class Spec:
    """How a function is lifted: its C prototype and the abstract values of its entry registers."""

    def __init__(self, cname, args, ret, entry=None, frame=None, extra_params='', ret_expr=None):
        self.cname = cname          # C function name
        self.args = args            # list of (reg, 'p'|'i'|'f'|'pi') in ABI order
        self.ret = ret              # 'void' | 'f' | 'i'
        self.entry = entry or {}    # extra reg -> AV at entry
        self.frame = frame
        self.extra_params = extra_params
        self.prolog = ret_expr or ''


class Lifter:
    def __init__(self, so_path, funcs, got, ro_base, ro_size, rules):
        self.so = so_path
        self.funcs = funcs          # (addr,name) -> [Ins]
        self.byaddr = {a: (a, n) for (a, n) in funcs}
        self.got = got              # got slot address -> symbol
        self.ro_base, self.ro_size = ro_base, ro_size
        self.rules = rules          # configuration (regions, sym map, call specs, cuts ...)
        self.out = []

    # ---- operand parsing -------------------------------------------------------------------
    def parse_mem(self, op):
        m = re.match(r'^(%fs:)?(-?0x[0-9a-f]+|-?\d+)?(?:\((%\w+)?(?:,(%\w+))?(?:,(\d))?\))?$', op)
        if not m:
            raise ValueError('mem operand? ' + op)
        seg, disp, base, index, scale = m.groups()
        disp = int(disp, 16) if disp else 0
        return dict(seg=seg, disp=disp, base=base[1:] if base else None,
                    index=index[1:] if index else None, scale=int(scale) if scale else 1)

    @staticmethod
    def is_reg(op):
        return op.startswith('%') and '(' not in op and ':' not in op

    @staticmethod
    def is_imm(op):
        return op.startswith('$')

    @staticmethod
    def is_xmm(op):
        return op.startswith('%xmm')

    # ---- analysis ----------------------------------------------------------------------------
    def build_cfg(self, inss):
        idx = {ins.addr: i for i, ins in enumerate(inss)}
        leaders = {inss[0].addr}
        for i, ins in enumerate(inss):
            if ins.mn.startswith('j'):
                if ins.target in idx:
                    leaders.add(ins.target)
                if i + 1 < len(inss):
                    leaders.add(inss[i + 1].addr)
            if ins.mn == 'ret' and i + 1 < len(inss):
                leaders.add(inss[i + 1].addr)
        return idx, leaders

    def av_of_mem_addr(self, st, mem, ins):
        """abstract value of the effective address of a memory operand"""
        if mem['seg']:
            return ('FS', mem['disp'])
        if mem['base'] == 'rip':
            return self.av_of_abs(ins.ripabs)
        parts = []
        if mem['base']:
            parts.append((st['r'][_SUB[mem['base']][0]], 1))
        if mem['index']:
            parts.append((st['r'][_SUB[mem['index']][0]], mem['scale']))
        region = None
        const = mem['disp']
        for av, sc in parts:
            if av[0] == 'top':
                return TOP
            if av[0] != 'int':
                if region is not None or sc != 1:
                    return TOP
                region = av[0]
            if av[1] is None:
                const = None
            elif const is not None:
                const += av[1] * sc
        return (region or 'int', const)

    def av_of_abs(self, a):
        """abstract value of an absolute (rip-relative) address: region + offset"""
        for name, lo, hi in self.rules['abs_regions']:
            if lo <= a < hi:
                return (name, a - lo if name != 'RO' else a)
        if a in self.got:
            return ('GOT', a)
        return ('ABS', a)

    def load_av(self, st, addr_av, width):
        """abstract value produced by an integer load from addr_av"""
        kind, c = addr_av
        if kind == 'GOT':
            sym = self.got[c]
            r = self.rules['got_syms'].get(sym)
            if r is None:
                raise ValueError('GOT symbol not mapped: ' + sym)
            return r
        if kind == 'STK' and c is not None and width == 64:
            return st['s'].get(c, INT)
        key = (kind, c)
        if key in self.rules['ptr_loads']:
            return self.rules['ptr_loads'][key]
        if (kind, None) in self.rules['ptr_loads_any']:
            base = self.rules['ptr_loads_any'][(kind, None)]
            return (base, c // 8 if c is not None else None)
        if kind == 'FS':
            return ('int', 0)
        return INT

    def transfer(self, st, ins):
        """update abstract state over one instruction (GPRs + tracked stack slots + flag kind)"""
        mn, ops = ins.mn, ins.ops
        R = st['r']

        def setreg(name, av):
            full, w = _SUB[name]
            R[full] = av if w >= 32 else INT

        def getreg(name):
            return R[_SUB[name][0]]

        if mn in NOPS or mn.startswith('j') or mn in ('push', 'pop', 'ret'):
            if mn == 'pop' or mn == 'push':
                pass
            return
        if mn == 'call':
            for r in ('rax', 'rcx', 'rdx', 'rsi', 'rdi', 'r8', 'r9', 'r10', 'r11'):
                R[r] = INT
            st['f'] = None
            st['fc'] = None
            return
        width = 64
        # GPR-affecting instructions
        if mn in ('mov', 'movq', 'movl', 'movabs') and len(ops) == 2:
            src, dst = ops
            if self.is_xmm(src) or self.is_xmm(dst):
                if self.is_reg(dst) and not self.is_xmm(dst):
                    setreg(dst[1:], INT)
                elif not self.is_reg(dst):
                    a = self.av_of_mem_addr(st, self.parse_mem(dst), ins)
                    if a[0] == 'STK' and a[1] is not None:
                        st['s'].pop(a[1], None)
                return
            if self.is_reg(dst):
                w = _SUB[dst[1:]][1]
                if self.is_reg(src):
                    setreg(dst[1:], getreg(src[1:]) if w == 64 else
                           (getreg(src[1:]) if getreg(src[1:])[0] == 'int' else INT))
                elif self.is_imm(src):
                    setreg(dst[1:], ('int', int(src[1:], 16)))
                else:
                    a = self.av_of_mem_addr(st, self.parse_mem(src), ins)
                    setreg(dst[1:], self.load_av(st, a, w))
            else:
                a = self.av_of_mem_addr(st, self.parse_mem(dst), ins)
                if a[0] == 'STK' and a[1] is not None:
                    if self.is_reg(src) and _SUB[src[1:]][1] == 64:
                        st['s'][a[1]] = getreg(src[1:])
                    else:
                        st['s'].pop(a[1], None)
            return
        if mn in ('movsd', 'movups', 'movdqu', 'movapd', 'movaps', 'movupd', 'movdqa', 'movss'):
            dst = ops[1]
            if not self.is_reg(dst):
                a = self.av_of_mem_addr(st, self.parse_mem(dst), ins)
                if a[0] == 'STK' and a[1] is not None:
                    st['s'].pop(a[1], None)
                    if mn != 'movsd':
                        st['s'].pop(a[1] + 8, None)
            return
        if mn == 'lea':
            a = self.av_of_mem_addr(st, self.parse_mem(ops[0]), ins)
            w = _SUB[ops[1][1:]][1]
            if w == 32 and a[0] != 'int':
                a = INT
            if w == 32 and a[1] is not None:
                a = (a[0], a[1] & 0xffffffff)
            setreg(ops[1][1:], a)
            return
        if mn.startswith('set') and len(ops) == 1:
            if self.is_reg(ops[0]):
                setreg(ops[0][1:], INT)
            return
        if mn in ('movslq', 'movzbl', 'movzwl', 'movsbl', 'cltq', 'cvttsd2si', 'cvttss2si') or mn.startswith('cmov'):
            dst = 'rax' if mn == 'cltq' else ops[-1][1:]
            if mn == 'cltq':
                av = R['rax']
                R['rax'] = av if (av[0] == 'int' and av[1] is not None and av[1] < 2**31) else INT
            elif mn == 'movslq' and self.is_reg(ops[0]):
                av = getreg(ops[0][1:])
                setreg(dst, av if (av[0] == 'int' and av[1] is not None and av[1] < 2**31) else INT)
            else:
                setreg(dst, INT)
            return
        if mn in ('add', 'sub', 'addq', 'subq', 'addl', 'subl'):
            st['f'] = 'res'
            st['fc'] = None
            src, dst = ops
            if not self.is_reg(dst):
                return
            w = _SUB[dst[1:]][1]
            d = getreg(dst[1:])
            if self.is_imm(src):
                s = ('int', int(src[1:], 16))
            elif self.is_reg(src):
                s = getreg(src[1:])
            else:
                s = INT
            sign = 1 if mn.startswith('add') else -1
            if d[0] == 'top' or s[0] == 'top':
                res = TOP
            elif d[0] != 'int' and s[0] != 'int':
                res = INT if (sign == -1 and d[0] == s[0]) else TOP
                if res == INT and d[1] is not None and s[1] is not None:
                    res = ('int', d[1] - s[1])
            else:
                region = d[0] if d[0] != 'int' else s[0]
                if sign == -1 and d[0] == 'int' and s[0] != 'int':
                    region = 'top'
                c = None
                if d[1] is not None and s[1] is not None:
                    c = d[1] + sign * s[1]
                    if w == 32:
                        c &= 0xffffffff
                res = (region, c)
            if w == 32 and res[0] != 'int':
                res = INT
            if dst[1:] in ('rsp',):
                return  # frame adjustments handled by the emitter (fixed frame)
            setreg(dst[1:], res)
            if res[1] is not None:
                st['fc'] = ('res', res[1], w)
            return
        if mn in ('xor', 'and', 'or', 'imul', 'shl', 'shr', 'sar', 'neg', 'not', 'btc', 'inc', 'dec'):
            st['f'] = 'res'
            st['fc'] = ('res', 0, 64) if (mn == 'xor' and len(ops) == 2 and ops[0] == ops[1]) else None
            dst = ops[-1]
            if self.is_reg(dst):
                if mn == 'xor' and ops[0] == ops[1]:
                    setreg(dst[1:], ('int', 0))
                else:
                    setreg(dst[1:], INT)
            return
        if mn in ('cmp', 'cmpl', 'cmpq', 'test', 'testl', 'cmpb', 'testb'):
            st['f'] = 'cmp' if mn.startswith('cmp') else 'test'
            st['fc'] = None
            src, dst = ops

            def cval(op):
                if self.is_imm(op):
                    return Emitter.simm(op), 64
                if self.is_reg(op):
                    av = getreg(op[1:])
                    return (av[1], _SUB[op[1:]][1]) if av[1] is not None else (None, 64)
                return None, 64
            a, wa = cval(dst)
            b, wb = cval(src)
            w = min(wa, wb) if (self.is_reg(dst) and self.is_reg(src)) else (wa if self.is_reg(dst) else wb)
            if mn == 'cmpl':
                w = 32
            if a is not None and b is not None:
                # pointers of the same region compare by offset; mixed region/int compares are not decided
                ra = getreg(dst[1:])[0] if self.is_reg(dst) else 'int'
                rb = getreg(src[1:])[0] if self.is_reg(src) else 'int'
                if ra == rb or 'int' in (ra, rb) and ra == rb:
                    st['fc'] = (st['f'], a, b, w)
            return
        if mn in ('comisd', 'ucomisd'):
            st['f'] = 'fcmp'
            st['fc'] = None
            return
        if mn == 'rep_stos':
            R['rcx'] = ('int', 0)
            R['rdi'] = (R['rdi'][0], None)
            return
        if mn in SSE_ARITH or mn in ('sqrtsd', 'maxsd', 'minsd', 'andpd', 'andnpd', 'orpd', 'xorpd', 'pxor',
                                      'cmplesd', 'cmpnlesd', 'cmpltsd', 'cmpnltsd', 'cmpeqsd', 'cmpneqsd',
                                      'cvtsi2sd', 'cvtsi2sdl', 'cvtsi2sdq', 'unpcklpd', 'cmpunordsd'):
            return
        raise ValueError('transfer: unhandled %r' % ins)

    def analyse(self, inss, spec):
        idx, leaders = self.build_cfg(inss)
        entry = {'r': {r: INT for r in GPR64}, 's': {}, 'f': None, 'fc': None}
        entry['r']['rsp'] = ('STK', 0)
        for reg, kind in spec.args:
            if kind == 'p':
                entry['r'][reg] = ('P_' + reg, 0)
        for reg, av in spec.entry.items():
            entry['r'][reg] = av
        states = {inss[0].addr: entry}
        work = [inss[0].addr]
        instate = {}
        def copy(st):
            return {'r': dict(st['r']), 's': dict(st['s']), 'f': st['f'], 'fc': st.get('fc')}
        def merge(addr, st):
            if addr not in states:
                states[addr] = copy(st); work.append(addr); return
            old = states[addr]; changed = False
            for r in GPR64:
                j = join(old['r'][r], st['r'][r])
                if j != old['r'][r]:
                    old['r'][r] = j; changed = True
            for k in list(old['s']):
                j = join(old['s'][k], st['s'].get(k, INT)) if k in st['s'] else INT
                if j != old['s'][k]:
                    old['s'][k] = j; changed = True
            if old['f'] != st['f'] and old['f'] is not None:
                old['f'] = None; changed = True
            if old.get('fc') != st.get('fc') and old.get('fc') is not None:
                old['fc'] = None; changed = True
            if changed:
                work.append(addr)
        cuts = self.rules.get('cuts', {})
        while work:
            a = work.pop()
            st = copy(states[a])
            i = idx[a]
            while True:
                ins = inss[i]
                instate[ins.addr] = copy(st)
                if ins.addr in cuts and cuts[ins.addr][0] == 'goto':
                    merge(cuts[ins.addr][1], st)
                    break
                if ins.mn == 'ret':
                    break
                if ins.mn == 'jmp':
                    if ins.target in idx:
                        merge(ins.target, st)
                    break
                if ins.mn == 'call' and ins.callname and 'stack_chk_fail' in ins.callname:
                    break
                self.transfer(st, ins)
                if ins.mn.startswith('j'):
                    if ins.target in idx:
                        merge(ins.target, st)
                i += 1
                if i >= len(inss):
                    break
                if inss[i].addr in leaders:
                    merge(inss[i].addr, st)
                    break
        return idx, leaders, instate


SSE_ARITH = {'addsd': '+', 'subsd': '-', 'mulsd': '*', 'divsd': '/'}


# ---------------------------------------------------------------------------------------------
# C emission
# ---------------------------------------------------------------------------------------------
CC_INT = {  # condition -> expression over (fua,fub unsigned ; fsa,fsb signed ; fres signed result)
    'e': 'fua == fub', 'ne': 'fua != fub', 'a': 'fua > fub', 'ae': 'fua >= fub', 'b': 'fua < fub',
    'be': 'fua <= fub', 'g': 'fsa > fsb', 'ge': 'fsa >= fsb', 'l': 'fsa < fsb', 'le': 'fsa <= fsb',
    's': 'fres < 0', 'ns': 'fres >= 0', 'z': 'fua == fub', 'nz': 'fua != fub',
}
CC_RES = {'e': 'fres == 0', 'ne': 'fres != 0', 's': 'fres < 0', 'ns': 'fres >= 0', 'le': 'fres <= 0',
          'g': 'fres > 0', 'z': 'fres == 0', 'nz': 'fres != 0', 'l': 'fres < 0', 'ge': 'fres >= 0'}
CC_F = {  # after (u)comisd src,dst : fda = dst, fdb = src
    'a': 'fda > fdb', 'ae': 'fda >= fdb', 'b': '!(fda >= fdb)', 'be': '!(fda > fdb)',
    'e': '!(fda < fdb || fda > fdb)', 'ne': '(fda < fdb || fda > fdb)',
    'p': '(fda != fda || fdb != fdb)', 'np': '!(fda != fda || fdb != fdb)',
}


class Emitter:
    def __init__(self, lifter, inss, spec, fname):
        self.L = lifter
        self.inss = inss
        self.spec = spec
        self.fname = fname
        self.lines = []
        self.regions_used = set()
        self.idx_vectors = set()
        self.stk_rd = set()
        self.stk_addr = set()
        self.stk_dynamic = False
        self.labelmap = None
        self.extra_targets = set()
        self.frame = 0

    # -- helpers ------------------------------------------------------------------------------
    def reg_read(self, name):
        full, w = _SUB[name]
        if w == 64:
            return full
        if w == 32:
            return '(uint64_t)(uint32_t)%s' % full
        if w == 16:
            return '(uint64_t)(uint16_t)%s' % full
        return '(uint64_t)(uint8_t)%s' % full

    def reg_write(self, name, expr):
        full, w = _SUB[name]
        if w == 64:
            return '%s = (uint64_t)(%s);' % (full, expr)
        if w == 32:
            return '%s = (uint64_t)(uint32_t)(%s);' % (full, expr)
        if w == 16:
            return '%s = (%s & ~0xffffULL) | ((uint64_t)(%s) & 0xffff);' % (full, full, expr)
        return '%s = (%s & ~0xffULL) | ((uint64_t)(%s) & 0xff);' % (full, full, expr)

    def sext(self, expr, w):
        return {64: '(int64_t)(%s)', 32: '(int64_t)(int32_t)(%s)', 16: '(int64_t)(int16_t)(%s)',
                8: '(int64_t)(int8_t)(%s)'}[w] % expr

    def x(self, op):
        return 'x' + op[4:]

    def memref(self, st, op, ins):
        mem = self.L.parse_mem(op)
        av = self.L.av_of_mem_addr(st, mem, ins)
        if av[0] in ('top', 'int', 'ABS', 'GOT'):
            raise ValueError('unresolved memory operand %r at %r (av=%r) regs=%r' % (op, ins, av,
                             {k: v for k, v in st['r'].items() if k in (mem['base'], mem['index'])}))
        region = av[0]
        if av[1] is not None:
            off = '0x%x' % av[1] if av[1] >= 0 else '(-0x%x)' % (-av[1])
        else:
            terms = []
            if mem['disp']:
                terms.append('0x%x' % mem['disp'] if mem['disp'] >= 0 else '(-0x%x)' % -mem['disp'])
            if mem['base']:
                terms.append(_SUB[mem['base']][0])
            if mem['index']:
                terms.append('%s*%d' % (_SUB[mem['index']][0], mem['scale']) if mem['scale'] != 1
                             else _SUB[mem['index']][0])
            off = '(int64_t)(' + ' + '.join(terms) + ')'
        self.regions_used.add(region)
        if region == 'RO' and av[1] is not None:
            self.L.ro_min = min(getattr(self.L, 'ro_min', 1 << 62), av[1])
        return region, off, av[1]

    def ld_d(self, region, off):
        if region == 'STK':
            if off.startswith('0x'):
                self.stk_rd.add(('d', int(off, 16)))
                return 'STKS_D(%s)' % off
            self.stk_dynamic = True
        return '%s_D(%s)' % (region, off)

    def st_d(self, region, off, val):
        if region == 'STK':
            if off.startswith('0x'):
                return 'STKS_ST_D(%s, %s);' % (off, val)
            self.stk_dynamic = True
            return 'STK_D(%s) = %s; STK_I(%s) = d2u(%s);' % (off, val, off, val)
        return '%s_D(%s) = %s;' % (region, off, val)

    def ld_i(self, region, off, w):
        if region == 'FS':
            return '0ULL'
        if region == 'STK':
            if off.startswith('0x'):
                self.stk_rd.add(('i' if w == 64 else 'w', int(off, 16)))
                return 'STKS_I(%s)' % off if w == 64 else 'STKS_W(%s)' % off
            self.stk_dynamic = True
            return 'STK_I(%s)' % off if w == 64 else 'STK_W(%s)' % off
        if region == 'M' or region in self.L.rules.get('int_regions', ()):
            return '%s_I%d(%s)' % (region, w, off)
        if w == 64:
            return 'd2u(%s_D(%s))' % (region, off)
        raise ValueError('narrow int load from %s' % region)

    def st_i(self, region, off, val, w):
        if region == 'STK':
            if off.startswith('0x'):
                return ('STKS_ST_I(%s, %s);' if w == 64 else 'STKS_ST_W(%s, %s);') % (off, val)
            self.stk_dynamic = True
            if w == 64:
                return 'STK_I(%s) = %s; STK_D(%s) = u2d(%s);' % (off, val, off, val)
            return 'STK_W(%s) = (uint32_t)(%s);' % (off, val)
        if region == 'M' or region in self.L.rules.get('int_regions', ()):
            return '%s_I%d(%s) = %s;' % (region, w, off, val)
        if w == 64:
            return '%s_D(%s) = u2d(%s);' % (region, off, val)
        raise ValueError('narrow int store to %s' % region)

    def off_plus(self, off, k):
        if off.startswith('0x'):
            return '0x%x' % (int(off, 16) + k)
        return '(%s + %d)' % (off, k)

    def cond(self, cc, st, ins):
        kind = st['f']
        if kind == 'fcmp':
            return CC_F[cc]
        if kind == 'cmp':
            return CC_INT[cc]
        if kind in ('test', 'res'):
            return CC_RES[cc]
        raise ValueError('flags unknown at %r' % ins)

    def ptr_expr(self, av, regname):
        region = av[0]
        if region in ('int', 'top'):
            raise ValueError('pointer arg %s is not a pointer: %r' % (regname, av))
        self.regions_used.add(region)
        off = '0x%x' % av[1] if av[1] is not None else '(int64_t)%s' % regname
        if region == 'STK':
            if av[1] is None:
                self.stk_dynamic = True
            else:
                self.stk_addr.add(av[1])
                return '&STKS_D(0x%x)' % av[1]
        if region == 'RO' and av[1] is not None:
            self.L.ro_min = min(getattr(self.L, 'ro_min', 1 << 62), av[1])
            self.L.tbl_ptrs = getattr(self.L, 'tbl_ptrs', set()) | {av[1]}
        return '&%s_D(%s)' % (region, off)

    # -- main ------------------------------------------------------------------------------------
    def lbl(self, addr):
        return self.labelmap(addr) if self.labelmap else 'L_%x' % addr

    @staticmethod
    def eval_cc(cc, fc):
        """decide a condition code from concrete flag operands; None if unknown"""
        if fc is None:
            return None
        def sx(v, w):
            v &= (1 << w) - 1
            return v - (1 << w) if v >> (w - 1) else v
        if fc[0] == 'cmp':
            _, a, b, w = fc
            ua, ub = a & ((1 << w) - 1), b & ((1 << w) - 1)
            sa, sb = sx(a, w), sx(b, w)
            res = sx(a - b, w)
            tbl = {'e': ua == ub, 'ne': ua != ub, 'z': ua == ub, 'nz': ua != ub, 'a': ua > ub, 'ae': ua >= ub,
                   'b': ua < ub, 'be': ua <= ub, 'g': sa > sb, 'ge': sa >= sb, 'l': sa < sb, 'le': sa <= sb,
                   's': res < 0, 'ns': res >= 0}
            return tbl.get(cc)
        if fc[0] == 'test':
            _, a, b, w = fc
            res = sx(a & b, w)
        else:
            res = sx(fc[1], fc[2])
        tbl = {'e': res == 0, 'ne': res != 0, 'z': res == 0, 'nz': res != 0, 's': res < 0, 'ns': res >= 0,
               'le': res <= 0, 'g': res > 0, 'l': res < 0, 'ge': res >= 0}
        return tbl.get(cc)

    def build_blocks(self, idx, leaders):
        inss = self.inss
        starts = sorted(idx[a] for a in leaders if a in idx)
        blocks = []
        for n, st in enumerate(starts):
            en = starts[n + 1] if n + 1 < len(starts) else len(inss)
            blocks.append((st, en))
        bid = {inss[st].addr: n for n, (st, en) in enumerate(blocks)}
        succ = {n: [] for n in range(len(blocks))}
        for n, (st, en) in enumerate(blocks):
            last = inss[en - 1]
            if last.mn == 'ret' or (last.mn == 'call' and last.callname and 'stack_chk_fail' in last.callname):
                continue
            if last.mn.startswith('j'):
                if last.target in bid:
                    succ[n].append(bid[last.target])
                if last.mn == 'jmp':
                    continue
            if n + 1 < len(blocks):
                succ[n].append(n + 1)
        pred = {n: [] for n in range(len(blocks))}
        for n, ss in succ.items():
            for t in ss:
                pred[t].append(n)
        return blocks, bid, succ, pred

    def find_loops(self, blocks, succ, pred):
        """natural loops keyed by header block: header -> set(body blocks).  A back edge is an edge whose
        target dominates its source (address order is no guide: compilers move cold blocks out of line)."""
        n = len(blocks)
        reach, stack = set(), [0]
        while stack:
            b = stack.pop()
            if b in reach:
                continue
            reach.add(b)
            stack.extend(succ[b])
        full = set(reach)
        dom = {b: set(full) for b in reach}
        dom[0] = {0}
        changed = True
        order = sorted(reach)
        while changed:
            changed = False
            for b in order:
                if b == 0:
                    continue
                ps = [p for p in pred[b] if p in reach]
                new = set(full)
                for p in ps:
                    new &= dom[p]
                new.add(b)
                if new != dom[b]:
                    dom[b] = new
                    changed = True
        loops = {}
        for b in order:
            for t in succ[b]:
                if t in dom[b]:      # back edge b -> t
                    body = {t}
                    stack = [b]
                    while stack:
                        x = stack.pop()
                        if x in body:
                            continue
                        body.add(x)
                        stack.extend(p for p in pred[x] if p in reach)
                    loops.setdefault(t, set()).update(body)
        return loops

    def try_unroll(self, header, body, blocks, bid, succ, pred, instate, copy_state, max_iter=64):
        """Concrete-integer unrolling of one loop.  Returns a list of emission items or None."""
        L, inss = self.L, self.inss
        cuts = L.rules.get('cuts', {})
        for b in body:
            st0, en = blocks[b]
            for i in range(st0, en):
                if inss[i].addr in cuts or inss[i].mn == 'ret' or inss[i].addr not in instate:
                    return None
            if b != header and any(p not in body and inss[blocks[p][0]].addr in instate for p in pred[b]):
                return None     # entered from outside other than through the header
        # state on the entry edges
        entry = None
        for p_ in pred[header]:
            if p_ in body:
                continue
            st0, en = blocks[p_]
            if inss[st0].addr not in instate:
                continue
            st = copy_state(instate[inss[st0].addr])
            for i in range(st0, en):
                if not (inss[i].mn == 'jmp' or inss[i].mn == 'ret'):
                    L.transfer(st, inss[i])
            entry = st if entry is None else self.join_states(entry, st)
        if entry is None:
            return None
        # topological order of the body without edges into the header
        order, seen = [], set()
        def dfs(b):
            if b in seen:
                return
            seen.add(b)
            for t in succ[b]:
                if t in body and t != header:
                    dfs(t)
            order.append(b)
        dfs(header)
        order.reverse()
        if set(order) != set(body):
            return None
        items = []
        state_h = entry
        haddr = inss[blocks[header][0]].addr
        k = 0
        while True:
            bstate = {header: state_h}
            nxt, exited = None, False
            def lm(addr, k=k):
                if addr == haddr:
                    return 'L_%x_u%d' % (addr, k + 1)
                if addr in bid and bid[addr] in body:
                    return 'L_%x_u%d' % (addr, k)
                return 'L_%x' % addr
            for b in order:
                if b not in bstate:
                    continue
                st = copy_state(bstate[b])
                st0, en = blocks[b]
                items.append(('label', 'L_%x_u%d' % (inss[st0].addr, k)))
                succs = None
                for i in range(st0, en):
                    ins = inss[i]
                    if ins.mn.startswith('j') and ins.mn != 'jmp':
                        dec = self.eval_cc(ins.mn[1:], st.get('fc'))
                        ft = bid.get(inss[en].addr) if en < len(inss) else None
                        tg = bid.get(ins.target)
                        if tg is None:
                            return None
                        if dec is True:
                            items.append(('raw', 'goto %s;' % lm(ins.target)))
                            succs = [tg]
                        elif dec is False:
                            succs = [ft]
                            items.append(('raw', 'goto %s;' % lm(inss[en].addr)))
                        else:
                            items.append(('ins', ins, copy_state(st), lm))
                            items.append(('raw', 'goto %s;' % lm(inss[en].addr)))
                            succs = [tg, ft]
                        break
                    if ins.mn == 'jmp':
                        tg = bid.get(ins.target)
                        if tg is None:
                            return None
                        items.append(('raw', 'goto %s;' % lm(ins.target)))
                        succs = [tg]
                        break
                    items.append(('ins', ins, copy_state(st), lm))
                    L.transfer(st, ins)
                if succs is None:   # fell off the end of the block
                    if en >= len(inss):
                        return None
                    items.append(('raw', 'goto %s;' % lm(inss[en].addr)))
                    succs = [bid[inss[en].addr]]
                for t in succs:
                    if t is None:
                        return None
                    if t == header:
                        nxt = copy_state(st) if nxt is None else self.join_states(nxt, st)
                    elif t in body:
                        bstate[t] = copy_state(st) if t not in bstate else self.join_states(bstate[t], st)
                    else:
                        exited = True
                        self.extra_targets.add(inss[blocks[t][0]].addr)
            if nxt is None:
                break
            if exited:
                return None      # both "continue" and "leave" reachable: trip count is data dependent
            k += 1
            if k > max_iter:
                return None
            state_h = nxt
        items.append(('label', 'L_%x_u%d' % (haddr, k + 1)))   # never targeted; keeps label set closed
        return items

    @staticmethod
    def join_states(a, b):
        out = {'r': {}, 's': {}, 'f': a['f'] if a['f'] == b['f'] else None,
               'fc': a.get('fc') if a.get('fc') == b.get('fc') else None}
        for r in GPR64:
            out['r'][r] = join(a['r'][r], b['r'][r])
        for k_ in a['s']:
            if k_ in b['s']:
                out['s'][k_] = join(a['s'][k_], b['s'][k_])
        return out

    def emit(self):
        L = self.L
        idx, leaders, instate = L.analyse(self.inss, self.spec)
        cuts = L.rules.get('cuts', {})
        out = self.lines
        self.labelmap = None
        self.extra_targets = set()

        def copy_state(st):
            return {'r': dict(st['r']), 's': dict(st['s']), 'f': st['f'], 'fc': st.get('fc')}
        blocks, bid, succ, pred = self.build_blocks(idx, leaders)
        unrolled, skip = {}, set()
        if L.rules.get('unroll', True):
            loops = self.find_loops(blocks, succ, pred)
            for h in sorted(loops):
                body = loops[h]
                if any(o != h and o in body for o in loops):    # only innermost loops
                    continue
                items = self.try_unroll(h, body, blocks, bid, succ, pred, instate, copy_state)
                if items is None:
                    continue
                unrolled[self.inss[blocks[h][0]].addr] = items
                for b in body:
                    for i in range(*blocks[b]):
                        skip.add(self.inss[i].addr)
        self.n_unrolled = len(unrolled)
        targets = set(self.extra_targets)
        for ins in self.inss:
            if ins.mn.startswith('j') and ins.target in idx:
                targets.add(ins.target)
        for a, c in cuts.items():
            if c[0] == 'goto':
                targets.add(c[1])

        def put(ins, st):
            try:
                stmts = self.emit_ins(ins, st)
            except Exception as e:
                raise RuntimeError('%s: at %r: %s' % (self.fname, ins, e))
            for n_, s_ in enumerate(stmts):
                out.append('  %s  /* %x %s %s */' % (s_, ins.addr, ins.mn, ins.raw) if n_ == 0 else '  ' + s_)

        for ins in self.inss:
            if ins.addr in unrolled:
                out.append('L_%x: ;' % ins.addr)
                for it in unrolled[ins.addr]:
                    if it[0] == 'label':
                        out.append('%s: ;' % it[1])
                    elif it[0] == 'raw':
                        out.append('  ' + it[1])
                    else:
                        self.labelmap = it[3]
                        put(it[1], it[2])
                        self.labelmap = None
                continue
            if ins.addr in skip:
                continue
            if ins.addr in targets:
                out.append('L_%x: ;' % ins.addr)
            if ins.addr not in instate:
                continue  # unreachable
            st = instate[ins.addr]
            if ins.addr in cuts:
                c = cuts[ins.addr]
                out.append('  /* cut @%x */ %s' % (ins.addr, 'goto L_%x;' % c[1] if c[0] == 'goto' else c[1]))
                if c[0] == 'goto':
                    continue
            put(ins, st)
        return self.wrap()

    def wrap(self):
        spec = self.spec
        params = []
        for reg, kind in spec.args:
            if kind == 'p':
                params.append('double *p_%s' % reg)
            elif kind == 'f':
                params.append('double a_%s' % reg)
            else:
                params.append('int64_t a_%s' % reg)
        if spec.extra_params:
            params.insert(0, spec.extra_params)
        rett = {'void': 'void', 'f': 'double', 'i': 'int64_t'}[spec.ret]
        head = 'LIFT_FN_%s %s %s(%s)\n{' % (self.fname.replace('citation_to_python_', ''), rett, spec.cname, ', '.join(params) or 'void')
        decl = ['  uint64_t ' + ', '.join('%s = 0' % r for r in GPR64 if r != 'rsp') + ';',
                '  double ' + ', '.join('x%d = 0, x%dh = 0' % (i, i) for i in range(16)) + ';',
                '  uint64_t fua = 0, fub = 0; int64_t fsa = 0, fsb = 0, fres = 0; double fda = 0, fdb = 0;',
                '  (void)fua; (void)fub; (void)fsa; (void)fsb; (void)fres; (void)fda; (void)fdb;']
        scalar = not self.stk_dynamic
        import re as _re
        if scalar:
            # every stack access has a constant offset: the x86 spill slots become C scalars
            rd_d = {o for k_, o in self.stk_rd if k_ == 'd'} | self.stk_addr
            rd_i = {o for k_, o in self.stk_rd if k_ == 'i'}
            rd_w = {o for k_, o in self.stk_rd if k_ == 'w'}
            def sub_st(m):
                kind, off, val = m.group(1), int(m.group(2), 16), m.group(3)
                outs = []
                if kind == 'D':
                    if off in rd_d: outs.append('sd_0x%x = %s;' % (off, val))
                    if off in rd_i: outs.append('si_0x%x = d2u(%s);' % (off, val))
                elif kind == 'I':
                    if off in rd_i: outs.append('si_0x%x = %s;' % (off, val))
                    if off in rd_d: outs.append('sd_0x%x = u2d(%s);' % (off, val))
                else:
                    if off in rd_w: outs.append('sw_0x%x = (uint32_t)(%s);' % (off, val))
                return ' '.join(outs) if outs else '/* dead spill store */;'
            pat = _re.compile(r'STKS_ST_([DIW])\((0x[0-9a-f]+), (.*?)\);')
            self.lines = [pat.sub(sub_st, ln) for ln in self.lines]
            self.lines = [_re.sub(r'STKS_D\((0x[0-9a-f]+)\)', r'sd_\1', ln) for ln in self.lines]
            self.lines = [_re.sub(r'STKS_I\((0x[0-9a-f]+)\)', r'si_\1', ln) for ln in self.lines]
            self.lines = [_re.sub(r'STKS_W\((0x[0-9a-f]+)\)', r'sw_\1', ln) for ln in self.lines]
            if rd_d: decl.append('  double ' + ', '.join('sd_0x%x = 0' % o for o in sorted(rd_d)) + ';')
            if rd_i: decl.append('  uint64_t ' + ', '.join('si_0x%x = 0' % o for o in sorted(rd_i)) + ';')
            if rd_w: decl.append('  uint32_t ' + ', '.join('sw_0x%x = 0' % o for o in sorted(rd_w)) + ';')
        else:
            self.lines = [_re.sub(r'STKS_ST_D\((0x[0-9a-f]+), (.*?)\);', r'STK_D(\1) = \2; STK_I(\1) = d2u(\2);', ln) for ln in self.lines]
            self.lines = [_re.sub(r'STKS_ST_I\((0x[0-9a-f]+), (.*?)\);', r'STK_I(\1) = \2; STK_D(\1) = u2d(\2);', ln) for ln in self.lines]
            self.lines = [_re.sub(r'STKS_ST_W\((0x[0-9a-f]+), (.*?)\);', r'STK_W(\1) = (uint32_t)(\2);', ln) for ln in self.lines]
            self.lines = [_re.sub(r'STKS_([DIW])\(', r'STK_\1(', ln) for ln in self.lines]
            n = (self.frame + 0x40) // 8
            decl.append('  double stk_d[%d]; uint64_t stk_i[%d]; uint32_t stk_w[%d];' % (n, n, 2 * n))
            decl.append('  (void)stk_d; (void)stk_i; (void)stk_w;')
        for reg, kind in spec.args:
            if kind == 'f':
                decl.append('  x%s = a_%s;' % (reg[3:], reg))
            elif kind in ('i',):
                decl.append('  %s = (uint64_t)a_%s;' % (reg, reg))
        if spec.prolog:
            decl.append('  ' + spec.prolog)
        for a, n in sorted(self.idx_vectors):
            decl.append('  LIFT_IDX_DECL(0x%x)  /* breakpoint vector, %d entries */' % (a, n))
        body = '\n'.join(self.lines)
        tail = '}\n'
        key = self.fname.replace('citation_to_python_', '')
        return ('#ifndef LIFT_OMIT_%s\n' % key + head + '\n' + '\n'.join(decl) + '\n' + body + '\n' + tail +
                '#endif /* LIFT_OMIT_%s */\n' % key)

    def emit_ins(self, ins, st):
        L = self.L
        mn, ops = ins.mn, ins.ops
        isx, isreg, isimm = L.is_xmm, L.is_reg, L.is_imm
        if mn in NOPS:
            if mn == 'xchg' and ops != ['%ax', '%ax']:
                raise ValueError('xchg')
            return []
        if mn in ('push', 'pop'):
            return []
        if mn == 'ret':
            return [{'void': 'return;', 'f': 'return x0;', 'i': 'return (int64_t)rax;'}[self.spec.ret]]
        if mn == 'jmp':
            if not (self.inss[0].addr <= ins.target <= self.inss[-1].addr):
                # tail call: call + return
                return self.emit_call(ins, st) + [
                    {'void': 'return;', 'f': 'return x0;', 'i': 'return (int64_t)rax;'}[self.spec.ret]]
            return ['goto %s;' % self.lbl(ins.target)]
        if mn.startswith('j'):
            return ['if (%s) goto %s;' % (self.cond(mn[1:], st, ins), self.lbl(ins.target))]
        if mn.startswith('set') and len(ops) == 1 and isreg(ops[0]):
            return [self.reg_write(ops[0][1:], '(%s) ? 1 : 0' % self.cond(mn[3:], st, ins))]
        if mn in ('movzbl', 'movzwl'):
            src, dst = ops
            w = 8 if mn == 'movzbl' else 16
            if not isreg(src):
                raise ValueError('movz from memory')
            return [self.reg_write(dst[1:], '(uint64_t)(uint%d_t)%s' % (w, _SUB[src[1:]][0]))]
        if mn.startswith('cmov'):
            return ['if (%s) { %s }' % (self.cond(mn[4:], st, ins),
                                       self.reg_write(ops[1][1:], self.reg_read(ops[0][1:])))]
        if mn == 'call':
            return self.emit_call(ins, st)
        # ---- SSE moves ----
        if mn == 'movsd':
            src, dst = ops
            if isx(src) and isx(dst):
                return ['%s = %s;' % (self.x(dst), self.x(src))]
            if isx(dst):
                r, o, _ = self.memref(st, src, ins)
                return ['%s = %s; %sh = 0;' % (self.x(dst), self.ld_d(r, o), self.x(dst))]
            r, o, _ = self.memref(st, dst, ins)
            return [self.st_d(r, o, self.x(src))]
        if mn in ('movapd', 'movaps', 'movups', 'movupd', 'movdqu', 'movdqa'):
            src, dst = ops
            if isx(src) and isx(dst):
                return ['%s = %s; %sh = %sh;' % (self.x(dst), self.x(src), self.x(dst), self.x(src))]
            if isx(dst):
                r, o, _ = self.memref(st, src, ins)
                return ['%s = %s; %sh = %s;' % (self.x(dst), self.ld_d(r, o), self.x(dst),
                                               self.ld_d(r, self.off_plus(o, 8)))]
            r, o, _ = self.memref(st, dst, ins)
            return [self.st_d(r, o, self.x(src)), self.st_d(r, self.off_plus(o, 8), self.x(src) + 'h')]
        if mn == 'movq' and (isx(ops[0]) or isx(ops[1])):
            src, dst = ops
            if isx(src) and isx(dst):
                return ['%s = %s; %sh = 0;' % (self.x(dst), self.x(src), self.x(dst))]
            if isx(dst):
                if isreg(src):
                    return ['%s = u2d(%s); %sh = 0;' % (self.x(dst), self.reg_read(src[1:]), self.x(dst))]
                r, o, _ = self.memref(st, src, ins)
                return ['%s = %s; %sh = 0;' % (self.x(dst), self.ld_d(r, o), self.x(dst))]
            if isreg(dst):
                return [self.reg_write(dst[1:], 'd2u(%s)' % self.x(src))]
            r, o, _ = self.memref(st, dst, ins)
            return [self.st_d(r, o, self.x(src))]
        if mn in SSE_ARITH:
            src, dst = ops
            if isx(src):
                s = self.x(src)
            else:
                r, o, _ = self.memref(st, src, ins)
                s = self.ld_d(r, o)
            return ['%s = %s %s %s;' % (self.x(dst), self.x(dst), SSE_ARITH[mn], s)]
        if mn in ('sqrtsd', 'maxsd', 'minsd'):
            src, dst = ops
            if isx(src):
                s = self.x(src)
            else:
                r, o, _ = self.memref(st, src, ins)
                s = self.ld_d(r, o)
            d = self.x(dst)
            if mn == 'sqrtsd':
                return ['%s = LIFT_SQRT(%s);' % (d, s)]
            if mn == 'maxsd':
                return ['%s = (%s > %s) ? %s : %s;' % (d, d, s, d, s)]
            return ['%s = (%s < %s) ? %s : %s;' % (d, d, s, d, s)]
        if mn in ('andpd', 'andnpd', 'orpd', 'xorpd', 'pxor', 'andps', 'xorps', 'orps'):
            src, dst = ops
            d = self.x(dst)
            if isx(src) and src == dst and mn in ('xorpd', 'pxor', 'xorps'):
                return ['%s = 0; %sh = 0;' % (d, d)]
            if isx(src):
                s, sh = self.x(src), self.x(src) + 'h'
            else:
                r, o, _ = self.memref(st, src, ins)
                s, sh = self.ld_d(r, o), self.ld_d(r, self.off_plus(o, 8))
            f = {'andpd': 'd2u(%s) & d2u(%s)', 'andps': 'd2u(%s) & d2u(%s)', 'andnpd': '(~d2u(%s)) & d2u(%s)',
                 'orpd': 'd2u(%s) | d2u(%s)', 'orps': 'd2u(%s) | d2u(%s)', 'xorpd': 'd2u(%s) ^ d2u(%s)',
                 'xorps': 'd2u(%s) ^ d2u(%s)', 'pxor': 'd2u(%s) ^ d2u(%s)'}[mn]
            return ['%s = u2d(%s); %sh = u2d(%s);' % (d, f % (d, s), d, f % (d + 'h', sh))]
        if mn in ('cmplesd', 'cmpnlesd', 'cmpltsd', 'cmpnltsd', 'cmpeqsd', 'cmpneqsd', 'cmpunordsd'):
            src, dst = ops
            if isx(src):
                s = self.x(src)
            else:
                r, o, _ = self.memref(st, src, ins)
                s = self.ld_d(r, o)
            d = self.x(dst)
            c = {'cmplesd': '%s <= %s', 'cmpnlesd': '!(%s <= %s)', 'cmpltsd': '%s < %s',
                 'cmpnltsd': '!(%s < %s)', 'cmpeqsd': '%s == %s', 'cmpneqsd': '!(%s == %s)',
                 'cmpunordsd': '(%s != %s)'}[mn] % (d, s)
            if mn == 'cmpunordsd':
                c = '(%s != %s || %s != %s)' % (d, d, s, s)
            return ['%s = u2d((%s) ? ~0ULL : 0ULL);' % (d, c)]
        if mn in ('comisd', 'ucomisd'):
            src, dst = ops
            if isx(src):
                s = self.x(src)
            else:
                r, o, _ = self.memref(st, src, ins)
                s = self.ld_d(r, o)
            return ['fda = %s; fdb = %s;' % (self.x(dst), s)]
        if mn == 'unpcklpd':
            return ['%sh = %s;' % (self.x(ops[1]), self.x(ops[0]))]
        if mn.startswith('cvtsi2sd'):
            src, dst = ops
            if isreg(src):
                w = _SUB[src[1:]][1]
                v = self.sext(_SUB[src[1:]][0], w)
            else:
                w = 64 if mn.endswith('q') else 32
                r, o, _ = self.memref(st, src, ins)
                v = self.sext(self.ld_i(r, o, w), w)
            return ['%s = (double)(%s);' % (self.x(dst), v)]
        if mn == 'cvttsd2si':
            src, dst = ops
            if isx(src):
                s = self.x(src)
            else:
                r, o, _ = self.memref(st, src, ins)
                s = self.ld_d(r, o)
            w = _SUB[dst[1:]][1]
            return [self.reg_write(dst[1:], '(int64_t)(%s)' % s if w == 64 else '(int32_t)(%s)' % s)]
        # ---- integer ----
        if mn in ('mov', 'movq', 'movl', 'movabs'):
            src, dst = ops
            if isreg(dst):
                w = _SUB[dst[1:]][1]
                if isreg(src):
                    return [self.reg_write(dst[1:], self.reg_read(src[1:]))]
                if isimm(src):
                    return [self.reg_write(dst[1:], src[1:])]
                mem = L.parse_mem(src)
                aav = L.av_of_mem_addr(st, mem, ins)
                lav = L.load_av(st, aav, w)
                if lav[0] not in ('int', 'top') and lav[1] is not None and aav[0] not in ('STK',):
                    return [self.reg_write(dst[1:], '0x%x' % lav[1]) + ' /* -> %s */' % lav[0]]
                if aav[0] == 'FS':
                    return [self.reg_write(dst[1:], '0')]
                r, o, _ = self.memref(st, src, ins)
                return [self.reg_write(dst[1:], self.ld_i(r, o, w))]
            w = 32 if mn == 'movl' else 64
            if isreg(src):
                w = _SUB[src[1:]][1]
                v = self.reg_read(src[1:])
            else:
                v = src[1:]
                if w == 64 and int(v, 16) >= 0x80000000:
                    v = '(uint64_t)(int64_t)(int32_t)%s' % v
            r, o, _ = self.memref(st, dst, ins)
            return [self.st_i(r, o, v, w)]
        if mn == 'lea':
            mem = L.parse_mem(ops[0])
            av = L.av_of_mem_addr(st, mem, ins)
            if av[1] is not None and av[0] not in ('top',):
                return [self.reg_write(ops[1][1:], '0x%x' % (av[1] & 0xffffffffffffffff))]
            terms = []
            if mem['disp']:
                terms.append('(uint64_t)(int64_t)%d' % mem['disp'])
            if mem['base']:
                terms.append(_SUB[mem['base']][0])
            if mem['index']:
                terms.append('%s*%d' % (_SUB[mem['index']][0], mem['scale']))
            return [self.reg_write(ops[1][1:], ' + '.join(terms))]
        if mn == 'movslq':
            src, dst = ops
            if isreg(src):
                return [self.reg_write(dst[1:], self.sext(_SUB[src[1:]][0], 32))]
            r, o, _ = self.memref(st, src, ins)
            return [self.reg_write(dst[1:], self.sext(self.ld_i(r, o, 32), 32))]
        if mn == 'cltq':
            return ['rax = (uint64_t)(int64_t)(int32_t)rax;']
        if mn in ('add', 'sub', 'xor', 'and', 'or', 'imul', 'addq', 'subq', 'addl', 'subl') and len(ops) == 2:
            src, dst = ops
            base = mn.rstrip('lq') if mn not in ('imul',) else mn
            opc = {'add': '+', 'sub': '-', 'xor': '^', 'and': '&', 'or': '|', 'imul': '*'}[base]
            if isreg(dst):
                w = _SUB[dst[1:]][1]
                if dst == '%rsp':
                    if base == 'sub':
                        self.frame = max(self.frame, int(src[1:], 16))
                    return []
                if isreg(src):
                    s = self.reg_read(src[1:])
                elif isimm(src):
                    s = '(uint64_t)(int64_t)%d' % self.simm(src)
                else:
                    r, o, _ = self.memref(st, src, ins)
                    s = self.ld_i(r, o, w)
                if base == 'xor' and src == dst:
                    return [self.reg_write(dst[1:], '0'), 'fres = 0;']
                return [self.reg_write(dst[1:], '%s %s %s' % (self.reg_read(dst[1:]), opc, s)),
                        'fres = %s;' % self.sext(_SUB[dst[1:]][0], w)]
            w = 32 if mn.endswith('l') else 64
            r, o, _ = self.memref(st, dst, ins)
            s = self.reg_read(src[1:]) if isreg(src) else '(uint64_t)(int64_t)%d' % self.simm(src)
            if isreg(src):
                w = _SUB[src[1:]][1]
            return [self.st_i(r, o, '%s %s %s' % (self.ld_i(r, o, w), opc, s), w),
                    'fres = %s;' % self.sext(self.ld_i(r, o, w), w)]
        if mn in ('shl', 'shr', 'sar'):
            if len(ops) == 1:
                src, dst = '$0x1', ops[0]
            else:
                src, dst = ops
            w = _SUB[dst[1:]][1]
            n = self.simm(src)
            if mn == 'shl':
                e = '%s << %d' % (self.reg_read(dst[1:]), n)
            elif mn == 'shr':
                e = '(%s) >> %d' % (self.reg_read(dst[1:]), n)
            else:
                e = '(uint64_t)(%s >> %d)' % (self.sext(_SUB[dst[1:]][0], w), n)
            return [self.reg_write(dst[1:], e), 'fres = %s;' % self.sext(_SUB[dst[1:]][0], w)]
        if mn == 'btc':
            n = self.simm(ops[0])
            return [self.reg_write(ops[1][1:], '%s ^ (1ULL << %d)' % (self.reg_read(ops[1][1:]), n))]
        if mn in ('cmp', 'cmpl', 'cmpq', 'test'):
            src, dst = ops
            w = 64
            if isreg(dst):
                w = _SUB[dst[1:]][1]
                d = _SUB[dst[1:]][0]
            elif isreg(src):
                w = _SUB[src[1:]][1]
            elif mn == 'cmpl':
                w = 32
            if not isreg(dst):
                r, o, _ = self.memref(st, dst, ins)
                d = self.ld_i(r, o, w)
            if isreg(src):
                s = _SUB[src[1:]][0]
            elif isimm(src):
                s = '(uint64_t)(int64_t)%d' % self.simm(src)
            else:
                r, o, _ = self.memref(st, src, ins)
                s = self.ld_i(r, o, w)
            mask = {64: '(uint64_t)(%s)', 32: '(uint64_t)(uint32_t)(%s)', 16: '(uint64_t)(uint16_t)(%s)',
                    8: '(uint64_t)(uint8_t)(%s)'}[w]
            if mn == 'test':
                return ['fres = %s;' % self.sext('(%s) & (%s)' % (d, s), w)]
            return ['fua = %s; fub = %s; fsa = %s; fsb = %s; fres = %s;' % (
                mask % d, mask % s, self.sext(d, w), self.sext(s, w), self.sext('(%s) - (%s)' % (d, s), w))]
        if mn == 'rep_stos':
            av = st['r']['rdi']
            if av[0] in ('int', 'top'):
                raise ValueError('rep stos dest')
            region = av[0]
            self.regions_used.add(region)
            cnt = st['r']['rcx']
            if av[1] is not None and cnt[1] is not None and cnt[1] <= 64:
                outl = []
                for k_ in range(cnt[1]):
                    outl.append(self.st_i(region, '0x%x' % (av[1] + 8 * k_), 'rax', 64))
                outl.append('rdi += 8 * rcx; rcx = 0;')
                return outl
            if region == 'STK':
                self.stk_dynamic = True
            return ['{ uint64_t k_; for (k_ = 0; k_ < rcx; ++k_) { %s } rdi += 8 * rcx; rcx = 0; }' %
                    self.st_i(region, '(int64_t)(rdi + 8 * k_)', 'rax', 64)]
        raise ValueError('emit: unhandled instruction')

    @staticmethod
    def simm(op):
        v = int(op[1:], 16)
        if v >= 1 << 63:
            v -= 1 << 64
        elif op.startswith('$0xffffffff') and len(op) == 11:
            v -= 1 << 32
        return v

    def emit_call(self, ins, st):
        L = self.L
        if ins.target is None:
            # indirect call through a SimStruct method table
            m = re.match(r'^\*(0x[0-9a-f]+)\((%\w+)\)$', ins.raw)
            if not m:
                raise ValueError('indirect call form')
            av = st['r'][_SUB[m.group(2)[1:]][0]]
            if av[0] != 'SFUN' or av[1] is None or int(m.group(1), 16) != 0x310:
                raise ValueError('indirect call through %r' % (av,))
            return ['SFUN_CALL_%d();' % av[1]]
        name = ins.callname
        if 'stack_chk_fail' in name:
            return [';']
        key = name if '@plt' in name else ins.target
        spec = L.rules['calls'].get(key)
        if spec is None:
            raise ValueError('no call spec for %r' % (key,))
        cname, args, ret = spec
        # table lookups with compile-time-constant breakpoint/table addresses: emit the macro form so a
        # flavour can attach per-breakpoint-vector index caches (LIFT_IDX_DECL / LIFT_L2D / LIFT_L1D)
        base = cname.split('_rt_')[-1] if '_rt_' in cname else ''
        if base in ('Lookup2D_Normal', 'Lookup'):
            R = st['r']
            need = ['rdi', 'rdx', 'r8', 'rsi', 'rcx'] if base == 'Lookup2D_Normal' else ['rdi', 'rdx', 'rsi']
            if all(R[r][1] is not None for r in need) and all(R[r][0] == 'RO' for r in need if r in ('rdi', 'rdx', 'r8')):
                for r in ('rdi', 'rdx', 'r8'):
                    if r in need:
                        self.L.ro_min = min(getattr(self.L, 'ro_min', 1 << 62), R[r][1])
                if base == 'Lookup2D_Normal':
                    self.idx_vectors.add((R['rdi'][1], R['rsi'][1] & 0xffffffff))
                    self.idx_vectors.add((R['rdx'][1], R['rcx'][1] & 0xffffffff))
                    return ['x0 = LIFT_L2D(%s, 0x%x, %d, 0x%x, %d, 0x%x, x0, x1);' % (
                        cname, R['rdi'][1], R['rsi'][1] & 0xffffffff, R['rdx'][1], R['rcx'][1] & 0xffffffff, R['r8'][1])]
                self.idx_vectors.add((R['rdi'][1], R['rsi'][1] & 0xffffffff))
                return ['x0 = LIFT_L1D(%s, 0x%x, %d, x0, 0x%x);' % (cname, R['rdi'][1], R['rsi'][1] & 0xffffffff, R['rdx'][1])]
        al = []
        for reg, kind in args:
            if kind == 'ctx':
                al.append(reg)
            elif kind == 'f':
                al.append('x' + reg[3:])
            elif kind == 'i':
                al.append(self.sext(reg, 32))
            elif kind == 'l':
                al.append('(int64_t)' + reg)
            elif kind == 'p':
                al.append(self.ptr_expr(st['r'][reg], reg))
        call = '%s(%s)' % (cname, ', '.join(al))
        if ret == 'f':
            return ['x0 = %s;' % call]
        if ret == 'i':
            return ['rax = (uint64_t)(%s);' % call]
        return [call + ';']
