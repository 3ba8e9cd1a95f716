"""Tiny parser for the statement subset that tools/lift emits (gen/citation_<variant>.inc).

The lifted model is C with one statement per x86-64 instruction.  tools/dag re-reads that text to
recover the *dataflow graph* of one model evaluation (symex.py), from which the wave-parallel GPU
code is generated (codegen.py).  This module only tokenises / parses; it knows nothing about the model.

AST:  ('num', int) | ('id', name) | ('call', name, [args]) | ('cast', ctype, e) | ('un', op, e)
      | ('bin', op, a, b) | ('tern', c, a, b) | ('addr', e)
Statements: ('label', name) | ('goto', name) | ('if', cond, name) | ('assign', lhs, rhs)
            | ('addassign', lhs, rhs) | ('expr', e) | ('return', e|None)
"""
import re

CTYPES = {'uint64_t', 'uint32_t', 'uint16_t', 'uint8_t', 'int64_t', 'int32_t', 'int16_t', 'int8_t', 'double'}
_TOK = re.compile(r'\s*(?:(0x[0-9a-fA-F]+|\d+\.\d*(?:[eE][-+]?\d+)?|\d+)(ULL|LL|UL|U|L)?|([A-Za-z_]\w*)|'
                  r'(\|\||&&|<=|>=|==|!=|<<|>>|\+=|[-+*/&|^~!<>?:(),=]))')


def tokenize(s):
    out, pos = [], 0
    s = s.strip()
    while pos < len(s):
        m = _TOK.match(s, pos)
        if not m:
            raise SyntaxError('bad token at %r' % s[pos:pos + 30])
        if m.group(1) is not None:
            t = m.group(1)
            out.append(('num', int(t, 16) if t.startswith('0x') else (float(t) if '.' in t else int(t))))
        elif m.group(3) is not None:
            out.append(('id', m.group(3)))
        else:
            out.append(('op', m.group(4)))
        pos = m.end()
    return out


class P:
    def __init__(self, toks):
        self.t, self.i = toks, 0

    def peek(self, k=0):
        return self.t[self.i + k] if self.i + k < len(self.t) else ('eof', None)

    def eat(self, kind=None, val=None):
        tk = self.peek()
        if (kind and tk[0] != kind) or (val is not None and tk[1] != val):
            raise SyntaxError('expected %s %s, got %s (toks %s)' % (kind, val, tk, self.t[max(0, self.i - 5):self.i + 5]))
        self.i += 1
        return tk

    def isop(self, v):
        tk = self.peek()
        return tk[0] == 'op' and tk[1] == v

    # precedence climbing
    BIN = [['||'], ['&&'], ['|'], ['^'], ['&'], ['==', '!='], ['<', '>', '<=', '>='], ['<<', '>>'], ['+', '-'], ['*', '/']]

    def expr(self):
        c = self.binary(0)
        if self.isop('?'):
            self.eat()
            a = self.expr()
            self.eat('op', ':')
            b = self.expr()
            return ('tern', c, a, b)
        return c

    def binary(self, lvl):
        if lvl == len(self.BIN):
            return self.unary()
        a = self.binary(lvl + 1)
        while self.peek()[0] == 'op' and self.peek()[1] in self.BIN[lvl]:
            op = self.eat()[1]
            b = self.binary(lvl + 1)
            a = ('bin', op, a, b)
        return a

    def unary(self):
        tk = self.peek()
        if tk[0] == 'op' and tk[1] in ('-', '~', '!', '&', '+'):
            self.eat()
            e = self.unary()
            if tk[1] == '&':
                return ('addr', e)
            if tk[1] == '+':
                return e
            if tk[1] == '-' and e[0] == 'num':
                return ('num', -e[1])
            return ('un', tk[1], e)
        if tk[0] == 'op' and tk[1] == '(':
            n1, n2 = self.peek(1), self.peek(2)
            if n1[0] == 'id' and n1[1] in CTYPES and n2 == ('op', ')'):
                self.eat(); self.eat(); self.eat()
                return ('cast', n1[1], self.unary())
            if n1[0] == 'id' and n1[1] == 'double' and n2 == ('op', '*'):   # (double *)cmd
                self.eat(); self.eat(); self.eat(); self.eat('op', ')')
                return self.unary()
            self.eat()
            e = self.expr()
            self.eat('op', ')')
            return e
        if tk[0] == 'num':
            self.eat()
            return ('num', tk[1])
        if tk[0] == 'id':
            self.eat()
            if self.isop('('):
                self.eat()
                args = []
                if not self.isop(')'):
                    args.append(self.expr())
                    while self.isop(','):
                        self.eat()
                        args.append(self.expr())
                self.eat('op', ')')
                return ('call', tk[1], args)
            return ('id', tk[1])
        raise SyntaxError('unexpected %s' % (tk,))


def parse_expr(s):
    p = P(tokenize(s))
    e = p.expr()
    if p.peek()[0] != 'eof':
        raise SyntaxError('trailing tokens in %r' % s)
    return e


_CMT = re.compile(r'/\*.*?\*/')


def parse_statements(lines):
    """lines of a function body -> list of statements"""
    out = []
    for ln in lines:
        if '__stack_chk_fail' in ln:          # noreturn; the canary compare that guards it folds to false
            out.append(('return', None))
            continue
        ln = _CMT.sub('', ln).strip()
        if not ln:
            continue
        m = re.match(r'^(L_\w+):\s*;$', ln)
        if m:
            out.append(('label', m.group(1)))
            continue
        if re.match(r'^(uint64_t|int64_t|double|const double|\(void\)|LIFT_IDX_DECL)', ln):
            continue
        for st in [s.strip() for s in ln.split(';')]:
            if not st:
                continue
            m = re.match(r'^goto (L_\w+)$', st)
            if m:
                out.append(('goto', m.group(1)))
                continue
            m = re.match(r'^if \((.*)\) goto (L_\w+)$', st)
            if m:
                out.append(('if', parse_expr(m.group(1)), m.group(2)))
                continue
            m = re.match(r'^return\b\s*(.*)$', st)
            if m:
                out.append(('return', parse_expr(m.group(1)) if m.group(1) else None))
                continue
            toks = tokenize(st)
            # find top-level '=' or '+='
            depth, k = 0, None
            for i, tk in enumerate(toks):
                if tk == ('op', '('):
                    depth += 1
                elif tk == ('op', ')'):
                    depth -= 1
                elif depth == 0 and tk[0] == 'op' and tk[1] in ('=', '+='):
                    k = i
                    break
            if k is None:
                p = P(toks)
                out.append(('expr', p.expr()))
                continue
            pl, pr = P(toks[:k]), P(toks[k + 1:])
            lhs, rhs = pl.expr(), pr.expr()
            out.append(('assign' if toks[k][1] == '=' else 'addassign', lhs, rhs))
    return out


def split_functions(text):
    """-> {name: [body lines]} for the LIFT_FN_* functions, plus the header lines before the first one"""
    lines = text.split('\n')
    fns, header, i = {}, [], 0
    while i < len(lines):
        m = re.match(r'^LIFT_FN_(\w+) .*\)$', lines[i])
        if m and i + 1 < len(lines) and lines[i + 1].startswith('{'):
            j = i + 2
            body = []
            while not lines[j].startswith('}'):
                body.append(lines[j])
                j += 1
            fns[m.group(1)] = body
            i = j + 1
        else:
            if not fns:
                header.append(lines[i])
            i += 1
    return header, fns
