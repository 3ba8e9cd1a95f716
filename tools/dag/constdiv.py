#!/usr/bin/env python3
"""Division by a compile-time constant in three operations, with a per-divisor PROOF of correct rounding.

The model divides by literals (288.15, 48.5, 1.225 ...): 17 of the 34 divisions of an evaluation.  The compiler's IEEE
division costs 13 instructions and ~71 dependent cycles; with y = RN(1/c) precomputed

    q  = RN(x * y)                 v_mul_f64
    r  = RN(x - q * c)             v_fma_f64     (one rounding)
    q' = RN(q + r * y)             v_fma_f64     (one rounding)        [+ v_div_fixup_f64 for zeros / infinities / NaN]

is 4 instructions and ~30 cycles.  It returns the correctly rounded quotient RN(x / c) for EVERY finite x whose quotient
neither overflows nor falls into the subnormal range (|x| in [2^-960, 2^960] for the divisors of this model), provided the
divisor passes the check below -- divisors that do not pass keep the IEEE division.

Why.  Write Q = x / c, y = (1 + e1) / c with |e1| <= 2^-53 (y is correctly rounded), q = Q (1 + e1)(1 + e2), |e2| <= 2^-53,
so r_exact = x - q c = c (Q - q) has |r_exact / c| <= (2^-52 + 2^-106) |Q|.  The fma delivers r = r_exact (1 + e3), |e3| <= 2^-53
(e3 = 0 when r_exact is representable, the usual case).  The value the last fma rounds is
    q + r y = Q + r_exact (y - 1/c) + r_exact y e3  =  Q (1 + eta),   |eta| <= 2 (1 + 2^-52) 2^-53 (2^-52 + 2^-106)  <  2^-104.
RN of that value equals RN(Q) unless a rounding boundary -- a midpoint mu between two adjacent doubles -- lies between Q and
Q (1 + eta).  With integer significands x = X 2^ex, c = C 2^ec (2^52 <= X, C < 2^53) the quotient's significand is
M = floor(X / C * 2^k') with k' = 52 (X >= C) or 53 (X < C), the midpoints are (2M + 1) / 2, and
    |Q - mu| / |Q| = |2^(k'+1) X - (2M + 1) C| / (2^(k'+1) X)  =  d / (2^(k'+1) X)  >  d 2^-107        (d a positive integer;
d = 0 is impossible: a quotient of two 53-bit numbers is never a midpoint).  Hence every x with d > 8 is safe by the bound, and
the x with 1 <= d <= 8 -- finitely many per divisor, found by solving the linear congruence 2^(k'+1) X = d (mod C) -- are
checked one by one in exact rational arithmetic (`verify`, margin d <= 64).  The result does not depend on the exponents or
the signs (no overflow / underflow by assumption; round-to-nearest is symmetric), so X in [2^52, 2^53) covers every x.
v_div_fixup_f64 restores the IEEE result for x = +-0, +-inf, NaN.
"""
import math
from fractions import Fraction

D_MAX = 64          # candidates with |2^(k'+1) X - (2M+1) C| <= D_MAX are checked exactly (the bound needs d <= 8)


def _rn(fr):
    """round-to-nearest-even of an exact rational to a double (int / int true division is correctly rounded in CPython)"""
    return fr.numerator / fr.denominator


def recip(c):
    return _rn(Fraction(1) / Fraction(c))


def algo(x, c, y):
    """the three-operation sequence, each operation rounded once (exact rational arithmetic in between)"""
    q = x * y                                               # RN(x * y): IEEE double multiplication
    r = _rn(Fraction(x) - Fraction(q) * Fraction(c))        # fma(-q, c, x)
    return _rn(Fraction(q) + Fraction(r) * Fraction(y))     # fma(r, y, q)


def candidates(c, d_max=D_MAX):
    """all significands X in [2^52, 2^53) whose quotient X / C lies within d_max units of a midpoint"""
    m, e = math.frexp(abs(c))
    C = int(m * (1 << 53))                                  # 2^52 <= C < 2^53
    assert Fraction(C, 1 << 53) * Fraction(2) ** e == Fraction(abs(c))
    out = []
    for kp1 in (53, 54):                                    # X >= C: 2^53 X = (2M+1) C + d ;  X < C: 2^54 X = (2M+1) C + d
        a = 1 << kp1
        g = math.gcd(a, C)
        a2, b2 = a // g, C // g
        inv = pow(a2 % b2, -1, b2) if b2 > 1 else 0
        for d in range(-d_max, d_max + 1):
            if d == 0 or d % g:
                continue
            x0 = ((d // g) * inv) % b2 if b2 > 1 else 0
            lo, hi = (C, 1 << 53) if kp1 == 53 else (1 << 52, C)      # X >= C  /  X < C
            X = x0 + ((lo - x0 + b2 - 1) // b2) * b2
            while X < hi:
                num = a * X - d
                if num % C == 0:
                    Y = num // C
                    if Y & 1 and (1 << 53) <= Y < (1 << 54):
                        out.append((X, d))
                X += b2
    return out


def verify(c, d_max=D_MAX):
    """-> (ok, number of near-midpoint candidates checked).  ok: the sequence returns RN(x / c) on every candidate."""
    c = float(c)
    if c == 0.0 or math.isinf(c) or math.isnan(c):
        return False, 0
    y = recip(c)
    cand = candidates(c, d_max)
    for X, d in cand:
        x = float(X)
        if algo(x, c, y) != x / c:
            return False, len(cand)
    return True, len(cand)


def spot_check(c, n=20000, seed=1):
    """random significands and exponents through the exact emulation (a sanity check of the emulation itself)"""
    import random
    rng = random.Random(seed)
    y = recip(c)
    for _ in range(n):
        x = math.ldexp(rng.randrange(1 << 52, 1 << 53), rng.randrange(-200, 200)) * rng.choice((1.0, -1.0))
        if algo(x, c, y) != x / c:
            return False
    return True


if __name__ == '__main__':
    import sys, struct
    sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
    import build_dag
    for variant in (sys.argv[1:] or ['nominal']):
        g, res, _ = build_dag.build(variant, fast_zero=True)
        consts = sorted({g.nodes[g.nodes[n][2]][1] for n in range(len(g.nodes)) if g.nodes[n][0] == 'div' and g.nodes[g.nodes[n][2]][0] == 'cf'})
        for bits in consts:
            c = struct.unpack('<d', struct.pack('<Q', bits))[0]
            ok, n = verify(c)
            print(variant, float.hex(c), c, 'ok' if ok else 'KEEP IEEE DIVISION', 'candidates', n, 'spot', spot_check(c, 2000))
