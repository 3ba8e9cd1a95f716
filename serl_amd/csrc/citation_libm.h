// citation_libm.h -- sincos / tan / pow of the model evaluation (product code; every kernel family calls these, so that the families
// stay bit-identical with each other).
//
// The reference binary calls glibc's sin / cos / tan / pow; any GPU libm differs from glibc in the last bit
// here and there, and the parity bar (1e-5 on the episodic return, 1e-9 on open-loop states over 3 000 steps) is far above that.
// ROCm's ocml bodies are general-purpose: sincos 130 instructions (Payne-Hanek fall-back, three-part pi/2), tan 140, pow 330
// (extended-precision log2 + exp2 for ANY base and exponent) -- 18 % of the instructions of a model evaluation, and they sit at the
// head of the hand-over chains every wavefront of a team waits for (profiles/r03_experiments.md: both gone = -2.2 us per env step).
// The model's arguments are flight angles (|x| < a few radians) and ONE power with a literal exponent of a temperature ratio in
// (0.7, 1.1], so the bodies below are short:
//
//   citw_sincos   k = rint(x 2/pi);  r = x - k pi/2 with pi/2 = HI + LO (two fma: exact product, 106-bit pi/2: good for |x| < 1e5);
//                 fdlibm's kernel polynomials on |r| <= pi/4 (degree 13 / 14 minimax, error < 2^-58); quadrant by sign-bit xor.
//                 ~40 instructions; measured against 80-bit long double on 2e6 arguments per range (tests/test_libm.py): <= 1.03 ulp on
//                 |x| <= 0.8, <= 1.45 ulp up to |x| = 300 (the reduced argument's rounding error is not carried along), mean 0.29 ulp.
//                 |x| >= 1e5, NaN: ocml's sincos (cold path).
//   citw_tan      sin / cos of the same reduction (one IEEE division): <= 2.8 ulp.
//   citw_pow      x^c = exp(c ln x),  ln x = 2 atanh(q), q = (x-1)/(x+1) with its rounding error carried along (q_lo), atanh by its
//                 Taylor series through q^21 (|q| <= 0.17: x in [0.71, 1.41] -- the troposphere branch of the ISA atmosphere that
//                 the model guards the call with), c ln x as head + tail, exp by k = rint(y / ln 2), Taylor through r^13 / 13!.
//                 ~110 instructions, <= 1.13 ulp on [0.71, 1.41] (mean 0.27); any other base takes ocml's pow (cold path: the model never
//                 gets there while its guard holds).
//
// + - x fma only besides one division each in tan / pow; explicit fma (the build's -ffp-contract=off only forbids the compiler to fuse
// on its own).  __host__ __device__: tests/test_libm.py compiles the same text for the CPU.
#pragma once
#ifndef __HIPCC__
#ifndef _GNU_SOURCE
#define _GNU_SOURCE 1
#endif
#include <math.h>
#endif
#ifndef CITW_LIBM_FN
#ifdef __HIPCC__
#define CITW_LIBM_FN static __host__ __device__ __forceinline__
#else
#define CITW_LIBM_FN static inline
#endif
#endif

// The coefficients as registers (round 5, team kernels; citation_wave.h "f64 literals ... in registers"): CITW_LK(i, literal) is slot
// KB + i of the caller's register set when it passed one (HAVE_K, a compile-time fact after inlining; KB = where this function's
// block starts in the role's set, chosen by tools/dag/codegen_team.py, which reads the blocks off this text) and the literal itself
// otherwise -- in the CPU build always.  Slots count from 0 per function: citw_sincos 0 .. 14, citw_pow 0 .. 23.
#ifdef __HIPCC__
#include "serl_kregs.h"
#define CITW_LIBM_KPARAMS , const bool HAVE_K = false, const CitwKRegs &KR = citw_no_kregs, const int KB = 0
#define CITW_LIBM_KARGS , HAVE_K, KR, KB
#define CITW_LK(i, lit) (HAVE_K ? KR.k[KB + (i)] : (lit))
#else
#define CITW_LIBM_KPARAMS
#define CITW_LIBM_KARGS
#define CITW_LK(i, lit) (lit)
#endif

CITW_LIBM_FN double citw_libm_hi_xor(double v, unsigned bits)
{
  union { double d; unsigned long long u; } t;
  t.d = v;
  t.u ^= (unsigned long long)bits << 32;
  return t.d;
}

// The general-purpose bodies behind the range guards below, OUT OF LINE on the device when CITW_LIBM_COLD_CALLS is set (the lane-group team kernels: 1.2 KB of
// ocml sincos at each of four call sites, 1.7 KB of pow, in the middle of a loop that has to live in a 64 KB instruction cache shared by two CUs; a guard
// that fails -- never on flight angles -- pays a call)
#if defined(__HIPCC__) && defined(CITW_LIBM_COLD_CALLS) && defined(__HIP_DEVICE_COMPILE__)
static __device__ __attribute__((noinline, cold)) void citw_general_sincos(const double x, double *s, double *c) { sincos(x, s, c); }
static __device__ __attribute__((noinline, cold)) double citw_general_pow(const double x, const double c) { return pow(x, c); }
static __device__ __attribute__((noinline, cold)) double citw_general_exp(const double x) { return exp(x); }
static __device__ __attribute__((noinline, cold)) double citw_general_log10(const double x) { return log10(x); }
static __device__ __attribute__((noinline, cold)) double citw_general_log(const double x) { return log(x); }
#else
#define citw_general_sincos sincos
#define citw_general_pow pow
#endif

// sin and cos of x
CITW_LIBM_FN void citw_sincos(const double x, double *s, double *c CITW_LIBM_KPARAMS)
{
#ifndef CITW_LIBM_NO_FALLBACK               // (tools/isa/role_isa.py counts the instructions of the short bodies without the cold paths)
  if (__builtin_expect(!(__builtin_fabs(x) < 1.0e5), 0)) {      // huge, inf, NaN: the general-purpose body (beyond 1e5 the two-part pi/2 loses accuracy gradually)
    citw_general_sincos(x, s, c);
    return;
  }
#endif
  const double TWO_OVER_PI = CITW_LK(0, 0x1.45f306dc9c883p-1);                                   // 2/pi
  const double PIO2_HI = CITW_LK(1, 0x1.921fb54442d18p+0), PIO2_LO = CITW_LK(2, 0x1.1a62633145c07p-54);      // pi/2 = HI + LO (+ 3e-33)
  const double S1 = CITW_LK(3, -0x1.5555555555549p-3), S2 = CITW_LK(4, 0x1.111111110f8a6p-7), S3 = CITW_LK(5, -0x1.a01a019c161d5p-13),
               S4 = CITW_LK(6, 0x1.71de357b1fe7dp-19), S5 = CITW_LK(7, -0x1.ae5e68a2b9cebp-26), S6 = CITW_LK(8, 0x1.5d93a5acfd57cp-33);
  const double C1 = CITW_LK(9, 0x1.555555555554cp-5), C2 = CITW_LK(10, -0x1.6c16c16c15177p-10), C3 = CITW_LK(11, 0x1.a01a019cb159p-16),
               C4 = CITW_LK(12, -0x1.27e4f809c52adp-22), C5 = CITW_LK(13, 0x1.1ee9ebdb4b1c4p-29), C6 = CITW_LK(14, -0x1.8fae9be8838d4p-37);
  const double kd = __builtin_rint(x * TWO_OVER_PI);
  double r = __builtin_fma(-kd, PIO2_HI, x);           // exact: |r| < 1 is a multiple of ulp(x) or of 2^-52
  r = __builtin_fma(-kd, PIO2_LO, r);
  const int n = (int)kd;
  const double z = r * r;
  // sin r = r + r^3 (S1 + z (S2 + ... z S6))
  double ps = __builtin_fma(z, S6, S5);
  ps = __builtin_fma(z, ps, S4);
  ps = __builtin_fma(z, ps, S3);
  ps = __builtin_fma(z, ps, S2);
  const double v = z * r;
  const double sr = __builtin_fma(v, __builtin_fma(z, ps, S1), r);
  // cos r = 1 - z/2 + z^2 (C1 + z (C2 + ... z C6)), the leading terms summed with their rounding error (fdlibm __kernel_cos)
  double pc = __builtin_fma(z, C6, C5);
  pc = __builtin_fma(z, pc, C4);
  pc = __builtin_fma(z, pc, C3);
  pc = __builtin_fma(z, pc, C2);
  pc = __builtin_fma(z, pc, C1);
  const double hz = 0.5 * z, w = 1.0 - hz;
  const double cr = w + (((1.0 - w) - hz) + (z * z) * pc);
  // quadrant: n mod 4 = 0: (s, c); 1: (c, -s); 2: (-s, -c); 3: (-c, s)
  const int swap = n & 1;
  const double s0 = swap ? cr : sr, c0 = swap ? sr : cr;
  *s = citw_libm_hi_xor(s0, ((unsigned)n & 2u) << 30);
  *c = citw_libm_hi_xor(c0, (((unsigned)n + 1u) & 2u) << 30);
}

CITW_LIBM_FN double citw_sin(const double x CITW_LIBM_KPARAMS) { double s, c; citw_sincos(x, &s, &c CITW_LIBM_KARGS); return s; }
CITW_LIBM_FN double citw_cos(const double x CITW_LIBM_KPARAMS) { double s, c; citw_sincos(x, &s, &c CITW_LIBM_KARGS); return c; }

CITW_LIBM_FN double citw_tan(const double x CITW_LIBM_KPARAMS)
{
  double s, c;
  citw_sincos(x, &s, &c CITW_LIBM_KARGS);
  return s / c;
}

// x^c: the short body for a base in [0.71, 1.41] and |c| <= 16 (the model's one call: a temperature ratio to the power 4.256), the
// general-purpose one for anything else
CITW_LIBM_FN double citw_pow(const double x, const double c CITW_LIBM_KPARAMS)
{
#ifndef CITW_LIBM_NO_FALLBACK
  if (__builtin_expect(!(x >= 0.71 && x <= 1.41 && __builtin_fabs(c) <= 16.0), 0)) return citw_general_pow(x, c);      // (expected cold: block placement moves the general body behind the loop)
#endif
  const double LN2_HI = CITW_LK(0, 0x1.62e42fefa39efp-1), LN2_LO = CITW_LK(1, 0x1.abc9e3b39803fp-56), INV_LN2 = CITW_LK(2, 0x1.71547652b82fep+0);
  // ln x = 2 q + q z (2/3 + z (2/5 + ... )), q = f / (2 + f), f = x - 1 (exact for x in [1/2, 2])
  const double f = x - 1.0, d = 2.0 + f;
  const double q = f / d;
  const double q_lo = __builtin_fma(-q, d, f) / d;                     // q + q_lo = f / d to ~2^-104 (d = 2 + f is exact within [1/2, 2] up to one rounding: carried below)
  const double d_lo = (2.0 - d) + f;                                   // d + d_lo = 2 + f exactly
  const double qc = q_lo - q * (d_lo / d);                              // the quotient's tail with the divisor's rounding error taken out
  const double z = q * q;
  double p = __builtin_fma(z, CITW_LK(3, 0x1.8618618618618p-4), CITW_LK(4, 0x1.af286bca1af28p-4));     // 2/21, 2/19
  p = __builtin_fma(z, p, CITW_LK(5, 0x1.e1e1e1e1e1e1ep-4));                        // 2/17
  p = __builtin_fma(z, p, CITW_LK(6, 0x1.1111111111111p-3));                        // 2/15
  p = __builtin_fma(z, p, CITW_LK(7, 0x1.3b13b13b13b14p-3));                        // 2/13
  p = __builtin_fma(z, p, CITW_LK(8, 0x1.745d1745d1746p-3));                        // 2/11
  p = __builtin_fma(z, p, CITW_LK(9, 0x1.c71c71c71c71cp-3));                        // 2/9
  p = __builtin_fma(z, p, CITW_LK(10, 0x1.2492492492492p-2));                        // 2/7
  p = __builtin_fma(z, p, CITW_LK(11, 0x1.999999999999ap-2));                        // 2/5
  p = __builtin_fma(z, p, CITW_LK(12, 0x1.5555555555555p-1));                        // 2/3
  const double ln_hi = 2.0 * q;                                         // exact
  const double ln_lo = __builtin_fma(q * z, p, 2.0 * qc);
  // y = c ln x as head + tail
  const double yh = c * ln_hi;
  const double yl = __builtin_fma(c, ln_hi, -yh) + c * ln_lo;
  // exp(yh + yl) = 2^k exp(r), r = yh - k ln 2 (+ yl)
  const double kd = __builtin_rint(yh * INV_LN2);
  double r = __builtin_fma(-kd, LN2_HI, yh);
  r = __builtin_fma(-kd, LN2_LO, r) + yl;
  double e = __builtin_fma(r, CITW_LK(13, 0x1.6124613a86d09p-33), CITW_LK(14, 0x1.1eed8eff8d898p-29));   // 1/13!, 1/12!
  e = __builtin_fma(r, e, CITW_LK(15, 0x1.ae64567f544e4p-26));                       // 1/11!
  e = __builtin_fma(r, e, CITW_LK(16, 0x1.27e4fb7789f5cp-22));                       // 1/10!
  e = __builtin_fma(r, e, CITW_LK(17, 0x1.71de3a556c734p-19));                       // 1/9!
  e = __builtin_fma(r, e, CITW_LK(18, 0x1.a01a01a01a01ap-16));                       // 1/8!
  e = __builtin_fma(r, e, CITW_LK(19, 0x1.a01a01a01a01ap-13));                       // 1/7!
  e = __builtin_fma(r, e, CITW_LK(20, 0x1.6c16c16c16c17p-10));                       // 1/6!
  e = __builtin_fma(r, e, CITW_LK(21, 0x1.1111111111111p-7));                        // 1/5!
  e = __builtin_fma(r, e, CITW_LK(22, 0x1.5555555555555p-5));                        // 1/4!
  e = __builtin_fma(r, e, CITW_LK(23, 0x1.5555555555555p-3));                        // 1/3!
  e = __builtin_fma(r, e, 0.5);
  const double em1 = __builtin_fma(r * r, e, r);                        // exp(r) - 1
  return __builtin_ldexp(1.0 + em1, (int)kd);
}

// atan x (the gust / test builds' turbulence model; round 5): the published fdlibm algorithm (s_atan.c) -- argument reduction to
// |t| < 7/16 around 0, atan(1/2), atan(1), atan(3/2), infinity (one division), an odd/even split of a degree-11 minimax polynomial in
// t^2, atan(breakpoint) added as hi + lo.  + - x / only, no fused operation: the CPU build and the kernels return the same bits (the
// same-libm flavour of the oracle relies on that), < 1 ulp.
CITW_LIBM_FN double citw_atan(const double x0)
{
  const double AT0 = 0x1.555555555550dp-2, AT1 = -0x1.999999998ebc4p-3, AT2 = 0x1.24924920083ffp-3, AT3 = -0x1.c71c6fe231671p-4,
               AT4 = 0x1.745cdc54c206ep-4, AT5 = -0x1.3b0f2af749a6dp-4, AT6 = 0x1.10d66a0d03d51p-4, AT7 = -0x1.dde2d52defd9ap-5,
               AT8 = 0x1.97b4b24760debp-5, AT9 = -0x1.2b4442c6a6c2fp-5, AT10 = 0x1.0ad3ae322da11p-6;
  if (x0 != x0) return x0;
  const double ax = __builtin_fabs(x0);
  if (ax >= 0x1.0p+66) return x0 < 0.0 ? -(0x1.921fb54442d18p+0 + 0x1.1a62633145c07p-54) : (0x1.921fb54442d18p+0 + 0x1.1a62633145c07p-54);
  double t, hi = 0.0, lo = 0.0;
  int id = -1;
  if (ax < 0.4375) {
    if (ax < 0x1.0p-29) return x0;
    t = x0;
  } else if (ax < 1.1875) {
    if (ax < 0.6875) { id = 0; t = (2.0 * ax - 1.0) / (2.0 + ax); hi = 0x1.dac670561bb4fp-2; lo = 0x1.a2b7f222f65e2p-56; }
    else { id = 1; t = (ax - 1.0) / (ax + 1.0); hi = 0x1.921fb54442d18p-1; lo = 0x1.1a62633145c07p-55; }
  } else {
    if (ax < 2.4375) { id = 2; t = (ax - 1.5) / (1.0 + 1.5 * ax); hi = 0x1.f730bd281f69bp-1; lo = 0x1.007887af0cbbdp-56; }
    else { id = 3; t = -1.0 / ax; hi = 0x1.921fb54442d18p+0; lo = 0x1.1a62633145c07p-54; }
  }
  const double z = t * t, w = z * z;
  const double s1 = z * (AT0 + w * (AT2 + w * (AT4 + w * (AT6 + w * (AT8 + w * AT10)))));
  const double s2 = w * (AT1 + w * (AT3 + w * (AT5 + w * (AT7 + w * AT9))));
  if (id < 0) return t - t * (s1 + s2);
  const double r = hi - ((t * (s1 + s2) - lo) - t);
  return x0 < 0.0 ? -r : r;
}

#ifdef __HIPCC__
// x / c for a literal c, rc = RN(1 / c): correctly rounded for every finite x whose quotient is a normal number -- proved per
// divisor by tools/dag/constdiv.py (error of q + r rc against x / c below 2^-104; the finitely many x whose quotient lies that
// close to a rounding boundary are enumerated and checked exactly), so the result is the IEEE quotient the reference computes.
// v_div_fixup_f64 restores the IEEE result for zeros (sign), infinities and NaN.  4 instructions, ~30 dependent cycles
// (IEEE division: 13 and ~71).  fma is explicit here; -ffp-contract=off only forbids the compiler to fuse on its own.
static __device__ __forceinline__ double citw_div_const(const double x, const double c, const double rc)
{
  const double q = x * rc;
  const double r = __builtin_fma(-q, c, x);
  const double q2 = __builtin_fma(r, rc, q);
  return __builtin_amdgcn_div_fixup(q2, c, x);
}
#endif
