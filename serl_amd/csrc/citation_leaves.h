// citation_leaves.h -- hand-written leaf routines of the Citation model for the GPU (product code).
//
// The lifted model body (gen/citation_<variant>.inc) calls four kinds of leaf routines.  Executed the
// way the reference executes them they dominate the instruction count of a model evaluation
// (123 sincos and 144 binary searches per evaluation), so they are restated by hand here,
// keeping the reference's arithmetic order so results stay bit-identical:
//
//   cit_lookup_index   rt_GetLookupIndex @0xf470  -- same interval, found by a branch-free count over the
//                      (<= 22 entry, strictly increasing -- asserted by tools/lift) breakpoint vector instead
//                      of a binary search: the loads are independent, so one LDS latency instead of five
//   cit_lookup1d       rt_Lookup @0xf530          (y1-y0)/(x1-x0)*(u-x0)+y0
//   cit_lookup2d       rt_Lookup2D_Normal @0xf590 column-major z[ix + nx*iy], interpolate x then y
//   cit_axes_*         ac_axes mdlOutputs @0x103e0 + matmultiply @0x103a0: the 23 frame transformations of one
//                      model evaluation all receive the same five angles (alpha beta phi theta psi), so
//                      the trigonometry and the three rotation matrices are built once per evaluation
//                      (cit_axes_prepare) and every call only applies them; each call still checks its
//                      angles bit-for-bit against the prepared ones and raises CIT_ERR_AXES_KEY otherwise.
//
// The header is written against CIT_HD (= __device__ __forceinline__ on the GPU) so that
// tests/test_leaves_host.py can also compile it for the host and compare it with the reference library.
#pragma once
#ifndef CIT_HD
#define CIT_HD static __device__ __forceinline__
#endif
#ifndef CIT_TBL
#define CIT_TBL const double *   // pointer type of breakpoint / table data
#endif

#define CIT_ERR_AXES_KEY 1

struct CitAxes {
  double key[5];   // alpha beta phi theta psi the matrices were built from
  double M1[9];    // wind  -> stability (beta):   [c5 -s5 0; s5 c5 0; 0 0 1]
  double M2[9];    // stab. -> body      (alpha):  [c4 0 -s4; 0 1 0; s4 0 c4]
  double M3[9];    // body  -> earth     (phi,theta,psi)
};

CIT_HD uint64_t cit_bits(double d) { union { double d; uint64_t u; } v; v.d = d; return v.u; }

// interval index of u in x[0..n-1] (rt_GetLookupIndex semantics):
//   u <= x[0] -> 0 ; u >= x[n-1] -> n-2 ; u < 0: x[i] <= u < x[i+1] ; u >= 0: x[i] < u <= x[i+1]
CIT_HD int cit_lookup_index(CIT_TBL x, int n, double u)
{
  int lt = 0, le = 0;
  for (int i = 0; i < n; ++i) { double v = x[i]; lt += (v < u) ? 1 : 0; le += (v <= u) ? 1 : 0; }
  int idx = ((u < 0.0) ? le : lt) - 1;
  idx = idx < 0 ? 0 : idx;
  return idx > n - 2 ? n - 2 : idx;
}

// out-of-line copy of the index search (one instance in the code object; reached only on a cache miss)
#ifndef CIT_NOINLINE
#define CIT_NOINLINE static __device__ __noinline__
#endif
CIT_NOINLINE int cit_lookup_index_slow(CIT_TBL x, int n, double u) { return cit_lookup_index(x, n, u); }

// ... with a HINT (round 6, the lane-per-episode kernels): the interval an input fell into in the previous model evaluation.  An input almost never leaves its
// interval between two evaluations (4 of 2 400 per episode), and idx == h holds exactly when (h == 0 or x[h] (<, <=) u) and (h == n - 2 or not x[h + 1] (<, <=) u)
// -- two loads and two compares instead of 2 n of each; the interval is unique, so a verified hint IS the count's result.  If any lane of the wavefront fails
// the test, the wavefront runs the count (which gives the other lanes what they had).
#ifdef __HIPCC__
// The hints are PACKED, five bits each (an interval index is below 31: the longest breakpoint vector has 22 entries), six to a word: 46 hints of an evaluation live
// in 8 registers instead of 46 across the whole straight-line evaluation (every register the evaluation body does not hold is a spill less: a reload costs the
// lane-per-episode kernels ~200 cycles, there is one wavefront per SIMD).
CIT_HD int cit_lookup_index_h(CIT_TBL x, int n, double u, uint32_t &word, const int shift)
{
  int h = (int)((word >> shift) & 31u);
  h = h > n - 2 ? n - 2 : h;
  const double xl = x[h], xh = x[h + 1];
  const bool neg = u < 0.0;
  const bool below = neg ? (xl <= u) : (xl < u);
  const bool above = neg ? (xh <= u) : (xh < u);
  const bool ok = ((h == 0) || below) && ((h == n - 2) || !above);
  if (__builtin_expect(__ballot(!ok) != 0ULL, 0)) {
    h = cit_lookup_index_slow(x, n, u);      // (out of line: 41 inlined counts would add 40 KB to an evaluation body that is at the instruction cache's size already)
    word = (word & ~(31u << shift)) | ((uint32_t)h << shift);
  }
  return h;
}
#endif

// index with a one-entry cache per breakpoint vector (cu/ci are locals of the model function): within one
// model evaluation most of the 144 searches repeat an earlier (vector, input) pair (41 distinct, nominal)
CIT_HD int cit_lookup_index_cached(CIT_TBL x, int n, double u, double *cu, int *ci)
{
  if (*ci >= 0 && cit_bits(*cu) == cit_bits(u)) return *ci;
  const int i = cit_lookup_index_slow(x, n, u);
  *cu = u; *ci = i;
  return i;
}

CIT_HD double cit_lookup1d_at(CIT_TBL x, int i, double u, CIT_TBL y)
{
  const double x0 = x[i], x1 = x[i + 1], y0 = y[i], y1 = y[i + 1];
  double r = y1 - y0;
  r = r / (x1 - x0);
  r = r * (u - x0);
  return r + y0;
}

CIT_HD double cit_lookup2d_at(CIT_TBL xr, int nr, CIT_TBL xc, CIT_TBL z, int ix, int iy, double u0, double u1)
{
  const double x0 = xr[ix], x1 = xr[ix + 1];
  const double dx = x1 - x0, wx = u0 - x0;
  const double z00 = z[ix + nr * iy], z10 = z[ix + 1 + nr * iy];
  const double z01 = z[ix + nr * (iy + 1)], z11 = z[ix + 1 + nr * (iy + 1)];
  double a = z10 - z00; a = a / dx; a = a * wx; a = a + z00;
  double b = z11 - z01; b = b / dx; b = b * wx; b = b + z01;
  const double y0 = xc[iy];
  const double dy = xc[iy + 1] - y0;
  double r = b - a; r = r / dy; r = r * (u1 - y0);
  return r + a;
}

// ... with the x-direction quotients precomputed per table and interval (round 6, the lane-per-episode kernels: g_sl, staged beside the tables by
// rollout_variant.inc with the operations above -- s[ix + (nr - 1) iy] = (z[ix + 1][iy] - z[ix][iy]) / (xr[ix + 1] - xr[ix]), 1-D: s[i] = (y[i + 1] - y[i]) / (x[i + 1] - x[i])):
// the same values enter the same multiply-adds, the interpolation keeps the one division whose dividend depends on the inputs
CIT_HD double cit_lookup1d_at_s(CIT_TBL x, int i, double u, CIT_TBL y, CIT_TBL s)
{
  double r = s[i];
  r = r * (u - x[i]);
  return r + y[i];
}

CIT_HD double cit_lookup2d_at_s(CIT_TBL xr, int nr, CIT_TBL xc, CIT_TBL z, CIT_TBL s, int ix, int iy, double u0, double u1)
{
  const double wx = u0 - xr[ix];
  const double z00 = z[ix + nr * iy], z01 = z[ix + nr * (iy + 1)];
  double a = s[ix + (nr - 1) * iy]; a = a * wx; a = a + z00;
  double b = s[ix + (nr - 1) * (iy + 1)]; b = b * wx; b = b + z01;
  const double y0 = xc[iy];
  const double dy = xc[iy + 1] - y0;
  double r = b - a; r = r / dy; r = r * (u1 - y0);
  return r + a;
}

CIT_HD double cit_lookup1d(CIT_TBL x, int n, double u, CIT_TBL y)
{
  const int i = cit_lookup_index(x, n, u);
  const double x0 = x[i], x1 = x[i + 1], y0 = y[i], y1 = y[i + 1];
  double r = y1 - y0;
  r = r / (x1 - x0);
  r = r * (u - x0);
  return r + y0;
}

CIT_HD double cit_lookup2d(CIT_TBL xr, int nr, CIT_TBL xc, int nc, CIT_TBL z, double u0, double u1)
{
  const int ix = cit_lookup_index(xr, nr, u0);
  const int iy = cit_lookup_index(xc, nc, u1);
  const double x0 = xr[ix], x1 = xr[ix + 1];
  const double dx = x1 - x0, wx = u0 - x0;
  const double z00 = z[ix + nr * iy], z10 = z[ix + 1 + nr * iy];
  const double z01 = z[ix + nr * (iy + 1)], z11 = z[ix + 1 + nr * (iy + 1)];
  double a = z10 - z00; a = a / dx; a = a * wx; a = a + z00;
  double b = z11 - z01; b = b / dx; b = b * wx; b = b + z01;
  const double y0 = xc[iy];
  const double dy = xc[iy + 1] - y0;
  double r = b - a; r = r / dy; r = r * (u1 - y0);
  return r + a;
}

// c = A(3x3 row-major) * b, accumulated exactly like matmultiply @0x103a0: ((0 + a0*b0) + a1*b1) + a2*b2
#define CIT_MATVEC(c0, c1, c2, A, b0, b1, b2)                                   \
  do {                                                                          \
    c0 = ((0.0 + (A)[0] * (b0)) + (A)[1] * (b1)) + (A)[2] * (b2);                \
    c1 = ((0.0 + (A)[3] * (b0)) + (A)[4] * (b1)) + (A)[5] * (b2);                \
    c2 = ((0.0 + (A)[6] * (b0)) + (A)[7] * (b1)) + (A)[8] * (b2);                \
  } while (0)
// c = A^T * b with the same accumulation order over the row index of A^T
#define CIT_MATVEC_T(c0, c1, c2, A, b0, b1, b2)                                 \
  do {                                                                          \
    c0 = ((0.0 + (A)[0] * (b0)) + (A)[3] * (b1)) + (A)[6] * (b2);                \
    c1 = ((0.0 + (A)[1] * (b0)) + (A)[4] * (b1)) + (A)[7] * (b2);                \
    c2 = ((0.0 + (A)[2] * (b0)) + (A)[5] * (b1)) + (A)[8] * (b2);                \
  } while (0)

// trigonometry + the three rotation matrices, in the reference's operation order (@0x1042d..0x10797)
CIT_HD void cit_axes_prepare(CitAxes *ax, double alpha, double beta, double phi, double theta, double psi)
{
  double s4, c4, s5, c5, s6, c6, s7, c7, s8, c8;
  CIT_SINCOS(alpha, &s4, &c4);
  CIT_SINCOS(beta, &s5, &c5);
  CIT_SINCOS(phi, &s6, &c6);
  CIT_SINCOS(theta, &s7, &c7);
  CIT_SINCOS(psi, &s8, &c8);
  ax->key[0] = alpha; ax->key[1] = beta; ax->key[2] = phi; ax->key[3] = theta; ax->key[4] = psi;
  ax->M1[0] = c5; ax->M1[1] = -s5; ax->M1[2] = 0.0;
  ax->M1[3] = s5; ax->M1[4] = c5;  ax->M1[5] = 0.0;
  ax->M1[6] = 0.0; ax->M1[7] = 0.0; ax->M1[8] = 1.0;
  ax->M2[0] = c4; ax->M2[1] = 0.0; ax->M2[2] = -s4;
  ax->M2[3] = 0.0; ax->M2[4] = 1.0; ax->M2[5] = 0.0;
  ax->M2[6] = s4; ax->M2[7] = 0.0; ax->M2[8] = c4;
  const double s6s7 = s6 * s7, c6s7 = c6 * s7;
  ax->M3[0] = c7 * c8;
  ax->M3[1] = s6s7 * c8 - c6 * s8;
  ax->M3[2] = c6s7 * c8 + s6 * s8;
  ax->M3[3] = c7 * s8;
  ax->M3[4] = s6s7 * s8 + c6 * c8;
  ax->M3[5] = c6s7 * s8 - c8 * s6;
  ax->M3[6] = -s7;
  ax->M3[7] = s6 * c7;
  ax->M3[8] = c6 * c7;
}

// one ac_axes call: u[15] = {.., alpha beta phi theta psi @4..8, .., v @12..14}; y[12] = v in the four frames
// {wind, stability, body, earth}; `mode` = frame v is given in.  Returns 0 or CIT_ERR_AXES_KEY.
CIT_HD int cit_axes_apply(const CitAxes *ax, const double *u, double *y, int mode)
{
  int err = 0;
  for (int k = 0; k < 5; ++k) if (cit_bits(u[4 + k]) != cit_bits(ax->key[k])) err = CIT_ERR_AXES_KEY;
  const double v0 = u[12], v1 = u[13], v2 = u[14];
  double f0[3], f1[3], f2[3], f3[3];
  if (mode == 0) {
    f0[0] = v0; f0[1] = v1; f0[2] = v2;
    CIT_MATVEC(f1[0], f1[1], f1[2], ax->M1, f0[0], f0[1], f0[2]);
    CIT_MATVEC(f2[0], f2[1], f2[2], ax->M2, f1[0], f1[1], f1[2]);
    CIT_MATVEC(f3[0], f3[1], f3[2], ax->M3, f2[0], f2[1], f2[2]);
  } else if (mode == 1) {
    f1[0] = v0; f1[1] = v1; f1[2] = v2;
    CIT_MATVEC_T(f0[0], f0[1], f0[2], ax->M1, f1[0], f1[1], f1[2]);
    CIT_MATVEC(f2[0], f2[1], f2[2], ax->M2, f1[0], f1[1], f1[2]);
    CIT_MATVEC(f3[0], f3[1], f3[2], ax->M3, f2[0], f2[1], f2[2]);
  } else if (mode == 2) {
    f2[0] = v0; f2[1] = v1; f2[2] = v2;
    CIT_MATVEC_T(f1[0], f1[1], f1[2], ax->M2, f2[0], f2[1], f2[2]);
    CIT_MATVEC(f3[0], f3[1], f3[2], ax->M3, f2[0], f2[1], f2[2]);
    CIT_MATVEC_T(f0[0], f0[1], f0[2], ax->M1, f1[0], f1[1], f1[2]);
  } else {
    f3[0] = v0; f3[1] = v1; f3[2] = v2;
    CIT_MATVEC_T(f2[0], f2[1], f2[2], ax->M3, f3[0], f3[1], f3[2]);
    CIT_MATVEC_T(f1[0], f1[1], f1[2], ax->M2, f2[0], f2[1], f2[2]);
    CIT_MATVEC_T(f0[0], f0[1], f0[2], ax->M1, f1[0], f1[1], f1[2]);
  }
  for (int i = 0; i < 3; ++i) { y[i] = f0[i]; y[3 + i] = f1[i]; y[6 + i] = f2[i]; y[9 + i] = f3[i]; }
  return err;
}
