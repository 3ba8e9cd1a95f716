// Host check of citation_leaves.h cit_lookup2d_at_s / cit_lookup1d_at_s (the lane-per-episode kernels' interpolation over precomputed x-direction quotients)
// against cit_lookup2d_at / cit_lookup1d_at (the reference's operation order, rt_Lookup2D_Normal): bit for bit on random tables, with the quotients computed the way
// rollout_variant.inc stages them.  Prints the number of mismatches.  Built by tests/test_dag_model.py with -ffp-contract=off.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CIT_HD static inline
#define CIT_NOINLINE static
#include <math.h>
#define CIT_SINCOS(x, s, c) sincos((x), (s), (c))
#include "citation_leaves.h"

static double rnd(uint64_t &s) { s = s * 6364136223846793005ULL + 1442695040888963407ULL; return (double)(s >> 11) / 9007199254740992.0; }

int main()
{
  uint64_t seed = 12345;
  long bad = 0, n = 0;
  for (int rep = 0; rep < 200; ++rep) {
    const int nr = 2 + (int)(rnd(seed) * 21), nc = 2 + (int)(rnd(seed) * 21);
    std::vector<double> xr(nr), xc(nc), z((size_t)nr * nc), s((size_t)(nr - 1) * nc), s1(nr - 1);
    double a = -3.0 + rnd(seed);
    for (int i = 0; i < nr; ++i) { a += 0.01 + rnd(seed); xr[i] = a; }
    a = -1.0 + rnd(seed);
    for (int i = 0; i < nc; ++i) { a += 0.001 + 0.3 * rnd(seed); xc[i] = a; }
    for (auto &v : z) v = (rnd(seed) - 0.5) * 7.0;
    for (int iy = 0; iy < nc; ++iy)
      for (int ix = 0; ix < nr - 1; ++ix) { double q = z[ix + 1 + nr * iy] - z[ix + nr * iy]; q = q / (xr[ix + 1] - xr[ix]); s[ix + (nr - 1) * iy] = q; }
    for (int ix = 0; ix < nr - 1; ++ix) { double q = z[ix + 1] - z[ix]; q = q / (xr[ix + 1] - xr[ix]); s1[ix] = q; }
    for (int k = 0; k < 500; ++k) {
      const double u0 = xr[0] - 0.5 + rnd(seed) * (xr[nr - 1] - xr[0] + 1.0), u1 = xc[0] - 0.1 + rnd(seed) * (xc[nc - 1] - xc[0] + 0.2);
      const int ix = cit_lookup_index(xr.data(), nr, u0), iy = cit_lookup_index(xc.data(), nc, u1);
      const double p = cit_lookup2d_at(xr.data(), nr, xc.data(), z.data(), ix, iy, u0, u1), q = cit_lookup2d_at_s(xr.data(), nr, xc.data(), z.data(), s.data(), ix, iy, u0, u1);
      const double p1 = cit_lookup1d_at(xr.data(), ix, u0, z.data()), q1 = cit_lookup1d_at_s(xr.data(), ix, u0, z.data(), s1.data());
      bad += memcmp(&p, &q, 8) != 0; bad += memcmp(&p1, &q1, 8) != 0; n += 2;
    }
  }
  printf("%ld %ld\n", bad, n);
  return bad != 0;
}
