/* CPU build of serl_amd/csrc/citation_libm.h for tests/test_libm.py (the same text the kernels compile): arrays in, arrays out. */
#include "../../serl_amd/csrc/citation_libm.h"
void host_sincos(const double *x, double *s, double *c, long n) { for (long i = 0; i < n; ++i) citw_sincos(x[i], &s[i], &c[i]); }
void host_tan(const double *x, double *t, long n) { for (long i = 0; i < n; ++i) t[i] = citw_tan(x[i]); }
void host_pow(const double *x, double c, double *y, long n) { for (long i = 0; i < n; ++i) y[i] = citw_pow(x[i], c); }
void host_atan(const double *x, double *y, long n) { for (long i = 0; i < n; ++i) y[i] = citw_atan(x[i]); }
