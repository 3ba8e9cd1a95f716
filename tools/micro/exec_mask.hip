// exec_mask.hip -- does a VALU instruction cost less issue time when only a few lanes of the wavefront are active?  The team kernels run wave-uniform
// (scalar-like) f64 code on 64 lanes: if the SIMD skipped the 16-lane passes whose lanes are all masked off, that code could run under a narrow EXEC mask.
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/micro/exec_mask.hip -o /tmp/exec_mask && /tmp/exec_mask
// Prints shader cycles per instruction of four independent f64 add chains and of a dependent chain, one wavefront on a SIMD, with 64 / 32 / 16 / 1
// active lanes, and the same with TWO wavefronts sharing the SIMD (the team kernels' situation).
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 4096
template <int MODE>
__global__ void chain(double *out, unsigned long long *cyc, double a0, double b0, int active)
{
  double a = a0 + threadIdx.x * 1e-9, b = b0, c = a0 * 0.5, d = b0 * 0.25, e = a0 * 0.125;
  unsigned long long t0 = 0, t1 = 0;
  if ((int)(threadIdx.x & 63) < active) {
    t0 = __builtin_readcyclecounter();
#pragma unroll 16
    for (int i = 0; i < N; ++i) {
      if (MODE == 0) a = a + b;
      if (MODE == 1) { a = a + b; c = c + b; d = d + b; e = e + b; }
      if (MODE == 2) { a = __builtin_fma(a, b, b); c = __builtin_fma(c, b, b); d = __builtin_fma(d, b, b); e = __builtin_fma(e, b, b); }
      if (MODE == 3) { float fa = (float)a, fb = (float)b; for (int k = 0; k < 4; ++k) fa = fa * fb + fb; a = fa; }
    }
    t1 = __builtin_readcyclecounter();
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + c + d + e;
  if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}
template <int MODE>
static double run(int waves, int active)
{
  double *out; unsigned long long *cyc;
  hipMalloc(&out, sizeof(double) * 64 * 16); hipMalloc(&cyc, sizeof(unsigned long long) * 16);
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(chain<MODE>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, 1.0000001, 0.9999999, active); hipDeviceSynchronize(); }
  unsigned long long h[16]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  hipFree(out); hipFree(cyc);
  return (double)h[0] / N;
}
int main()
{
  const int act[4] = {64, 32, 16, 1};
  printf("{");
  for (int w : {1, 4, 8}) {      // 1: one wavefront; 4: one per SIMD; 8: two per SIMD
    for (int k = 0; k < 4; ++k) {
      printf("\"w%d_a%d\": {\"dep_add_f64\": %.2f, \"indep_add_f64_x4\": %.2f, \"indep_fma_f64_x4\": %.2f}%s", w, act[k],
             run<0>(w, act[k]), run<1>(w, act[k]), run<2>(w, act[k]), (w == 8 && k == 3) ? "" : ", ");
    }
  }
  printf("}\n");
  return 0;
}
