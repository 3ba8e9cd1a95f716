// valu_latency.hip -- micro-benchmark behind the issue / latency floor quoted in DESIGN.md and the bench line:
// what ONE wavefront alone on a SIMD (the team kernels' situation) pays per instruction.
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/valu_latency.hip -o /tmp/valu_latency && /tmp/valu_latency
// Prints one JSON line: shader cycles per operation for dependent and independent chains of f64 add / mul / fma, the
// IEEE f64 division and sqrt the compiler emits, a v_cndmask select pair, and an LDS write -> read round trip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define N 4096

template <int MODE>
__global__ void chain(double *out, unsigned long long *cyc, double a0, double b0)
{
  __shared__ double lds[64];
  double a = a0 + threadIdx.x * 1e-9, b = b0, c = a0 * 0.5, d = b0 * 0.25, e = a0 * 0.125;
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 16
  for (int i = 0; i < N; ++i) {
    if (MODE == 0) a = a + b;                               // dependent add
    if (MODE == 1) a = a * b;                               // dependent mul
    if (MODE == 2) a = __builtin_fma(a, b, b);              // dependent fma
    if (MODE == 3) { a = a + b; c = c + b; d = d + b; e = e + b; }       // four independent adds
    if (MODE == 4) a = b / a;                               // dependent division
    if (MODE == 5) a = __builtin_sqrt(a + b);               // dependent sqrt (+ add)
    if (MODE == 6) a = (a > b) ? c : (a + d);               // compare + select + add
    if (MODE == 7) { lds[threadIdx.x & 63] = a; a = lds[(threadIdx.x + 1) & 63] + b; }   // LDS round trip (+ add)
    if (MODE == 8) a = (a + b) * c;                         // add feeding mul
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + c + d + e;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
double run(int waves_per_block)
{
  double *out; unsigned long long *cyc;
  hipMalloc(&out, sizeof(double) * 64 * 8);
  hipMalloc(&cyc, sizeof(unsigned long long));
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(chain<MODE>, dim3(1), dim3(64 * waves_per_block), 0, 0, out, cyc, 1.0000001, 0.9999999);
    hipDeviceSynchronize();
  }
  unsigned long long h = 0;
  hipMemcpy(&h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  hipFree(out); hipFree(cyc);
  return (double)h / N;
}

int main()
{
  const char *names[] = {"dep_add_f64", "dep_mul_f64", "dep_fma_f64", "indep_add_f64_x4", "dep_div_f64", "dep_sqrt_plus_add",
                         "cmp_select_add", "lds_roundtrip_plus_add", "dep_add_then_mul"};
  double v[9] = {run<0>(1), run<1>(1), run<2>(1), run<3>(1), run<4>(1), run<5>(1), run<6>(1), run<7>(1), run<8>(1)};
  double v4[9] = {run<0>(4), run<1>(4), run<2>(4), run<3>(4), run<4>(4), run<5>(4), run<6>(4), run<7>(4), run<8>(4)};
  printf("{\"what\": \"shader cycles per loop iteration, one wavefront per SIMD (1 wave per block / 4 waves per block = one per SIMD)\"");
  for (int i = 0; i < 9; ++i) printf(", \"%s\": [%.2f, %.2f]", names[i], v[i], v4[i]);
  printf("}\n");
  return 0;
}
