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

// ---- measured FP64 vector peak ("microbenchmark v_fma_f64 and use the measured value"): every CU, 16 wavefronts per CU (four per
// SIMD), sixteen independent fma chains per lane -- enough independent work that neither dependency latency nor issue gaps limit the rate.
// FLOP/s = 2 x 64 lanes x fma instructions / wall time (HIP events).
__global__ void __launch_bounds__(256) fma_peak(double *out, int iters, double a0, double b0)
{
  double r[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) r[j] = a0 + (threadIdx.x + j) * 1e-9;
  const double m = b0, c = 1e-12;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) r[j] = __builtin_fma(r[j], m, c);
  }
  double s = 0.0;
#pragma unroll
  for (int j = 0; j < 16; ++j) s += r[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static double fma_peak_tflops(int *cus_out)
{
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount, blocks = cus * 4, iters = 1 << 16;
  double *out; hipMalloc(&out, sizeof(double) * blocks * 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  double best = 0.0;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(fma_peak, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001, 0.9999999);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 256.0 * blocks * 16.0 * iters;
    if (rep > 0 && flops / (ms * 1e-3) > best) best = flops / (ms * 1e-3);
  }
  hipFree(out);
  *cus_out = cus;
  return best / 1e12;
}

int main()
{
  const char *names[] = {"dep_add_f64", "dep_mul_f64", "dep_fma_f64", "indep_add_f64_x4", "dep_div_f64", "dep_sqrt_plus_add",
                         "cmp_select_add", "lds_roundtrip_plus_add", "dep_add_then_mul"};
  double v[9] = {run<0>(1), run<1>(1), run<2>(1), run<3>(1), run<4>(1), run<5>(1), run<6>(1), run<7>(1), run<8>(1)};
  double v4[9] = {run<0>(4), run<1>(4), run<2>(4), run<3>(4), run<4>(4), run<5>(4), run<6>(4), run<7>(4), run<8>(4)};
  printf("{\"what\": \"shader cycles per loop iteration, one wavefront per SIMD (1 wave per block / 4 waves per block = one per SIMD)\"");
  for (int i = 0; i < 9; ++i) printf(", \"%s\": [%.2f, %.2f]", names[i], v[i], v4[i]);
  int cus = 0;
  const double peak = fma_peak_tflops(&cus);
  printf(", \"fp64_fma_peak_tflops_measured\": %.2f, \"fp64_peak_note\": \"v_fma_f64, %d CUs x 16 wavefronts, 16 independent chains per lane, HIP events; datasheet vector FP64: 78.6 TFLOP/s\"", peak, cus);
  printf("}\n");
  return 0;
}
