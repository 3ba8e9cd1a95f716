// actor_bench.hip -- cycles of one actor forward pass on a lone wavefront, by shape and activation (development aid):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iserl_amd/csrc -Iinclude -DCITW_MAX_WAVES=1 -DCITW_M_ROWS=8 \
//         -DCITW_OUT2_ROWS=1 -DCITW_INV_SLOTS=8 tools/actor_bench.hip -o /tmp/actor_bench && /tmp/actor_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "citation_wave.h"
#include "rollout_device.h"

__global__ void __launch_bounds__(64) bench(serl_rollout_desc d, const float *w, int reps, unsigned long long *cyc, float *out)
{
  float o[7] = {0.01f, -0.02f, 0.005f, 0.1f, -0.05f, 0.02f, 0.03f}, a[3] = {0, 0, 0};
  SerlNoSync ns;
  serl_actor_forward(d, w, o, a, ns);                 // warm the caches
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; ++r) {
    o[0] = a[0] * 0.01f;                              // (a dependency from pass to pass, like the episode loop's)
    serl_actor_forward(d, w, o, a, ns);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { cyc[0] = (t1 - t0) / reps; out[0] = a[0]; }
}

// the same with the generic streaming forward (run-time shape, 32-column chunks): what shapes without a specialised path use
__global__ void __launch_bounds__(64) bench_generic(serl_rollout_desc d, const float *w, int reps, unsigned long long *cyc, float *out)
{
  float o[7] = {0.01f, -0.02f, 0.005f, 0.1f, -0.05f, 0.02f, 0.03f}, a[3] = {0, 0, 0};
  SerlNoSync ns;
  serl_actor_forward_wave(d, w, o, a, ns);
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; ++r) {
    o[0] = a[0] * 0.01f;
    serl_actor_forward_wave(d, w, o, a, ns);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { cyc[0] = (t1 - t0) / reps; out[0] = a[0]; }
}

int main()
{
  const int shapes[][2] = {{32, 0}, {32, 2}, {64, 0}, {72, 0}, {72, 2}, {96, 0}, {96, 2}, {128, 0}};
  for (auto &sh : shapes) {
    const int H = sh[0], act = sh[1], S = 7, L = 3, A = 3;
    const int P = H * S + H + L * (H * H + 3 * H) + A * H + A, stride = (P + 3) / 4 * 4;
    std::vector<float> w(stride);
    unsigned s = 12345;
    for (int i = 0; i < P; ++i) { s = s * 1664525u + 1013904223u; w[i] = ((s >> 8) / 16777216.0f - 0.5f) * 0.3f; }
    for (int l = 0; l < L; ++l) for (int i = 0; i < H; ++i) w[H * S + H + l * (H * H + 3 * H) + H * H + H + i] = 1.0f;   // gamma
    float *dw, *dout; unsigned long long *dc;
    hipMalloc(&dw, stride * 4); hipMalloc(&dout, 16); hipMalloc(&dc, 8);
    hipMemcpy(dw, w.data(), stride * 4, hipMemcpyHostToDevice);
    serl_rollout_desc d = {};
    d.state_dim = S; d.action_dim = A; d.hidden = H; d.num_layers = L; d.activation = act; d.weights = dw; d.weight_stride = stride;
    hipLaunchKernelGGL(bench, dim3(1), dim3(64), 0, 0, d, dw, 200, dc, dout);
    unsigned long long c = 0; float o = 0;
    hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost); hipMemcpy(&o, dout, 4, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(bench_generic, dim3(1), dim3(64), 0, 0, d, dw, 200, dc, dout);
    unsigned long long cg = 0; float og = 0;
    hipMemcpy(&cg, dc, 8, hipMemcpyDeviceToHost); hipMemcpy(&og, dout, 4, hipMemcpyDeviceToHost);
    printf("{\"hidden\": %d, \"activation\": %d, \"cycles_per_forward\": %llu, \"out0\": %g, \"cycles_per_forward_generic\": %llu, \"same_bits_as_generic\": %d}\n", H, act, c, o, cg, o == og);
    hipFree(dw); hipFree(dout); hipFree(dc);
  }
  return 0;
}
