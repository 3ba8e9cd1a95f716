// hwid_probe.hip -- where the dispatcher puts the workgroups of a one-workgroup-per-CU launch (512 threads, 150 KB of LDS: the shape of the
// four-episodes-per-team kernels).  Prints one JSON line: for grids of 192 and 256 workgroups the (XCC_ID, SE_ID, SH_ID, CU_ID) of every
// workgroup in blockIdx order (HW_REG_HW_ID: CU_ID[11:8], SH_ID[12], SE_ID[15:13]; HW_REG_XCC_ID[3:0]).  Round 5 used it to place the code
// variants of a mixed-fault sweep on CUs that share an instruction cache (profiles/r05_experiments.md section 11).
//   hipcc --offload-arch=gfx950 -O2 tools/hwid_probe.hip -o /tmp/hwid_probe && /tmp/hwid_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(512) probe(unsigned *out, int spin)
{
  extern __shared__ double lds[];
  if (threadIdx.x == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);
    const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);
    out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc;
    lds[0] = (double)hw;
  }
  const unsigned long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(8);     // keep every workgroup resident
  __syncthreads();
  if (lds[0] < 0.0) out[0] = 0;
}
int main()
{
  unsigned *d;
  hipMalloc(&d, 2 * 1024 * sizeof(unsigned));
  hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  printf("{");
  const int grids[3] = {192, 256, 512};
  for (int gi = 0; gi < 3; ++gi) {
    const int g = grids[gi];
    hipMemset(d, 0xff, 2 * 1024 * sizeof(unsigned));
    hipLaunchKernelGGL(probe, dim3(g), dim3(512), 150 * 1024, 0, d, 400000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(2 * g);
    hipMemcpy(h.data(), d, 2 * g * sizeof(unsigned), hipMemcpyDeviceToHost);
    printf("%s\"grid_%d\": [", gi ? ", " : "", g);
    for (int b = 0; b < g; ++b)
      printf("%s[%u,%u,%u,%u]", b ? "," : "", h[2 * b + 1] & 15u, (h[2 * b] >> 13) & 7u, (h[2 * b] >> 12) & 1u, (h[2 * b] >> 8) & 15u);
    printf("]");
  }
  printf("}\n");
  return 0;
}
