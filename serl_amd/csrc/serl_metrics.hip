// serl_metrics.hip -- episode metrics of the reference on the device, for batches of episodes of DIFFERENT lengths.
//
//   serl_smoothness   calc_smoothness (base/core/utils.py:82-120) of E action traces in one launch.  The reference takes
//                     Y = fft(y, N) with N = the episode's own length; a training generation (untrained actors crash after
//                     2 .. 20 s) has ~150 distinct lengths, and an FFT library plans -- rocFFT even compiles -- per length
//                     (measured: 1.7 s of host time per generation against 47 ms of rollout kernel).  The metric only needs
//                     sum_i |Y_i|^2 f_i over 1 <= i < N/2, so the DFT is evaluated directly: one thread per frequency, the
//                     N twiddles of the episode's length in LDS (index i*k mod N advanced incrementally), the samples read
//                     as wave-uniform loads; O(N^2) per episode but dense f64 FMA work the GPU has to spare (N = 2 001:
//                     12 M FMA per episode) and no plan, no host synchronisation, no dependence on the set of lengths.
#include <hip/hip_runtime.h>
#include <math.h>
#include "serl_ctx.h"

namespace {

constexpr int SM_THREADS = 256;
constexpr int SM_MAX_N = 8192;          // twiddle table in LDS: 16 B per entry (128 KB)

// partial[e][chunk] = sum over the chunk's frequencies i of  f_i * dt * sum_c |Y_i,c|^2
__global__ void __launch_bounds__(SM_THREADS) smoothness_partial_kernel(const double *__restrict__ actions, int64_t episode_stride,
                                                                        const int32_t *__restrict__ lengths, int max_chunks, double dt,
                                                                        double *__restrict__ partial)
{
  extern __shared__ double2 tw[];                        // (cos, sin)(2 pi m / N)
  __shared__ double red[SM_THREADS / 64];
  const int e = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
  int N = lengths[e];
  N = N < 0 ? -N : N;
  const int nf = N / 2 - 1;                              // frequencies 1 .. N/2 - 1   (Y[1:N//2])
  if (N < 4 || chunk * SM_THREADS >= nf) {
    if (tid == 0) partial[(size_t)e * max_chunks + chunk] = 0.0;
    return;
  }
  for (int m = tid; m < N; m += SM_THREADS) {
    double s, c;
    sincospi(2.0 * (double)m / (double)N, &s, &c);
    tw[m] = make_double2(c, s);
  }
  __syncthreads();
  const int i = 1 + chunk * SM_THREADS + tid;
  const bool active = i <= nf;
  const int step = active ? i : 1;
  const double *y = actions + (size_t)e * episode_stride;
  double re0 = 0, re1 = 0, re2 = 0, im0 = 0, im1 = 0, im2 = 0;
  int m = 0;
  for (int k = 0; k < N; ++k) {
    const double y0 = y[(size_t)k * 3], y1 = y[(size_t)k * 3 + 1], y2 = y[(size_t)k * 3 + 2];     // the same address in every lane
    const double2 w = tw[m];
    re0 = fma(y0, w.x, re0); im0 = fma(-y0, w.y, im0);
    re1 = fma(y1, w.x, re1); im1 = fma(-y1, w.y, im1);
    re2 = fma(y2, w.x, re2); im2 = fma(-y2, w.y, im2);
    m += step;
    if (m >= N) m -= N;
  }
  // numpy.linspace(dt, 1 / (2 dt), N//2 - 1)[i - 1]
  const double f = nf > 1 ? dt + (double)(i - 1) * ((1.0 / (2.0 * dt) - dt) / (double)(nf - 1)) : dt;
  double v = active ? ((re0 * re0 + im0 * im0) + (re1 * re1 + im1 * im1) + (re2 * re2 + im2 * im2)) * dt * f : 0.0;
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);                 // fixed order: the result does not depend on scheduling
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  if (tid == 0) partial[(size_t)e * max_chunks + chunk] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out[e] = -sqrt(S * 2 / N) * 100 * (80 / (N dt)),  S = sum of the episode's partials in chunk order
__global__ void smoothness_final_kernel(const double *__restrict__ partial, int max_chunks, const int32_t *__restrict__ lengths,
                                        int n_episodes, double dt, double *__restrict__ out)
{
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_episodes) return;
  int N = lengths[e];
  N = N < 0 ? -N : N;
  if (N < 4) { out[e] = 0.0; return; }
  double S = 0.0;
  const int chunks = (N / 2 - 1 + SM_THREADS - 1) / SM_THREADS;
  for (int c = 0; c < chunks; ++c) S += partial[(size_t)e * max_chunks + c];
  out[e] = -(sqrt(S * 2.0 / (double)N) * 100.0 * (80.0 / ((double)N * dt)));
}

}  // namespace

extern "C" {

int serl_smoothness_work_size(int32_t n_episodes, int32_t max_len)
{
  const int chunks = (max_len / 2 - 1 + SM_THREADS - 1) / SM_THREADS;
  return n_episodes * (chunks < 1 ? 1 : chunks);
}

int serl_smoothness(serl_ctx *c, const double *actions, int64_t episode_stride, const int32_t *lengths, int32_t n_episodes,
                    int32_t max_len, double dt, double *work, double *out, void *stream_)
{
  if (!c || !actions || !lengths || !work || !out || n_episodes <= 0 || max_len <= 0 || !(dt > 0.0))
    return serl_fail(SERL_E_INVALID, "serl_smoothness: bad argument");
  if (max_len > SM_MAX_N) return serl_fail(SERL_E_UNSUPPORTED, "serl_smoothness: episodes longer than 8192 steps (twiddle table in LDS)");
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t stream = (hipStream_t)stream_;
  int chunks = (max_len / 2 - 1 + SM_THREADS - 1) / SM_THREADS;
  if (chunks < 1) chunks = 1;
  const size_t lds = (size_t)max_len * sizeof(double2);
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(smoothness_partial_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(SM_MAX_N * sizeof(double2))));
  hipLaunchKernelGGL(smoothness_partial_kernel, dim3(chunks, n_episodes), dim3(SM_THREADS), lds, stream, actions, episode_stride, lengths,
                     chunks, dt, work);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(smoothness_final_kernel, dim3((n_episodes + 255) / 256), dim3(256), 0, stream, work, chunks, lengths, n_episodes, dt, out);
  HIP_TRY(hipGetLastError());
  return SERL_OK;
}

}  // extern "C"
