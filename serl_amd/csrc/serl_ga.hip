// serl_ga.hip -- the SSNE operators that need the actor itself (base/core/mod_neuro_evo.py), and the replay-ring append:
//
//   serl_ga_sensitivity   the per-weight output sensitivity of proximal_mutate / safe_mutate (:183-223, :254-298): the
//                         reference runs one torch backward pass per action output over a batch of stored states,
//                         collects the three gradient vectors of the 2-D weights ("jacobian"), and scales the Gaussian
//                         perturbation by sqrt(sum_i jacobian_i^2) with its clamp rules.  Here: forward + three analytic
//                         backward passes of the 7-H-(H+LayerNorm)xL-3 MLP per batch element, one workgroup per member,
//                         the gradient accumulators of one output resident in LDS.
//   serl_ga_novelty       Actor.get_novelty (base/core/genetic_agent.py:111-115): mean over a batch of
//                         sum_a (action - actor(state))^2 -- the distance of SSNE.get_distance / sort_groups_by_distance
//                         (:411-445), all (actor, batch) pairs of an epoch in one launch.
//   serl_replay_scatter   append whole stored episodes (rows the rollout kernel wrote) to device replay rings; cost-flagged
//                         rows compacted for the critical rings (base/core/agent.py:101-112).
//
// f32 throughout like the reference's torch CPU kernels (the summation order differs: agreement is to f32 rounding).
#include <hip/hip_runtime.h>
#include <math.h>
#include "serl_ctx.h"

namespace {

struct NetDims { int S, H, L, A, act; };

__device__ __forceinline__ float ga_act(float v, int act)
{
  if (act == SERL_ACT_TANH) return tanhf(v);
  if (act == SERL_ACT_ELU) return v > 0.0f ? v : expm1f(v);
  return v > 0.0f ? v : 0.01f * v;
}

// d act / d pre-activation, from the pre-activation y and the activation value a
__device__ __forceinline__ float ga_dact(float y, float a, int act)
{
  if (act == SERL_ACT_TANH) return 1.0f - a * a;
  if (act == SERL_ACT_ELU) return y > 0.0f ? 1.0f : a + 1.0f;
  return y > 0.0f ? 1.0f : 0.01f;
}

// workgroup-wide sum of one float per thread (blockDim.x threads, red[] >= blockDim.x / 64 floats + 1)
__device__ float ga_block_sum(float v, float *red)
{
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float s = 0.0f;
  for (int i = 0; i < nw; ++i) s += red[i];
  return s;
}

// LDS layout of one forward pass (floats): x[S] | a[(L+1)][H] post-activation | y[(L+1)][H] pre-activation (layer 0: z0,
// hidden layers: LayerNorm output) | c[L][H] centred z | sig[L] | out[A] | pre_out[A]
struct FwdBuf {
  float *x, *a, *y, *c, *sig, *out, *red;
  __device__ static int floats(const NetDims &n) { return n.S + 2 * (n.L + 1) * n.H + n.L * n.H + n.L + 2 * n.A + 16; }
  __device__ void bind(float *p, const NetDims &n)
  {
    x = p; p += n.S; a = p; p += (n.L + 1) * n.H; y = p; p += (n.L + 1) * n.H; c = p; p += n.L * n.H;
    sig = p; p += n.L; out = p; p += n.A; red = p;
  }
};

// forward of one state (Actor.forward, genetic_agent.py:104; LayerNorm mod_utils.py:47-50), all threads of the block
__device__ void ga_forward(const NetDims &n, const float *w, const float *state, FwdBuf &f)
{
  const int H = n.H, t = threadIdx.x;
  if (t < n.S) f.x[t] = state[t];
  __syncthreads();
  const float *W = w, *b = w + (size_t)H * n.S;
  for (int r = t; r < H; r += blockDim.x) {
    float acc = b[r];
    for (int j = 0; j < n.S; ++j) acc += W[(size_t)r * n.S + j] * f.x[j];
    f.y[r] = acc;
    f.a[r] = ga_act(acc, n.act);
  }
  __syncthreads();
  const float *p = b + H;
  for (int l = 0; l < n.L; ++l) {
    const float *Wl = p, *bl = p + (size_t)H * H, *g = bl + H, *be = g + H;
    const float *prev = f.a + (size_t)l * H;
    float z = 0.0f;
    if (t < H) {
      z = bl[t];
      for (int j = 0; j < H; ++j) z += Wl[(size_t)t * H + j] * prev[j];
    }
    const float mean = ga_block_sum(t < H ? z : 0.0f, f.red) / (float)H;
    const float cz = t < H ? z - mean : 0.0f;
    const float var = ga_block_sum(cz * cz, f.red) / (float)(H - 1);
    const float sd = sqrtf(var);
    if (t < H) {
      const float yv = g[t] * cz / (sd + 1e-6f) + be[t];
      f.c[(size_t)l * H + t] = cz;
      f.y[(size_t)(l + 1) * H + t] = yv;
      f.a[(size_t)(l + 1) * H + t] = ga_act(yv, n.act);
    }
    if (t == 0) f.sig[l] = sd;
    __syncthreads();
    p = be + H;
  }
  const float *Wo = p, *bo = p + (size_t)n.A * H;
  const float *last = f.a + (size_t)n.L * H;
  if (t < n.A) {
    float acc = bo[t];
    for (int j = 0; j < H; ++j) acc += Wo[(size_t)t * H + j] * last[j];
    f.out[t] = tanhf(acc);
  }
  __syncthreads();
}

// One workgroup per listed member.  Dynamic LDS: J[G] (gradient of sum_b out[b][i] w.r.t. the genome, one output at a time)
// | forward buffers | g[H] dz[H].  scaling[member][G] accumulates sum_i J_i^2 and ends as the clamped square root.
__global__ void __launch_bounds__(256) ga_sensitivity_kernel(const float *weights, int64_t stride, NetDims n, const int32_t *members,
                                                             const float *states, int B, float *scaling, int G)
{
  extern __shared__ float lds[];
  const int H = n.H, t = threadIdx.x, m = blockIdx.x;
  const float *w = weights + (size_t)members[m] * stride;
  float *J = lds;
  FwdBuf f; f.bind(lds + G, n);
  float *gvec = lds + G + FwdBuf::floats(n), *dz = gvec + H;
  float *sc = scaling + (size_t)m * G;
  for (int e = t; e < G; e += blockDim.x) sc[e] = 0.0f;
  // genome order = named_parameters order of the 2-D weights: W0[H][S], W1..WL[H][H], Wo[A][H]
  const int offW0 = 0, offWl = H * n.S, offWo = H * n.S + n.L * H * H;
  const size_t pWl0 = (size_t)H * n.S + H;                  // packed-row offset of W1
  const size_t pStep = (size_t)H * H + 3 * H;
  const float *Wo = w + pWl0 + (size_t)n.L * pStep;
  for (int i = 0; i < n.A; ++i) {
    for (int e = t; e < G; e += blockDim.x) J[e] = 0.0f;
    __syncthreads();
    for (int b = 0; b < B; ++b) {
      ga_forward(n, w, states + ((size_t)m * B + b) * n.S, f);
      // output layer: d out_i / d pre_i = 1 - out_i^2 (nn.Tanh)
      const float dout = 1.0f - f.out[i] * f.out[i];
      const float *last = f.a + (size_t)n.L * H;
      if (t < H) {
        J[offWo + i * H + t] += dout * last[t];
        gvec[t] = Wo[(size_t)i * H + t] * dout;               // d / d a_L
      }
      __syncthreads();
      for (int l = n.L - 1; l >= 0; --l) {
        const float *Wl = w + pWl0 + (size_t)l * pStep;
        const float *gam = Wl + (size_t)H * H + H;
        const float *cz = f.c + (size_t)l * H;
        const float sd = f.sig[l], D = sd + 1e-6f;
        // through the activation and the LayerNorm (unbiased std, eps added to the std)
        float gh = 0.0f, czt = 0.0f;
        if (t < H) {
          const float dy = gvec[t] * ga_dact(f.y[(size_t)(l + 1) * H + t], f.a[(size_t)(l + 1) * H + t], n.act);
          gh = dy * gam[t];
          czt = cz[t];
        }
        const float gmean = ga_block_sum(gh, f.red) / (float)H;
        const float gdotc = ga_block_sum(gh * czt, f.red);
        // sd == 0 (all pre-LayerNorm values equal): torch's std_backward masks the term to 0 there, the reference stays finite
        if (t < H) dz[t] = (gh - gmean) / D - (sd > 0.0f ? gdotc / (D * D) * czt / ((float)(H - 1) * sd) : 0.0f);

        __syncthreads();
        const float *prev = f.a + (size_t)l * H;
        float *Jl = J + offWl + (size_t)l * H * H;
        for (int e = t; e < H * H; e += blockDim.x) Jl[e] += dz[e / H] * prev[e % H];
        float gn = 0.0f;
        if (t < H) for (int r = 0; r < H; ++r) gn += Wl[(size_t)r * H + t] * dz[r];
        __syncthreads();
        if (t < H) gvec[t] = gn;
        __syncthreads();
      }
      if (t < H) dz[t] = gvec[t] * ga_dact(f.y[t], f.a[t], n.act);
      __syncthreads();
      for (int e = t; e < H * n.S; e += blockDim.x) J[offW0 + e] += dz[e / n.S] * f.x[e % n.S];
      __syncthreads();
    }
    for (int e = t; e < G; e += blockDim.x) sc[e] += J[e] * J[e];
    __syncthreads();
  }
  // mod_neuro_evo.py:213-217: scaling = sqrt(sum); scaling[scaling == 0] = 1; scaling[scaling < 0.01] = 0.01
  for (int e = t; e < G; e += blockDim.x) {
    float s = sqrtf(sc[e]);
    if (s == 0.0f) s = 1.0f;
    if (s < 0.01f) s = 0.01f;
    sc[e] = s;
  }
}

// grid = pairs: novelty[p] = mean_b sum_a (actions[p][b][a] - actor_{members[p]}(states[p][b]))^2
__global__ void __launch_bounds__(256) ga_novelty_kernel(const float *weights, int64_t stride, NetDims n, const int32_t *members,
                                                         const float *states, const float *actions, int B, float *novelty)
{
  extern __shared__ float lds[];
  FwdBuf f; f.bind(lds, n);
  const int p = blockIdx.x;
  const float *w = weights + (size_t)members[p] * stride;
  float acc = 0.0f;
  for (int b = 0; b < B; ++b) {
    ga_forward(n, w, states + ((size_t)p * B + b) * n.S, f);
    if (threadIdx.x == 0) {
      float s = 0.0f;
      for (int a = 0; a < n.A; ++a) { const float d = actions[((size_t)p * B + b) * n.A + a] - f.out[a]; s += d * d; }
      acc += s;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) novelty[p] = acc / (float)B;
}

struct ReplayJob { float *ring; int32_t capacity, position, episode, length, cost_only, skip; };

// One workgroup per job: rows of staged[episode][0 .. length) -> ring[(position + k) % capacity], k = the row's rank among
// the rows the job takes (all of them, or the cost-flagged ones: row[19] != 0); the first `skip` ranks are dropped
// (an episode longer than the ring: only what n sequential add() calls would have left).
__global__ void __launch_bounds__(256) replay_scatter_kernel(const float *staged, int64_t T, const ReplayJob *jobs)
{
  __shared__ int wsum[4], base;
  const ReplayJob j = jobs[blockIdx.x];
  const float *src = staged + (size_t)j.episode * T * 20;
  const int t = threadIdx.x;
  if (t == 0) base = 0;
  __syncthreads();
  for (int r0 = 0; r0 < j.length; r0 += blockDim.x) {
    const int r = r0 + t;
    const bool take = r < j.length && (!j.cost_only || src[(size_t)r * 20 + 19] != 0.0f);
    // exclusive prefix count of `take` over the block
    const unsigned long long bal = __ballot(take);
    const int lane = t & 63, w = t >> 6;
    const int in_wave = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[w] = __popcll(bal);
    __syncthreads();
    int before = base;
    for (int i = 0; i < w; ++i) before += wsum[i];
    const int rank = before + in_wave;
    if (take && rank >= j.skip) {
      float *dst = j.ring + (size_t)((j.position + rank) % j.capacity) * 20;
      const float4 *s4 = reinterpret_cast<const float4 *>(src + (size_t)r * 20);
      float4 *d4 = reinterpret_cast<float4 *>(dst);
      d4[0] = s4[0]; d4[1] = s4[1]; d4[2] = s4[2]; d4[3] = s4[3]; d4[4] = s4[4];
    }
    __syncthreads();
    if (t == 0) base += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
}

int check_net(const serl_ctx *c, int S, int H, int L, int A, int act, size_t lds_bytes, const char *who)
{
  if (S < 1 || S > 64 || A < 1 || A > 16 || H < 2 || H > 256 || L < 0 || L > 16 || act < 0 || act > 2)
    return serl_fail(SERL_E_UNSUPPORTED, std::string(who) + ": network shape out of range");
  if (lds_bytes > (size_t)c->lds_per_block)
    return serl_fail(SERL_E_UNSUPPORTED, std::string(who) + ": network too large for the LDS-resident accumulators");
  return SERL_OK;
}

}  // namespace

extern "C" {

int serl_ga_sensitivity(serl_ctx *c, const float *weights, int64_t stride, int32_t state_dim, int32_t hidden, int32_t num_layers,
                        int32_t action_dim, int32_t activation, const int32_t *members, int32_t n_members, const float *states,
                        int32_t batch, float *scaling, void *stream)
{
  if (!c || !weights || !members || !states || !scaling || n_members < 0 || batch <= 0)
    return serl_fail(SERL_E_INVALID, "serl_ga_sensitivity: bad argument");
  if (n_members == 0) return SERL_OK;
  const NetDims n{state_dim, hidden, num_layers, action_dim, activation};
  const int G = hidden * state_dim + num_layers * hidden * hidden + action_dim * hidden;
  const int fwd = n.S + 2 * (n.L + 1) * n.H + n.L * n.H + n.L + 2 * n.A + 16;
  const size_t lds = sizeof(float) * ((size_t)G + fwd + 2 * (size_t)hidden);
  if (int rc = check_net(c, state_dim, hidden, num_layers, action_dim, activation, lds, "serl_ga_sensitivity")) return rc;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(ga_sensitivity_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(ga_sensitivity_kernel, dim3(n_members), dim3(256), lds, (hipStream_t)stream, weights, stride, n, members, states,
                     batch, scaling, G);
  HIP_TRY(hipGetLastError());
  return SERL_OK;
}

int serl_ga_novelty(serl_ctx *c, const float *weights, int64_t stride, int32_t state_dim, int32_t hidden, int32_t num_layers,
                    int32_t action_dim, int32_t activation, const int32_t *members, int32_t n_pairs, const float *states,
                    const float *actions, int32_t batch, float *novelty, void *stream)
{
  if (!c || !weights || !members || !states || !actions || !novelty || n_pairs < 0 || batch <= 0)
    return serl_fail(SERL_E_INVALID, "serl_ga_novelty: bad argument");
  if (n_pairs == 0) return SERL_OK;
  const NetDims n{state_dim, hidden, num_layers, action_dim, activation};
  const int fwd = n.S + 2 * (n.L + 1) * n.H + n.L * n.H + n.L + 2 * n.A + 16;
  const size_t lds = sizeof(float) * (size_t)fwd;
  if (int rc = check_net(c, state_dim, hidden, num_layers, action_dim, activation, lds, "serl_ga_novelty")) return rc;
  HIP_TRY(hipSetDevice(c->device));
  hipLaunchKernelGGL(ga_novelty_kernel, dim3(n_pairs), dim3(256), lds, (hipStream_t)stream, weights, stride, n, members, states, actions,
                     batch, novelty);
  HIP_TRY(hipGetLastError());
  return SERL_OK;
}

int serl_replay_scatter(serl_ctx *c, const float *staged, int64_t rows_per_episode, const serl_replay_job *jobs, int32_t n_jobs,
                        void *stream)
{
  static_assert(sizeof(serl_replay_job) == sizeof(ReplayJob), "job layout");
  if (!c || !staged || !jobs || n_jobs < 0 || rows_per_episode <= 0) return serl_fail(SERL_E_INVALID, "serl_replay_scatter: bad argument");
  if (n_jobs == 0) return SERL_OK;
  HIP_TRY(hipSetDevice(c->device));
  hipLaunchKernelGGL(replay_scatter_kernel, dim3(n_jobs), dim3(256), 0, (hipStream_t)stream, staged, rows_per_episode,
                     reinterpret_cast<const ReplayJob *>(jobs));
  HIP_TRY(hipGetLastError());
  return SERL_OK;
}

}  // extern "C"
