// serl_distill.hip -- the training loop of the distillation crossover as ONE launch for all pairs of an epoch.
//
// Reference: SSNE.distilation_crossover (base/core/mod_neuro_evo.py:131-147) trains every child for 12 x (len(buffer) // 128)
// Adam steps of GeneticAgent.update_parameters (base/core/genetic_agent.py:22-59): Q-filtered behaviour cloning on minibatches
// of 128 states.  As PyTorch on the GPU a step is ~200 tiny launches -- 750 steps x ~15 pairs took 25 - 33 s per generation
// (tools/gen_timing.py --epoch), 200 x the generation's whole evaluation stage.  What a step needs of the parents and the
// critic -- the target action of every state of the child's buffer and whether the Q-filter keeps the state -- does not
// depend on the child, so it is computed once per pair up front (distill.py, batched torch).  What is left is supervised
// regression of a 7-32x4-3 MLP on minibatches the host has already drawn (the reference's `random.sample` stream): this
// kernel.  One workgroup per pair, one thread per sample of the minibatch: forward and backward of the sample in
// registers, weights / gradients / Adam moments of the child in LDS, the batch reductions of the weight gradients as small
// LDS-staged matrix products, Adam applied by all threads -- all steps of a pair inside one launch, pairs side by side on
// different CUs.  f32 like torch; summation orders differ from torch's (agreement to rounding, tests/test_gpu_ga.py).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include "serl_ctx.h"

namespace {

constexpr int DB = 128;                 // threads per workgroup = the largest minibatch (mod_neuro_evo.py:137)
constexpr int DS_MAX = 16;              // observations of the widest env configuration

__device__ __forceinline__ float d_act(float v, int act)
{
  if (act == SERL_ACT_TANH) return tanhf(v);
  if (act == SERL_ACT_ELU) return v > 0.0f ? v : expm1f(v);
  return v > 0.0f ? v : 0.01f * v;
}

// derivative of the activation, from its VALUE a (tanh: 1 - a^2; ELU: a > 0 ? 1 : a + 1; LeakyReLU: a > 0 ? 1 : 0.01)
__device__ __forceinline__ float d_dact(float a, int act)
{
  if (act == SERL_ACT_TANH) return 1.0f - a * a;
  if (act == SERL_ACT_ELU) return a > 0.0f ? 1.0f : a + 1.0f;
  return a > 0.0f ? 1.0f : 0.01f;
}

struct DistillArgs {
  float *child;                 // [pairs][stride]  in: the second parent's parameters, out: the trained child
  int64_t stride;
  const float *states;          // [pairs][rows][S]   the child's buffer
  const float *targets;         // [pairs][rows][A]   the better parent's action per state
  const float *keep;            // [pairs][rows]      1 = the Q-filter keeps the state, 0 = dropped
  const int32_t *slots;         // [pairs][steps][batch] rows of the minibatches, in the reference's sampling order
  const int32_t *n_steps;       // [pairs]
  const int32_t *batch;         // [pairs] (<= 128)
  int32_t rows, steps, S, A, L, act;
  float lr, beta1, beta2, eps;
};

// G[i0 .. i0+1][j0 .. j0+3] = sum_b D[i][b] * X[j][b]   (D, X staged transposed: row = feature, DB samples per row)
__device__ __forceinline__ void tile_2x4(const float *D, const float *X, int i0, int j0, float (&acc)[2][4])
{
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[r][c] = 0.0f;
  for (int b = 0; b < DB; b += 4) {
    float4 d[2], x[4];
#pragma unroll
    for (int r = 0; r < 2; ++r) d[r] = *reinterpret_cast<const float4 *>(D + (size_t)(i0 + r) * DB + b);
#pragma unroll
    for (int c = 0; c < 4; ++c) x[c] = *reinterpret_cast<const float4 *>(X + (size_t)(j0 + c) * DB + b);
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        acc[r][c] += d[r].x * x[c].x + d[r].y * x[c].y + d[r].z * x[c].z + d[r].w * x[c].w;
  }
}

__device__ __forceinline__ float row_sum(const float *R)
{
  float s = 0.0f;
  for (int b = 0; b < DB; b += 4) {
    const float4 v = *reinterpret_cast<const float4 *>(R + b);
    s += (v.x + v.y) + (v.z + v.w);
  }
  return s;
}

__device__ __forceinline__ float row_dot(const float *R, const float *Q)
{
  float s = 0.0f;
  for (int b = 0; b < DB; b += 4) {
    const float4 v = *reinterpret_cast<const float4 *>(R + b), u = *reinterpret_cast<const float4 *>(Q + b);
    s += (v.x * u.x + v.y * u.y) + (v.z * u.z + v.w * u.w);
  }
  return s;
}

// Dynamic LDS (floats): w[P4] | g[P4] | m[P4] | v[P4] | HA[L + 1][H][DB] | D[H][DB] | Y[H][DB]   (P4 = parameter count rounded up
// to 4).  HA[l][i][b] = post-activation i of layer l for sample b: a thread touches only its own column b of HA / D / Y
// outside the staged reductions (conflict-free: consecutive threads, consecutive banks); the weight-gradient products read
// HA as their second operand directly.  Loops over the output row i stay rolled (LDS-indexed), the inner loops over the 32
// inputs are unrolled over registers: a fully unrolled network spills thousands of registers.
template <int H, int L>
__global__ void __launch_bounds__(DB) distill_kernel(DistillArgs a)
{
  static_assert(H == 32, "the 2 x 4 gradient tiles below cover a 32 x 32 matrix with 128 threads");
  extern __shared__ float lds[];
  const int p = blockIdx.x, t = threadIdx.x;
  const int S = a.S, A = a.A, act = a.act;
  const int P = H * S + H + L * (H * H + 3 * H) + A * H + A;
  const int P4 = (P + 3) & ~3;
  float *w = lds, *g = w + P4, *am = g + P4, *av = am + P4, *HA = av + P4, *D = HA + (size_t)(L + 1) * H * DB, *Y = D + (size_t)H * DB;
  float *row = a.child + (size_t)p * a.stride;
  for (int e = t; e < P4; e += DB) { w[e] = e < P ? row[e] : 0.0f; g[e] = 0.0f; am[e] = 0.0f; av[e] = 0.0f; }
  __syncthreads();
  const int steps = a.n_steps[p], B = a.batch[p];
  const float *st_p = a.states + (size_t)p * a.rows * S;
  const float *tg_p = a.targets + (size_t)p * a.rows * A;
  const float *kp_p = a.keep + (size_t)p * a.rows;
  const int32_t *sl_p = a.slots + (size_t)p * a.steps * DB;
  // packed-row offsets
  const int oW0 = 0, ob0 = H * S, oL = H * S + H, lstride = H * H + 3 * H, oWo = oL + L * lstride, obo = oWo + A * H;
  double b1t = 1.0, b2t = 1.0;
  const int ti = (t >> 3) * 2, tj = (t & 7) * 4;       // this thread's 2 x 4 tile of an [H][H] gradient (16 x 8 tiles)

  // y_i = b_i + sum_j W[i][j] hp[j] for all rows i of one hidden layer -> Y[i][t]; returns the sum of the y_i
  auto matvec = [&](const float *Wl, const float *bl, const float (&hp)[H]) {
    float sum = 0.0f;
#pragma unroll 2
    for (int i = 0; i < H; ++i) {
      float acc = bl[i];
#pragma unroll
      for (int q = 0; q < H / 4; ++q) {
        const float4 wv = *reinterpret_cast<const float4 *>(Wl + i * H + 4 * q);
        acc += wv.x * hp[4 * q] + wv.y * hp[4 * q + 1] + wv.z * hp[4 * q + 2] + wv.w * hp[4 * q + 3];
      }
      Y[(size_t)i * DB + t] = acc;
      sum += acc;
    }
    return sum;
  };

  for (int step = 0; step < steps; ++step) {
    const bool live = t < B;
    const int slot = live ? sl_p[(size_t)step * DB + t] : 0;
    float s[DS_MAX];
#pragma unroll
    for (int j = 0; j < DS_MAX; ++j) s[j] = (live && j < S) ? st_p[(size_t)slot * S + j] : 0.0f;
    const float keep = live ? kp_p[slot] : 0.0f;
    // ---------------- forward (genetic_agent.py:104, LayerNorm mod_utils.py:47-50)
    float mean[L], sd[L];
#pragma unroll 2
    for (int i = 0; i < H; ++i) {
      float acc = w[ob0 + i];
#pragma unroll
      for (int j = 0; j < DS_MAX; ++j) if (j < S) acc += w[oW0 + i * S + j] * s[j];
      HA[(size_t)i * DB + t] = d_act(acc, act);
    }
    float hp[H];
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const float *Wl = w + oL + l * lstride, *bl = Wl + H * H, *gam = bl + H, *bet = gam + H;
      const float *Hin = HA + (size_t)l * H * DB;
      float *Hout = HA + (size_t)(l + 1) * H * DB;
#pragma unroll
      for (int j = 0; j < H; ++j) hp[j] = Hin[(size_t)j * DB + t];
      mean[l] = matvec(Wl, bl, hp) / (float)H;
      float var = 0.0f;
      for (int i = 0; i < H; ++i) { const float c = Y[(size_t)i * DB + t] - mean[l]; var += c * c; }
      sd[l] = sqrtf(var / (float)(H - 1));
      const float Dn = sd[l] + 1e-6f;
      for (int i = 0; i < H; ++i) Hout[(size_t)i * DB + t] = d_act(gam[i] * (Y[(size_t)i * DB + t] - mean[l]) / Dn + bet[i], act);
    }
    const float *HL = HA + (size_t)L * H * DB;
#pragma unroll
    for (int j = 0; j < H; ++j) hp[j] = HL[(size_t)j * DB + t];
    float out[4], dpre[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      out[k] = 0.0f;
      if (k < A) {
        float acc = w[obo + k];
#pragma unroll
        for (int j = 0; j < H; ++j) acc += w[oWo + k * H + j] * hp[j];
        out[k] = tanhf(acc);
      }
    }
    // ---------------- loss: sum (out - target)^2 + mean(out^2) over the kept states (genetic_agent.py:49-52)
    const int kept = __syncthreads_count(keep != 0.0f);
    if (kept == 0) continue;             // (the reference's mean over an empty batch is NaN; no step is taken here)
    const float inv_n = 1.0f / (float)(kept * A);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float tgt = (live && k < A) ? tg_p[(size_t)slot * A + k] : 0.0f;
      dpre[k] = k < A ? keep * (2.0f * (out[k] - tgt) + 2.0f * out[k] * inv_n) * (1.0f - out[k] * out[k]) : 0.0f;
    }
    // ---------------- backward.  Output layer: dWo = dpre^T h_L, dbo = sum dpre, dh_L = Wo^T dpre
#pragma unroll
    for (int k = 0; k < 4; ++k) Y[(size_t)k * DB + t] = dpre[k];
    __syncthreads();
    for (int e = t; e < A * H; e += DB) g[oWo + e] = row_dot(Y + (size_t)(e / H) * DB, HL + (size_t)(e % H) * DB);
    if (t < A) g[obo + t] = row_sum(Y + (size_t)t * DB);
#pragma unroll
    for (int j = 0; j < H; ++j) {
      float acc = 0.0f;
#pragma unroll
      for (int k = 0; k < 4; ++k) if (k < A) acc += w[oWo + k * H + j] * dpre[k];
      D[(size_t)j * DB + t] = acc;                                   // D = dL / dh_L
    }
    __syncthreads();
#pragma unroll
    for (int l = L - 1; l >= 0; --l) {
      const float *Wl = w + oL + l * lstride, *bl = Wl + H * H, *gam = bl + H;
      float *gW = g + oL + l * lstride, *gb = gW + H * H, *gg = gb + H, *gbe = gg + H;
      const float *Hin = HA + (size_t)l * H * DB, *Hout = HA + (size_t)(l + 1) * H * DB;
      const float Dn = sd[l] + 1e-6f;
      // the layer's pre-LayerNorm values again (not kept: LDS holds the activations of all layers for 128 samples)
#pragma unroll
      for (int j = 0; j < H; ++j) hp[j] = Hin[(size_t)j * DB + t];
      matvec(Wl, bl, hp);
      // through the activation: dz -> D; the normalised value n -> Y; sums for the LayerNorm backward
      float gmean = 0.0f, gdotc = 0.0f;
      for (int i = 0; i < H; ++i) {
        const float dz = D[(size_t)i * DB + t] * d_dact(Hout[(size_t)i * DB + t], act);
        const float c = Y[(size_t)i * DB + t] - mean[l];
        D[(size_t)i * DB + t] = dz;
        Y[(size_t)i * DB + t] = c / Dn;
        gmean += dz * gam[i];
        gdotc += dz * gam[i] * c;
      }
      __syncthreads();
      if (t < H) { gg[t] = row_dot(D + (size_t)t * DB, Y + (size_t)t * DB); gbe[t] = row_sum(D + (size_t)t * DB); }   // dgamma, dbeta
      __syncthreads();
      // through the LayerNorm (unbiased std, eps on the std): dy = (dn - mean(dn)) / D - c * sum(dn c) / (D^2 (H - 1) sd)
      gmean = gmean / (float)H;
      const float k2 = sd[l] > 0.0f ? gdotc / (Dn * Dn * (float)(H - 1) * sd[l]) : 0.0f;
      for (int i = 0; i < H; ++i) {
        const float c = Y[(size_t)i * DB + t] * Dn;
        D[(size_t)i * DB + t] = (D[(size_t)i * DB + t] * gam[i] - gmean) / Dn - k2 * c;        // dy_i
      }
      __syncthreads();
      {
        float acc[2][4];
        tile_2x4(D, Hin, ti, tj, acc);                                // dW_l = dy^T h_{l-1}
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) gW[(ti + r) * H + tj + cc] = acc[r][cc];
      }
      if (t < H) gb[t] = row_sum(D + (size_t)t * DB);
      // dh_{l-1} = Wl^T dy
      float dh[H];
#pragma unroll
      for (int j = 0; j < H; ++j) dh[j] = 0.0f;
#pragma unroll 2
      for (int i = 0; i < H; ++i) {
        const float dy = D[(size_t)i * DB + t];
#pragma unroll
        for (int q = 0; q < H / 4; ++q) {
          const float4 wv = *reinterpret_cast<const float4 *>(Wl + i * H + 4 * q);
          dh[4 * q] += wv.x * dy; dh[4 * q + 1] += wv.y * dy; dh[4 * q + 2] += wv.z * dy; dh[4 * q + 3] += wv.w * dy;
        }
      }
      __syncthreads();                                                // (the products above have read all of D)
#pragma unroll
      for (int j = 0; j < H; ++j) D[(size_t)j * DB + t] = dh[j];
    }
    // first layer: dW0 = dz0^T s, db0 = sum dz0
    for (int i = 0; i < H; ++i) D[(size_t)i * DB + t] = D[(size_t)i * DB + t] * d_dact(HA[(size_t)i * DB + t], act);
#pragma unroll
    for (int j = 0; j < DS_MAX; ++j) if (j < S) Y[(size_t)j * DB + t] = s[j];
    __syncthreads();
    for (int e = t; e < H * S; e += DB) g[oW0 + e] = row_dot(D + (size_t)(e / S) * DB, Y + (size_t)(e % S) * DB);
    if (t < H) g[ob0 + t] = row_sum(D + (size_t)t * DB);
    __syncthreads();
    // ---------------- Adam (torch.optim.Adam defaults: lr 1e-3, betas (0.9, 0.999), eps 1e-8; genetic_agent.py:17)
    b1t *= (double)a.beta1; b2t *= (double)a.beta2;
    const float step_size = (float)((double)a.lr / (1.0 - b1t));
    const float bc2_sqrt = (float)sqrt(1.0 - b2t);
    for (int e = t; e < P; e += DB) {
      const float ge = g[e];
      const float m1 = am[e] + (ge - am[e]) * (1.0f - a.beta1);
      const float v1 = av[e] * a.beta2 + ge * ge * (1.0f - a.beta2);
      am[e] = m1; av[e] = v1;
      w[e] -= step_size * m1 / (sqrtf(v1) / bc2_sqrt + a.eps);
    }
    __syncthreads();
  }
  for (int e = t; e < P; e += DB) row[e] = w[e];
}

}  // namespace

extern "C" {

// The rows `random.sample(memory, k)` picks from a list of n transitions (base/core/replay_memory.py:72-73, 83-85), for `calls`
// consecutive calls, replayed from the raw 32-bit outputs of the generator.  CPython's sample() draws j = _randbelow(m)
// = getrandbits(m.bit_length()) -- one 32-bit output shifted right -- rejected while >= m; for n above its set-size threshold
// (21 + 4 ** ceil(log4(3 k)) for k > 5) it draws from m = n and rejects a j it already holds, below it draws from a shrinking
// pool (m = n - i, result[i] = pool[j], pool[j] = pool[m - 1]).  words: the next outputs of the Mersenne Twister in order
// (random.getrandbits(32 * m) split little-endian); out: int32 [calls][out_stride]; returns the number of words consumed, or
// -1 if `n_words` did not suffice, -2 for bad arguments.  Host code (index logic, no compute).
long long serl_host_sample_slots(const uint32_t *words, long long n_words, int32_t n, int32_t k, int32_t calls, int32_t *out,
                                 int32_t out_stride)
{
  if (!words || !out || n <= 0 || k <= 0 || k > n || calls < 0 || out_stride < k) return -2;
  long long setsize = 21;
  if (k > 5) {
    long long pw = 1;
    while (pw < 3LL * k) pw *= 4;
    setsize += pw;
  }
  auto bit_length = [](uint32_t v) { int b = 0; for (; v; v >>= 1) ++b; return b; };
  long long pos = 0;
  if ((long long)n <= setsize) {                       // pool branch
    int32_t *pool = (int32_t *)malloc((size_t)n * sizeof(int32_t));
    if (!pool) return -2;
    for (int c = 0; c < calls; ++c) {
      int32_t *o = out + (size_t)c * out_stride;
      for (int i = 0; i < n; ++i) pool[i] = i;
      for (int i = 0; i < k; ++i) {
        const uint32_t m = (uint32_t)(n - i);
        const int shift = 32 - bit_length(m);
        uint32_t j;
        do {
          if (pos >= n_words) { free(pool); return -1; }
          j = words[pos++] >> shift;
        } while (j >= m);
        o[i] = pool[j];
        pool[j] = pool[m - 1];
      }
    }
    free(pool);
    return pos;
  }
  const int shift = 32 - bit_length((uint32_t)n);
  unsigned char *seen = (unsigned char *)calloc((size_t)n, 1);
  if (!seen) return -2;
  for (int c = 0; c < calls; ++c) {
    int32_t *o = out + (size_t)c * out_stride;
    for (int i = 0; i < k; ++i) {
      uint32_t j;
      for (;;) {
        if (pos >= n_words) { free(seen); return -1; }
        j = words[pos++] >> shift;
        if (j >= (uint32_t)n) continue;          // _randbelow_with_getrandbits
        if (seen[j]) continue;                   // `while j in selected`
        break;
      }
      seen[j] = 1;
      o[i] = (int32_t)j;
    }
    for (int i = 0; i < k; ++i) seen[o[i]] = 0;
  }
  free(seen);
  return pos;
}

int serl_ga_distill(serl_ctx *c, float *child, int64_t stride, int32_t n_pairs, int32_t state_dim, int32_t hidden, int32_t num_layers,
                    int32_t action_dim, int32_t activation, const float *states, const float *targets, const float *keep, int32_t rows,
                    const int32_t *slots, int32_t steps, const int32_t *n_steps, const int32_t *batch, float lr, void *stream)
{
  if (!c || !child || !states || !targets || !keep || !slots || !n_steps || !batch || n_pairs < 0 || rows <= 0 || steps < 0)
    return serl_fail(SERL_E_INVALID, "serl_ga_distill: bad argument");
  if (hidden != 32 || num_layers != 3 || state_dim < 1 || state_dim > DS_MAX || action_dim < 1 || action_dim > 4 || activation < 0 || activation > 2)
    return serl_fail(SERL_E_UNSUPPORTED, "serl_ga_distill: compiled for the SERL50 actor family (hidden 32, 3 hidden layers); other shapes train in PyTorch");
  HIP_TRY(hipSetDevice(c->device));
  DistillArgs a;
  a.child = child; a.stride = stride; a.states = states; a.targets = targets; a.keep = keep; a.slots = slots; a.n_steps = n_steps;
  a.batch = batch; a.rows = rows; a.steps = steps; a.S = state_dim; a.A = action_dim; a.L = num_layers; a.act = activation;
  a.lr = lr; a.beta1 = 0.9f; a.beta2 = 0.999f; a.eps = 1e-8f;
  const int P = hidden * state_dim + hidden + num_layers * (hidden * hidden + 3 * hidden) + action_dim * hidden + action_dim;
  const size_t lds = ((size_t)4 * ((P + 3) & ~3) + (size_t)(num_layers + 3) * hidden * DB) * sizeof(float);
  if (lds > (size_t)c->lds_per_block)
    return serl_fail(SERL_E_UNSUPPORTED, "serl_ga_distill: the training state does not fit this device's LDS per workgroup; the caller trains pair by pair in PyTorch");
  if (n_pairs == 0 || steps == 0) return SERL_OK;      // (n_pairs = 0 is the caller's probe: "would this shape run here?")
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(distill_kernel<32, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((distill_kernel<32, 3>), dim3(n_pairs), dim3(DB), lds, (hipStream_t)stream, a);
  HIP_TRY(hipGetLastError());
  return SERL_OK;
}

}  // extern "C"
