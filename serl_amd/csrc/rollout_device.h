// rollout_device.h -- the fused population-rollout kernel body (product code, gfx950).
//
// One episode = reset -> { actor MLP forward (f32) -> action scaling -> actuator faults -> dynamics
// step (ODE5 over the lifted model, f64) -> tracking error / reward / cost / bounds } until done,
// entirely inside one kernel launch; per step only ref[k] is read from HBM and (optionally) traces
// are written.  Mirrors, in this order:
//   base/core/agent.py:63-138 (Agent.evaluate)          base/core/genetic_agent.py:104-109 (Actor)
//   base/core/mod_utils.py:14-18,39-50 (activations, LayerNorm: unbiased std, eps added to std)
//   envs/phlabenv.py:62-73 (scale_action), :347-399 (reward/cost/bounds), :401-482 (reset/step)
//   envs/{be,jr,sa,se}/citation.py:71-79 (actuator faults as a per-episode row)
// Included once per dynamics code variant with DYN_STEP defined.
#pragma once
#include "../../include/serl_amd.h"

struct RolloutArgs {
  serl_rollout_desc d;
  const double *ro, *t3, *x0, *dw0;   // device copies of the build tables
  double dyn_dt;
  int32_t lanes;                      // active lanes per 64-lane wavefront
  int32_t block;                      // threads per workgroup (64 x wavefronts sharing one LDS table copy)
};

static __device__ __forceinline__ float serl_act(float v, int act)
{
  if (act == SERL_ACT_TANH) return tanhf(v);
  if (act == SERL_ACT_ELU) return v > 0.0f ? v : expm1f(v);
  return v > 0.0f ? v : 0.01f * v;
}

#define SERL_MAX_HIDDEN 128
#define SERL_BLOCK 256          // launch bound: up to 4 wavefronts (one per SIMD, 512 registers each) per workgroup share one LDS copy of the tables

// ---- actor MLP, wave-cooperative ----------------------------------------------------------------
// One forward pass of ONE member's actor by all 64 lanes of a wavefront: lane r owns hidden rows r and
// r+64; the previous layer's activations are broadcast lane-by-lane with v_readlane (the loop index is
// wave-uniform), so every row accumulates  acc = bias; acc += W[i][j]*h[j]  in index order j = 0..n-1 with
// separate multiply and add -- bit-identical to the sequential restatement in oracle/rollout_ref.c.
// LayerNorm (base/core/mod_utils.py:39-50): mean and the unbiased variance are summed in index order too
// (every lane redundantly), std = sqrt(var/(H-1)), y = gamma*(x-mean)/(std+1e-6)+beta.
// `w`, `obs` and the result are wave-uniform.
static __device__ __forceinline__ float serl_bcast(float v, int srclane)
{
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), srclane));
}

static __device__ void serl_actor_forward_wave(const serl_rollout_desc &d, const float *__restrict__ w,
                                               const float obs[7], float act_out[3])
{
  const int S = d.state_dim, H = d.hidden, A = d.action_dim, L = d.num_layers;
  const int lane = threadIdx.x & 63;
  const int i0 = lane < H ? lane : H - 1, i1 = lane + 64 < H ? lane + 64 : H - 1;   // clamped row ids
  const bool two = H > 64;
  float h0a, h0b = 0.0f;
  {
    const float *W = w, *b = w + (size_t)H * S;
    float acc0 = b[i0], acc1 = b[i1];
    float wa[7], wb[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) { wa[j] = W[i0 * 7 + j]; wb[j] = two ? W[i1 * 7 + j] : 0.0f; }
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      acc0 = acc0 + wa[j] * obs[j];
      if (two) acc1 = acc1 + wb[j] * obs[j];
    }
    h0a = serl_act(acc0, d.activation);
    if (two) h0b = serl_act(acc1, d.activation);
    w = b + H;
  }
  for (int l = 0; l < L; ++l) {
    const float *Wl = w, *bl = w + (size_t)H * H, *g = bl + H, *be = g + H;
    float acc0 = bl[i0], acc1 = bl[i1];
    const float *r0 = Wl + (size_t)i0 * H, *r1 = Wl + (size_t)i1 * H;
    // rows are walked in chunks of 32 columns: the 8 (16) dwordx4 loads of a chunk are issued together so
    // that one memory latency is paid per chunk, then the 32 multiply-adds run in index order
    for (int jc = 0; jc < H; jc += 32) {
      float wa[32], wb[32];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int j = jc + 4 * q;
        const float4 v = (j + 3 < H) ? *(const float4 *)(r0 + j) : make_float4(0.f, 0.f, 0.f, 0.f);
        wa[4 * q] = v.x; wa[4 * q + 1] = v.y; wa[4 * q + 2] = v.z; wa[4 * q + 3] = v.w;
        if (two) {
          const float4 u = (j + 3 < H) ? *(const float4 *)(r1 + j) : make_float4(0.f, 0.f, 0.f, 0.f);
          wb[4 * q] = u.x; wb[4 * q + 1] = u.y; wb[4 * q + 2] = u.z; wb[4 * q + 3] = u.w;
        }
      }
#pragma unroll
      for (int q = 0; q < 32; ++q) {
        const int j = jc + q;
        if (j < H) {
          const float hj = (j < 64) ? serl_bcast(h0a, j & 63) : serl_bcast(h0b, j & 63);
          acc0 = acc0 + wa[q] * hj;
          if (two) acc1 = acc1 + wb[q] * hj;
        }
      }
    }
    float mean = 0.0f;
    for (int i = 0; i < H; ++i) mean = mean + ((i < 64) ? serl_bcast(acc0, i) : serl_bcast(acc1, i - 64));
    mean = mean / (float)H;
    const float d0 = acc0 - mean, d1 = acc1 - mean;
    const float q0 = d0 * d0, q1 = d1 * d1;
    float var = 0.0f;
    for (int i = 0; i < H; ++i) var = var + ((i < 64) ? serl_bcast(q0, i) : serl_bcast(q1, i - 64));
    const float den = sqrtf(var / (float)(H - 1)) + 1e-6f;
    h0a = serl_act(g[i0] * d0 / den + be[i0], d.activation);
    if (two) h0b = serl_act(g[i1] * d1 / den + be[i1], d.activation);
    w = be + H;
  }
  {
    const float *Wo = w, *bo = w + (size_t)A * H;
    const int io = lane < A ? lane : A - 1;
    float acc = bo[io];
    const float *ro_ = Wo + (size_t)io * H;
    for (int jc = 0; jc < H; jc += 32) {
      float wa[32];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int j = jc + 4 * q;
        const float4 v = (j + 3 < H) ? *(const float4 *)(ro_ + j) : make_float4(0.f, 0.f, 0.f, 0.f);
        wa[4 * q] = v.x; wa[4 * q + 1] = v.y; wa[4 * q + 2] = v.z; wa[4 * q + 3] = v.w;
      }
#pragma unroll
      for (int q = 0; q < 32; ++q) {
        const int j = jc + q;
        if (j < H) {
          const float hj = (j < 64) ? serl_bcast(h0a, j & 63) : serl_bcast(h0b, j & 63);
          acc = acc + wa[q] * hj;
        }
      }
    }
    const float t = tanhf(acc);
    for (int i = 0; i < 3; ++i) act_out[i] = serl_bcast(t, i);
  }
}

static __device__ __forceinline__ double serl_clip(double v, double lo, double hi)
{
  return v < lo ? lo : (v > hi ? hi : v);
}
