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

// Actor forward for one lane: sequential f32 accumulation in index order (matches oracle/rollout_ref.c)
static __device__ void serl_actor_forward(const serl_rollout_desc &d, const float *__restrict__ w,
                                          const float *obs, float *act_out)
{
  const int S = d.state_dim, H = d.hidden, A = d.action_dim, L = d.num_layers;
  float h0[SERL_MAX_HIDDEN], h1[SERL_MAX_HIDDEN];
  const float *W = w, *b = w + (size_t)H * S;
  for (int i = 0; i < H; ++i) {
    float acc = b[i];
    for (int j = 0; j < S; ++j) acc = acc + W[i * S + j] * obs[j];
    h0[i] = serl_act(acc, d.activation);
  }
  w = b + H;
  for (int l = 0; l < L; ++l) {
    const float *Wl = w, *bl = w + (size_t)H * H, *g = bl + H, *be = g + H;
    for (int i = 0; i < H; ++i) {
      float acc = bl[i];
      for (int j = 0; j < H; ++j) acc = acc + Wl[i * H + j] * h0[j];
      h1[i] = acc;
    }
    float mean = 0.0f;
    for (int i = 0; i < H; ++i) mean = mean + h1[i];
    mean = mean / (float)H;
    float var = 0.0f;
    for (int i = 0; i < H; ++i) { float dl = h1[i] - mean; var = var + dl * dl; }
    float den = sqrtf(var / (float)(H - 1)) + 1e-6f;
    for (int i = 0; i < H; ++i) h0[i] = serl_act(g[i] * (h1[i] - mean) / den + be[i], d.activation);
    w = be + H;
  }
  const float *Wo = w, *bo = w + (size_t)A * H;
  for (int i = 0; i < A; ++i) {
    float acc = bo[i];
    for (int j = 0; j < H; ++j) acc = acc + Wo[i * H + j] * h0[j];
    act_out[i] = tanhf(acc);
  }
}

static __device__ __forceinline__ double serl_clip(double v, double lo, double hi)
{
  return v < lo ? lo : (v > hi ? hi : v);
}
