// skid.hpp - rollouts of the skid-steer robot family: SkidSteerRobot.step (dust/models/skid_steer_robot.py:73-122) under
// MultiDISCO._rollout / _compute_cost (disco.py:139-209, 294-346) with a QUADRATIC cost family.  The reference ships no cost for this
// model (its MultiDISCO takes any callable); the family here is what `dust_amd.costs.QuadraticCost` evaluates on the host, so that
// the same object can be handed to the reference's controller:
//     inst(x, a) = sum_k w_state[k] (x_k - goal_k)^2 + sum_d w_ctrl[d] a_d^2          (state BEFORE the action, raw action)
//     term(x)    = sum_k w_term[k]  (x_k - goal_k)^2
// One lane = one (action sample s, policy n) pair, the M dynamics samples in sequence; costs go to a [S][N] buffer that the regular
// rollout kernel consumes in its injected-costs mode (weights, likelihood score, a_mat update) - the scheme of rollout_states.hpp.
// This family is a completeness row, not a tuned one: plain loads, no LDS staging.
//
// Arithmetic follows the reference's fp32 tensor expressions operation by operation (Python floats enter as fp32 scalars):
//   linear  = ((r + l) * pi) * wheel_radius            angular = ((((r - l) * 2) * pi) * wheel_radius) / axial_distance
//   forward = linear * dt                               lateral = ((-angular) * x_icr) * dt
//   x' = (x + forward cos th) - lateral sin th          y' = (y + forward sin th) + lateral cos th          th' = th + angular * dt
//   state' = (x', y', th', linear, angular)
#pragma once
#include "rollout.hpp"

namespace dust {

struct SkidModel {
  DevParam x_icr, wheel_radius, axial_distance;
  float lo[2], hi[2];  // action_space bounds (wheel speeds)
  float goal[5], w_state[5], w_term[5], w_ctrl[2];
};

struct SkidArgs {
  SkidModel sk;
  int N_total, n0, n_local, S, M, H, D, P;
  int noise_mode;       // NOISE_EPS / NOISE_ACTIONS / NOISE_PHILOX (rollout.hpp)
  int log_space, interleave;
  float dt;
  float chol_a[2];
  float chol_off;       // full a_cov: L[1][0] (0: diagonal)
  uint64_t seed;
  const uint32_t *ctr;  // {tick, iter, ..}: Philox stream position, as the regular kernel reads it
  const float *noise;   // [S][N][D] eps or actions
  const float *theta;   // [N][D]
  const float *state;   // [5]
  const float *params;  // [M][P] raw samples or nullptr
  float *costs_sn;      // [S][N]
  float *costsT;        // [N][S] the context's cost record (the regular kernel does not write it in its injected-costs mode)
  float *states_out;    // [M][S][N][H+1][5] or nullptr
  float *actions_out;   // [S][N][D] or nullptr
};

__device__ __forceinline__ float skid_param(const DevParam &p, const float *prow, int log_space) {
  if (p.kind == DUST_PARAM_SAMPLED && prow) {
    const float v = prow[p.col];
    return log_space ? expf(v) : v;
  }
  return (float)p.value;
}

__global__ __launch_bounds__(256) void skid_rollout_kernel(const SkidArgs a) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= a.n_local * a.S) return;
  const int s = idx / a.n_local, n = a.n0 + (idx - s * a.n_local);  // (n fastest: the rows of one sample are adjacent)
  const int D = a.D, H = a.H, N = a.N_total;
  const float *nz = a.noise ? a.noise + ((size_t)s * N + n) * D : nullptr;
  const float *th = a.theta + (size_t)n * D;
  const uint32_t ctr_tick = a.ctr[0], ctr_iter = a.ctr[1];
  auto action = [&](const int j) -> float {  // theta + L eps (an odd column of a full L takes its partner draw too)
    if (a.noise_mode == NOISE_ACTIONS) return nz[j];
    if (a.noise_mode == NOISE_EPS) return (j & 1) && a.chol_off != 0.f ? th[j] + (a.chol_off * nz[j - 1] + a.chol_a[1] * nz[j]) : th[j] + a.chol_a[j & 1] * nz[j];
    float z[8];
    philox_normal8(a.seed, (uint32_t)(j >> 3), (uint32_t)(s * N + n), ctr_iter, ctr_tick, z);  // (the regular kernel's stream)
    return (j & 1) && a.chol_off != 0.f ? th[j] + (a.chol_off * z[(j & 7) - 1] + a.chol_a[1] * z[j & 7]) : th[j] + a.chol_a[j & 1] * z[j & 7];
  };
  if (a.actions_out)
    for (int j = 0; j < D; ++j) a.actions_out[((size_t)s * N + n) * D + j] = action(j);
  float x0[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) x0[k] = a.state[k];
  double acc = 0.0;
  for (int m = 0; m < a.M; ++m) {
    // scalar-event params_dist quirk (disco.py:177-179): rollout r = (m, s, n) flattened uses params[r % M]
    const int mi = a.interleave ? (int)((((long)m * a.S + s) * N + n) % a.M) : m;
    const float *prow = a.params ? a.params + (size_t)mi * a.P : nullptr;
    const float xicr = skid_param(a.sk.x_icr, prow, a.log_space), wr = skid_param(a.sk.wheel_radius, prow, a.log_space),
                ad = skid_param(a.sk.axial_distance, prow, a.log_space);
    float x[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) x[k] = x0[k];
    float *so = a.states_out ? a.states_out + ((((size_t)m * a.S + s) * N + n) * (size_t)(H + 1)) * 5 : nullptr;
    if (so)
#pragma unroll
      for (int k = 0; k < 5; ++k) so[k] = x[k];
    double tot = 0.0;
    for (int t = 0; t < H; ++t) {
      const float a0 = action(2 * t), a1 = action(2 * t + 1);
      double sc = 0.0;
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const float d = x[k] - a.sk.goal[k];
        sc += (double)((d * d) * a.sk.w_state[k]);
      }
      const double cc = (double)((a0 * a0) * a.sk.w_ctrl[0]) + (double)((a1 * a1) * a.sk.w_ctrl[1]);
      tot += (double)((float)sc + (float)cc);
      const float r = clampf(a0, a.sk.lo[0], a.sk.hi[0]), l = clampf(a1, a.sk.lo[1], a.sk.hi[1]);
      const float lin = ((r + l) * PI_F) * wr;
      const float ang = ((((r - l) * 2.0f) * PI_F) * wr) / ad;
      const float fwd = lin * a.dt, lat = ((-ang) * xicr) * a.dt;
      const float cs = fast_cosf(x[2]), sn = fast_sinf(x[2]);
      const float nx = (x[0] + fwd * cs) - lat * sn;
      const float ny = (x[1] + fwd * sn) + lat * cs;
      x[2] = x[2] + ang * a.dt;
      x[0] = nx;
      x[1] = ny;
      x[3] = lin;
      x[4] = ang;
      if (so)
#pragma unroll
        for (int k = 0; k < 5; ++k) so[(size_t)(t + 1) * 5 + k] = x[k];
    }
    double tc = 0.0;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const float d = x[k] - a.sk.goal[k];
      tc += (double)((d * d) * a.sk.w_term[k]);
    }
    acc += (double)((float)tot + (float)tc);
  }
  const float cost = a.M == 1 ? (float)acc : (float)(acc / a.M);
  a.costs_sn[(size_t)s * N + n] = cost;
  a.costsT[(size_t)n * a.S + s] = cost;
}

}  // namespace dust
