// rollout.hpp - kernel A: policy-noise sampling + batched rollouts + trajectory costs + softmax weights +
// likelihood score + MPPI side update, one workgroup per Stein particle n.
//
// Replaces (reference file:line): CostLikelihood.sample likelihoods.py:81-101, MultiDISCO._rollout disco.py:139-209,
// PendulumModel.step pendulum.py:61-100 / Particle.step particle.py:117-166, MultiDISCO._compute_cost disco.py:294-346,
// MultiDISCO.forward disco.py:380-393 (omega, a_mat, eta) and the likelihood half of SVMPC.phi svmpc.py:44-54.
//
// Layout / mapping (MI355X): a workgroup owns particle n and its S action samples.  The S x D action tile
// (a[s][j] = theta[n][j] + L eps[s][n][j]) is staged ONCE in LDS with coalesced loads of the S contiguous D-float
// rows of eps (row stride N*D floats in HBM); every later use - H-step rollout (lane = sample s, the M dynamics samples
// looped in registers), weighted reductions over s for grad_lik and the a_mat update - reads LDS, never HBM again.
// Row stride in LDS is D|1 dwords so lane-strided reads are bank-conflict free.  Costs are written as costsT[n][s]
// (coalesced); the host-facing [S][N] view is produced on demand.
#pragma once
#include "common.hpp"

namespace dust {

enum { NOISE_EPS = 0, NOISE_ACTIONS = 1, NOISE_PHILOX = 2 };

struct RolloutArgs {
  DevModel dm;
  int N_total, n0, S, M, H, da, ds, D;
  int noise_mode;
  int lik;           // dust_likelihood
  int eps_base_mode; // 0: eps = a - a_seq (ext actions, disco.py:161-164); 1: eps = a - a_mat[n] (internal noise, 155-160)
  int update_a_mat;
  float alpha, temp, a_reg;
  float chol_a[4], sigma_a[4], a_pre[4];
  const float *state;   // [ds]
  const float *theta;   // [N_total][D] base of the noise (theta, or a_mat for MultiDISCO's own sampling)
  const float *noise;   // eps or actions [S][N_total][D] (device), or nullptr for Philox
  const float *params;  // [M][P] raw samples or nullptr
  const float *a_seq;   // [D]
  float *a_mat;         // [N_total][D]
  float *costsT;        // [N_total][S]
  const float *costs_in; // [S][N_total] or nullptr: stage-wise mode (SVMPC.phi with a user log_p): skip the rollouts
  float *grad_lik;      // [N_total][D]
  float *logl;          // [N_total]   likelihood.log_prob per particle
  float *eta;           // [N_total]   logsumexp_s(-c/temp)  (a_mix = softmax_n(eta), beta cancels)
  float *omegaT;        // [N_total][S] or nullptr
  float *actions_out;   // [S][N_total][D] or nullptr
  float *states_out;    // [M][S][N_total][H+1][ds] or nullptr
  uint64_t seed;
  uint32_t tick, iter;
};

template <int MODEL>
__global__ __launch_bounds__(256) void rollout_kernel(const RolloutArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, nt = blockDim.x;
  const int n = a.n0 + blockIdx.x;
  const int S = a.S, D = a.D, H = a.H, da = a.da, N = a.N_total;
  const int Dp = D | 1;
  float *tile = lds;                 // [S][Dp] actions
  float *cst = tile + (size_t)S * Dp;  // [S] costs -> weights
  float *omg = cst + S;              // [S] omega
  float *red = omg + S;              // [32] reduction scratch
  float *part = red + 32;            // [nt] partial sums for the weighted reductions (2 x nt)

  // ---- 1. stage the action tile (a1: actions = theta + L eps) ----
  const float *th = a.theta + (size_t)n * D;
  if (a.noise_mode == NOISE_PHILOX) {
    const int D4 = (D + 3) >> 2;
    for (int idx = tid; idx < S * D4; idx += nt) {
      const int s = idx / D4, j4 = idx - s * D4;
      float z[4];
      philox_normal4(a.seed, (uint32_t)j4, (uint32_t)(s * N + n), a.iter, a.tick, z);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int j = j4 * 4 + q;
        if (j < D) tile[s * Dp + j] = th[j] + a.chol_a[j % da] * z[q];
      }
    }
  } else {
    for (int idx = tid; idx < S * D; idx += nt) {
      const int s = idx / D, j = idx - s * D;
      const float e = a.noise[((size_t)s * N + n) * D + j];
      tile[s * Dp + j] = a.noise_mode == NOISE_EPS ? th[j] + a.chol_a[j % da] * e : e;
    }
  }
  __syncthreads();
  if (a.actions_out) {
    for (int idx = tid; idx < S * D; idx += nt) {
      const int s = idx / D, j = idx - s * D;
      a.actions_out[((size_t)s * N + n) * D + j] = tile[s * Dp + j];
    }
  }

  // ---- 2. rollouts: lane = sample s, dynamics samples m looped in registers (a2-a5) ----
  const long SN = (long)S * N;
  if (a.costs_in) {
    for (int s = tid; s < S; s += nt) cst[s] = a.costs_in[(size_t)s * N + n];
  } else
  for (int s = tid; s < S; s += nt) {
    const float *act = tile + s * Dp;
    double acc_m = 0.0;
    for (int m = 0; m < a.M; ++m) {
      const long r = (long)m * SN + (long)s * N + n;
      const float *prow = a.params ? a.params + (size_t)(a.dm.interleave ? (int)(r % a.M) : m) * a.dm.P : nullptr;
      const Coef cf = make_coef(a.dm, prow);
      float x[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) x[k] = k < a.ds ? a.state[k] : 0.f;
      float *so = a.states_out ? a.states_out + (size_t)r * (H + 1) * a.ds : nullptr;
      if (so)
        for (int k = 0; k < a.ds; ++k) so[k] = x[k];
      double tot = 0.0;
      for (int t = 0; t < H; ++t) {
        float at[2];
        at[0] = act[t * da];
        at[1] = da > 1 ? act[t * da + 1] : 0.f;
        tot += (double)inst_cost<MODEL>(a.dm, x, at);  // cost of the state BEFORE the action (disco.py:306)
        model_step<MODEL>(a.dm, cf, x, at);
        if (so)
          for (int k = 0; k < a.ds; ++k) so[(size_t)(t + 1) * a.ds + k] = x[k];
      }
      const float traj = (float)tot + term_cost<MODEL>(a.dm, x);
      acc_m += (double)traj;
    }
    float cost = (float)(acc_m / a.M);
    if (a.a_reg != 0.0f) {  // disco.py:338-346, diagonal of the [S,N,N] tensordot only
      double cc = 0.0;
      for (int j = 0; j < D; ++j) {
        const float e = act[j] - a.a_seq[j];
        cc += (double)(-e) * (double)(a.a_mat[(size_t)n * D + j] * a.a_pre[j % da]);
      }
      cost = cost + a.a_reg * (float)cc;
    }
    cst[s] = cost;
    a.costsT[(size_t)n * S + s] = cost;
  }
  __syncthreads();

  // ---- 3. softmax over samples: likelihood weights w (alpha) and MPPI weights omega (1/temp) ----
  float cmin = INFINITY, csum = 0.f;
  for (int s = tid; s < S; s += nt) {
    cmin = fminf(cmin, cst[s]);
    csum += cst[s];
  }
  cmin = block_reduce<RED_MIN>(cmin, red);
  csum = block_reduce<RED_SUM>(csum, red);
  float zw = 0.f, zo = 0.f;
  for (int s = tid; s < S; s += nt) {
    const float c = cst[s];
    const float lo = (-1.0f * (c - cmin)) / a.temp;  // disco.py:381 with beta := per-policy min (beta cancels in omega)
    const float eo = expf(lo);
    const float ew = expf(-c * a.alpha - (-cmin * a.alpha));  // svmpc.py:51 softmax(-costs * alpha)
    omg[s] = eo;
    zo += eo;
    cst[s] = ew;
    zw += ew;
  }
  zw = block_reduce<RED_SUM>(zw, red);
  zo = block_reduce<RED_SUM>(zo, red);
  for (int s = tid; s < S; s += nt) {
    cst[s] = cst[s] / zw;
    omg[s] = omg[s] / zo;
    if (a.omegaT) a.omegaT[(size_t)n * S + s] = omg[s];
  }
  if (tid == 0 && !a.costs_in) {
    if (a.lik == DUST_LIK_EXP_UTILITY)  // likelihoods.py:127-135
      a.logl[n] = ((-cmin * a.alpha) + logf(zw)) - logf((float)S);
    else  // likelihoods.py:113-119
      a.logl[n] = -a.alpha * (csum / (float)S);
    a.eta[n] = (-cmin / a.temp) + logf(zo);
  }
  __syncthreads();

  // ---- 4. weighted reductions over s: grad_lik (svmpc.py:52-54) and a_mat += sum_s omega eps (disco.py:387-392) ----
  const int Q = nt / D > 0 ? nt / D : 1;
  float g = 0.f, am = 0.f;
  const int j = tid % D, q = tid / D;
  if (q < Q) {
    const float thj = th[j];
    const float s2 = a.sigma_a[j % da] * a.sigma_a[j % da];
    const float base = a.eps_base_mode ? thj : a.a_seq[j];
    for (int s = q; s < S; s += Q) {
      const float av = tile[s * Dp + j];
      g = fmaf(cst[s], (av - thj) / s2, g);
      am = fmaf(omg[s], av - base, am);
    }
  }
  part[tid] = g;
  part[nt + tid] = am;
  __syncthreads();
  for (int jj = tid; jj < D; jj += nt) {
    float gs = 0.f, as = 0.f;
    for (int qq = 0; qq < Q; ++qq) {
      gs += part[qq * D + jj];
      as += part[nt + qq * D + jj];
    }
    a.grad_lik[(size_t)n * D + jj] = gs;
    if (a.update_a_mat) a.a_mat[(size_t)n * D + jj] += as;
  }
}

static inline size_t rollout_lds_bytes(int S, int D, int nt) { return sizeof(float) * ((size_t)S * (D | 1) + 2 * (size_t)S + 32 + 2 * (size_t)nt); }

}  // namespace dust
