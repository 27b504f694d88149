// particle_general.hpp - the Particle rollouts the specialised kernels (rollout.hpp packed pairs, rollout_states.hpp whole lines,
// persist.hpp) do not take: control-channel noise and velocity control.
//
// Replaces (reference file:line): Particle.step particle.py:117-166 in full -
//   * `if not self.deterministic: acts += self.dyn_std * torch.randn_like(acts)` (particle.py:145-148; deterministic=False is the
//     constructor DEFAULT, particle.py:31): every model.step call of MultiDISCO._rollout (disco.py:193-200) draws one [M*S*N, d_a]
//     normal tensor, so rollout r = (m*S + s)*N + n at step t uses draw [t][r].  The noise moves the DYNAMICS only: the costs see
//     the raw actions (disco.py:306-310);
//   * `control_type == "velocity"` (particle.py:41-48, 152-153): a TWO-state model (x, y): acts.clamp_(+-max_speed), x_dot = acts,
//     next = states + x_dot * dt * (1 - mask), and the closing `next_states[..., -2:].clamp_` (particle.py:165) then lands on the
//     POSITIONS (reproduced: it is what the reference computes);
// under MultiDISCO._rollout / _compute_cost (disco.py:139-209, 294-346) with Particle.default_inst_cost / default_term_cost
// (particle.py:170-225; w_state / w_term / target have dim_s entries).
//
// One lane = one (action sample s, policy n) pair, its action row in LDS, the M dynamics samples in sequence; costs go to a [S][N]
// buffer that the regular rollout kernel consumes in its injected-costs mode (weights, likelihood score, a_mat update).  Control noise:
// recorded draws `cz` [H][M*S*N][d_a] (parity runs: the reference's own torch.randn_like tensors, dust_set_ctrl_noise) or a Philox
// stream of its own, keyed (tick, iter, r, t / 2) - in registers, never in HBM.
#pragma once
#include "rollout.hpp"

namespace dust {

struct PartGenArgs {
  DevModel dm;
  int N_total, n0, n_local, S, M, H, D;
  int noise_mode;  // policy noise: NOISE_EPS / NOISE_ACTIONS / NOISE_PHILOX (rollout.hpp)
  int noise_f16;   // caller's eps / actions are binary16
  int store_f16;   // states_out is binary16
  int velocity;    // control_type == "velocity": dim_s = 2
  int mc;          // lanes per (sample, policy) pair: a power of two <= min(M, 8); lane c runs dynamics samples c, c + mc, ...
  int ctrl_noise;  // Particle(deterministic=False)
  float dyn_std[2];
  float chol_a[2], a_pre[2];
  float chol_off, a_pre_off;  // full a_cov (disco.py:91-98): L[1][0] and the off-diagonal entry of inverse(a_cov) (0: diagonal forms)
  float a_reg;
  uint64_t seed;
  const uint32_t *ctr;  // {tick, iter, ..}: Philox stream position, as the regular kernel reads it
  const float *noise;   // [S][N][D] eps or actions
  const float *theta;   // [N][D]
  const float *state;   // [ds]
  const float *params;  // [M][P] raw samples or nullptr
  const float *cz;      // [H][M*S*N][2] recorded control-noise draws, or nullptr: Philox
  const float *a_seq;   // [D]
  const float *a_mat;   // [N][D]
  float *costs_sn;      // [S][N]
  float *costsT;        // [N][S]
  void *states_out;     // [M][S][N][H+1][ds] or nullptr
};

enum { PARTGEN_NT = 128 };
static inline size_t particle_general_lds_bytes(int D, int grid_words, int mc) { return sizeof(float) * ((size_t)(PARTGEN_NT / mc) * (D + 1) + (size_t)grid_words); }

// Round 6: a (sample, policy) pair's action row is drawn ONCE, into an LDS row (the first form drew a whole
// philox_normal8 block for every column it looked at, again for every dynamics sample: 8 M times too often - 7.5 ms per cfg3 tick
// against 1.0 deterministic), and `mc` lanes share the pair: each runs every mc-th dynamics sample (a row per LANE left room for
// 1.5 waves per SIMD: 5.6 ms; the dependent chain of a rollout wants many); the occupancy grid sits in LDS beside the rows; one
// philox_normal8 serves the control noise of FOUR steps; the step cost is the reference's own fp32 products summed pairwise with a
// compensated running sum (the arithmetic of the other Particle kernels: rollout.hpp PairKahan) instead of fp64 adds.
__global__ __launch_bounds__(PARTGEN_NT) void particle_general_kernel(const PartGenArgs a) {
  extern __shared__ __attribute__((aligned(16))) float pg_lds[];
  const int D = a.D, H = a.H, N = a.N_total, tid = threadIdx.x;
  const int MC = a.mc, lr = tid / MC, mc = tid - lr * MC, rows = PARTGEN_NT / MC;
  float *arow = pg_lds + (size_t)lr * (D + 1);
  uint32_t *gridl = reinterpret_cast<uint32_t *>(pg_lds + (size_t)rows * (D + 1));
  DevModel dm = a.dm;
  const int gw = (dm.with_obstacle && dm.grid_bits) ? (dm.nx * dm.ny + 31) / 32 : 0;
  for (int w = tid; w < gw; w += PARTGEN_NT) gridl[w] = dm.grid_bits[w];
  if (gw) dm.grid_bits = gridl;
  const int idx = blockIdx.x * rows + lr;  // the (sample, policy) pair
  const bool live = idx < a.n_local * a.S;
  const int idc = live ? idx : 0;
  const int s = idc / a.n_local, n = a.n0 + (idc - s * a.n_local);  // (n fastest: the rows of one sample are adjacent)
  const int DSm = a.velocity ? 2 : 4;
  const size_t row = ((size_t)s * N + n) * D;
  const float *th = a.theta + (size_t)n * D;
  const uint32_t ctr_tick = a.ctr[0], ctr_iter = a.ctr[1];
  // the action row: theta + L eps (an odd column of a full L takes its partner draw too), or the caller's actions; the pair's lanes
  // share the work
  if (a.noise_mode == NOISE_PHILOX) {
    for (int j0 = 8 * mc; j0 < D; j0 += 8 * MC) {
      float z[8];
      philox_normal8(a.seed, (uint32_t)(j0 >> 3), (uint32_t)(s * N + n), ctr_iter, ctr_tick, z);  // (the regular kernel's stream)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int j = j0 + k;
        if (j < D) {
          const bool pair = (j & 1) && a.chol_off != 0.f;
          arow[j] = pair ? th[j] + (a.chol_off * z[k - 1 < 0 ? 0 : k - 1] + a.chol_a[1] * z[k]) : th[j] + a.chol_a[j & 1] * z[k];
        }
      }
    }
  } else {
    const float *nz = !a.noise_f16 ? a.noise + row : nullptr;
    const _Float16 *nzh = a.noise_f16 ? reinterpret_cast<const _Float16 *>(a.noise) + row : nullptr;
    for (int j = mc; j < D; j += MC) {
      const float v = nzh ? (float)nzh[j] : nz[j];
      if (a.noise_mode == NOISE_ACTIONS) arow[j] = v;
      else if ((j & 1) && a.chol_off != 0.f) arow[j] = th[j] + (a.chol_off * (nzh ? (float)nzh[j - 1] : nz[j - 1]) + a.chol_a[1] * v);
      else arow[j] = th[j] + a.chol_a[j & 1] * v;
    }
  }
  __syncthreads();  // the grid and the rows
  const float dt = (float)dm.dt;
  const bool obst = dm.with_obstacle != 0, crash = dm.can_crash && dm.with_obstacle;
  float x0[4] = {0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < DSm; ++k) x0[k] = a.state[k];
  const size_t SN = (size_t)a.S * N;
  double acc = 0.0;
  for (int m = mc; m < a.M; m += MC) {
    const size_t r = (size_t)m * SN + (size_t)s * N + n;
    // scalar-event params_dist quirk (disco.py:177-179): rollout r = (m, s, n) flattened uses params[r % M]
    const int mi = dm.interleave ? (int)(r % (size_t)a.M) : m;
    const Coef cf = make_coef(dm, a.params ? a.params + (size_t)mi * dm.P : nullptr);
    float x[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = x0[k];
    const size_t so = r * (size_t)(H + 1) * DSm;
    auto put = [&](const int t) {
      if (!a.states_out || !live) return;
      for (int k = 0; k < DSm; ++k) {
        if (a.store_f16) reinterpret_cast<_Float16 *>(a.states_out)[so + (size_t)t * DSm + k] = (_Float16)x[k];
        else reinterpret_cast<float *>(a.states_out)[so + (size_t)t * DSm + k] = x[k];
      }
    };
    put(0);
    float tot = 0.f, comp = 0.f;  // compensated running sum of the step costs
    float zc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < H; ++t) {
      const float a0 = arow[2 * t], a1 = arow[2 * t + 1];
      // cost of the state BEFORE the action, raw action (disco.py:306; particle.py:170-198)
      const float coll = obst ? collision(dm, x[0], x[1]) : 0.f;
      float sc;
      {
        const float d0 = x[0] - dm.target[0], d1 = x[1] - dm.target[1];
        const float t0 = (d0 * d0) * dm.w_state[0], t1 = (d1 * d1) * dm.w_state[1];
        sc = t0 + t1;
        if (!a.velocity) {
          const float d2 = x[2] - dm.target[2], d3 = x[3] - dm.target[3];
          sc = sc + ((d2 * d2) * dm.w_state[2] + (d3 * d3) * dm.w_state[3]);
        }
      }
      const float cc = (a0 * a0) * dm.w_ctrl[0] + (a1 * a1) * dm.w_ctrl[1];
      const float ob = obst ? dm.w_obs * coll : 0.0f;
      {
        const float term = (sc + cc) + ob;
        const float y = term - comp, tn = tot + y;
        comp = (tn - tot) - y;
        tot = tn;
      }
      // the action that drives the dynamics (particle.py:144-153)
      float u0 = a0, u1 = a1;
      if (a.ctrl_noise) {
        float z0, z1;
        if (a.cz) {
          const float *zp = a.cz + ((size_t)t * a.M * SN + r) * 2;
          z0 = zp[0];
          z1 = zp[1];
        } else {  // a stream of its own (key word 0x63747264 "ctrd"), one block of EIGHT normals per four steps (philox_normal8: the policy
                  // noise's generator - half the Philox rounds per normal of the four-normal block this loop drew from before)
          if ((t & 3) == 0) philox_normal8(a.seed ^ 0x6374726400000000ull, (uint32_t)r, (uint32_t)(r >> 32) ^ ((uint32_t)(t >> 2) << 8), ctr_iter, ctr_tick, zc);
          const int q = t & 3;
          z0 = q == 0 ? zc[0] : (q == 1 ? zc[2] : (q == 2 ? zc[4] : zc[6]));
          z1 = q == 0 ? zc[1] : (q == 1 ? zc[3] : (q == 2 ? zc[5] : zc[7]));
        }
        u0 = u0 + a.dyn_std[0] * z0;
        u1 = u1 + a.dyn_std[1] * z1;
      }
      const float om = crash ? 1.0f - coll : 1.0f;
      if (a.velocity) {
        u0 = clampf(u0, -dm.max_speed, dm.max_speed);
        u1 = clampf(u1, -dm.max_speed, dm.max_speed);
        if (crash) {
          x[0] = x[0] + (u0 * dt) * om;
          x[1] = x[1] + (u1 * dt) * om;
        } else {
          x[0] = x[0] + u0 * dt;
          x[1] = x[1] + u1 * dt;
        }
        x[0] = clampf(x[0], -dm.max_speed, dm.max_speed);  // particle.py:165 on a two-state row: the positions
        x[1] = clampf(x[1], -dm.max_speed, dm.max_speed);
      } else {
        u0 = clampf(u0 / cf.c0, -dm.max_acc, dm.max_acc);
        u1 = clampf(u1 / cf.c0, -dm.max_acc, dm.max_acc);
        const float xd[4] = {x[2], x[3], u0, u1};
        if (crash) {
#pragma unroll
          for (int k = 0; k < 4; ++k) x[k] = x[k] + (xd[k] * dt) * om;
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) x[k] = x[k] + xd[k] * dt;
        }
        x[2] = clampf(x[2], -dm.max_speed, dm.max_speed);
        x[3] = clampf(x[3], -dm.max_speed, dm.max_speed);
      }
      put(t + 1);
    }
    float tc;
    {
      const float d0 = x[0] - dm.target[0], d1 = x[1] - dm.target[1];
      tc = (d0 * d0) * dm.w_term[0] + (d1 * d1) * dm.w_term[1];
      if (!a.velocity) {
        const float d2 = x[2] - dm.target[2], d3 = x[3] - dm.target[3];
        tc = tc + ((d2 * d2) * dm.w_term[2] + (d3 * d3) * dm.w_term[3]);
      }
    }
    const float tob = obst ? dm.w_obs * collision(dm, x[0], x[1]) : 0.0f;
    acc += (double)(tot + (tc + tob));
  }
  for (int o = 1; o < MC; o <<= 1) acc += __shfl_xor(acc, o);  // the pair's lanes (consecutive lanes of one wave): every lane ends with the sum
  if (!live || mc != 0) return;
  float cost = a.M == 1 ? (float)acc : (float)(acc / a.M);
  if (a.a_reg != 0.0f) {  // disco.py:338-346, diagonal of the [S,N,N] tensordot only
    double cc = 0.0;
    for (int j = 0; j < D; ++j) {
      const float e = arow[j] - a.a_seq[j];
      float ap = a.a_mat[(size_t)n * D + j] * a.a_pre[j & 1];
      if (a.a_pre_off != 0.f) {  // (a_mat[n, t, :] @ a_pre)[d] with a full symmetric a_pre
        const float m0 = a.a_mat[(size_t)n * D + (j & ~1)], m1 = a.a_mat[(size_t)n * D + (j | 1)];
        ap = (j & 1) ? (m0 * a.a_pre_off + m1 * a.a_pre[1]) : (m0 * a.a_pre[0] + m1 * a.a_pre_off);
      }
      cc += (double)(-e) * (double)ap;
    }
    cost = cost + a.a_reg * (float)cc;
  }
  a.costs_sn[(size_t)s * N + n] = cost;
  a.costsT[(size_t)n * a.S + s] = cost;
}

}  // namespace dust
