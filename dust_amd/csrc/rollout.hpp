// rollout.hpp - kernel A: policy-noise sampling + batched rollouts + trajectory costs + softmax weights +
// likelihood score + MPPI side update (+ combine of the prior-score partials), one workgroup per Stein particle n.
//
// Replaces (reference file:line): CostLikelihood.sample likelihoods.py:81-101, MultiDISCO._rollout disco.py:139-209,
// PendulumModel.step pendulum.py:61-100 / Particle.step particle.py:117-166, MultiDISCO._compute_cost disco.py:294-346,
// MultiDISCO.forward disco.py:380-393 (omega, a_mat, eta) and the likelihood half of SVMPC.phi svmpc.py:44-56.
//
// Layout / mapping (MI355X): a workgroup owns particle n and its S action samples.  The S x D action tile
// (a[s][j] = theta[n][j] + L eps[s][n][j]) is staged ONCE in LDS; every later use - H-step rollout (lane = sample s, the
// M dynamics samples looped in registers), weighted reductions over s for grad_lik and the a_mat update - reads LDS,
// never HBM again.  Row stride in LDS is D|1 dwords so lane-strided reads are bank-conflict free.
//   * caller-supplied noise: the S contiguous D-float rows of eps (row stride N*D floats in HBM) are fetched with
//     coalesced loads, 8 in flight per lane before the first LDS write (the kernel is latency-, not bandwidth-bound);
//   * no noise supplied: each lane fills ITS OWN tile row from a Philox counter stream and rolls it out straight away -
//     eps never exists in HBM and no workgroup barrier separates sampling from the rollout.
// Costs are written as costsT[n][s] (coalesced); the host-facing [S][N] view is produced on demand.
// At N*S = 131072 rollouts a launch is only 2 waves per SIMD, so the kernel is instruction-issue bound: the pendulum
// step is specialised (one argument reduction serves sin(theta+pi) and cos(theta), see pendulum_trig).
#pragma once
#include <type_traits>

#include "common.hpp"
#include "stein.hpp"

#ifndef DUST_STG_CHUNK
#define DUST_STG_CHUNK 128  // bytes of a trajectory staged per flush of the stored-states form (rollout_body)
#endif
#ifndef DUST_STG_CHUNK_PART
#define DUST_STG_CHUNK_PART 64  // the same for the Particle family (two staged trajectories per lane)
#endif

namespace dust {

enum { NOISE_EPS = 0, NOISE_ACTIONS = 1, NOISE_PHILOX = 2 };
enum { CNT_STRIDE = 32 };  // dwords between in-launch arrival counters: one 128-byte line each

struct RolloutArgs {
  DevModel dm;
  int N_total, n0, S, M, H, D;
  int noise_mode;
  int noise_f16;     // the caller's eps / actions are IEEE binary16 (storage only: converted on load)
  int store_f16;     // states_out / actions_out are binary16 buffers
  int wq_iters;      // ceil(S / (nt / D)): trip count of the weighted reductions over s
  int lgW;           // log2 of the lanes per action row in the noise staging (2^lgW >= D)
  int G;             // dynamics-sample groups: lane = (sample, group), group g rolls out m = g, g+G, ... (G > 1 only when S <= nt/G)
  int lik;           // dust_likelihood
  int eps_base_mode; // 0: eps = a - a_seq (ext actions, disco.py:161-164); 1: eps = a - a_mat[n] (internal noise, 155-160)
  int update_a_mat;
  int merge_prior;   // combine the prior-pass partials and write grad_pri / score = grad_lik + grad_pri
  uint32_t magicD;   // floor(2^32 / D) + 1: exact division of tile indices by D
  float alpha, temp, a_reg;
  float chol_a[4], sigma_a[4], a_pre[4];
  int full_cov;        // full 2 x 2 a_cov (disco.py:91-98): actions = theta + L eps with L[1][0] = chol_off, control cost through the full
  float chol_off;      // a_pre (a_pre_off = its off-diagonal entry)
  float a_pre_off;
  PriorMerge pm;
  const float *state;   // [ds] (device; refreshed by a 4-lane launch carrying the plant state as its argument before each tick)
  uint32_t *ctr;        // device counters {tick, iter, adam_step}: the Philox stream position (static under hipGraph replay)
  int bump_adam;        // an optimiser step follows: advance adam_step (read by update_kernel, never by this kernel)
  unsigned int *rearm;  // arrival counters of the Stein+update launch (fused.hpp), zeroed here for its next use, or nullptr
  int rearm_n;
  int grid_words;       // Particle: the bit-packed occupancy grid is staged into LDS (this many words, a multiple of 4), or 0: HBM.
                        // A lookup from HBM is a vector load whose in-order vmcnt wait also drains every state store in flight
                        // (one wait per time step): with the map in LDS the stores of the staged-states form overlap the loop
  int coef_given;       // no sampled parameters: the model coefficients were evaluated once on the host
  float coef_host[2];
  // Particle(deterministic=False, noise_std != 0), acceleration control (particle.py:144-148: every step of every rollout draws
  // acts = action + noise_std z, and the step cost sees the raw action): drawn inside the rollout loops (round 6) - a Philox stream
  // of its own per rollout; recorded draws (tests) and velocity control keep particle_general.hpp
  int ctrl_noise;
  float dyn_std[2];
  const float *theta;   // [N_total][D] base of the noise (theta, or a_mat for MultiDISCO's own sampling)
  const float *noise;   // eps or actions [S][N_total][D] (device), or nullptr for Philox
  const float *params;  // [M][P] raw samples or nullptr
  const float *mw;      // [M] unscented-transform weights (sigma-point rollouts) or nullptr: plain mean over m
  const float *a_seq;   // [D]
  float *a_mat;         // [N_total][D]
  float *costsT;        // [N_total][S]
  const float *costs_in; // [S][N_total] or nullptr: stage-wise mode (SVMPC.phi with a user log_p): skip the rollouts
  int costs_own;         // costs_in holds THIS sample's costs (a first pass of the same call rolled them out: rollout_states.hpp,
                         // skid.hpp): the likelihood record (logl, eta) is refreshed as after a one-pass sample
  float *grad_lik;      // [N_total][D]
  float *grad_pri;      // [N_total][D] (merge_prior)
  float *score;         // [N_total][D] (merge_prior)
  float *logl;          // [N_total]   likelihood.log_prob per particle
  float *eta;           // [N_total]   logsumexp_s(-c/temp)  (a_mix = softmax_n(eta), beta cancels)
  float *omegaT;        // [N_total][S] or nullptr
  float *actions_out;   // [S][N_total][D] or nullptr
  float *states_out;    // [M][S][N_total][H+1][ds] or nullptr
  float *tile_scratch;  // [n_local][S][D|1] HBM slab for action tiles too large for LDS, else nullptr
  uint64_t seed;
  unsigned long long *stamps;  // diagnostic build only
};

// In-launch hand-off from the prior-pass workgroups of a fused launch (fused.hpp): per query tile, a monotonic arrival
// counter; `target` arrivals mean every key slice of that tile has published its partials.
struct FusedWait {
  const unsigned int *cnt;  // [tiles][CNT_STRIDE]
  unsigned int target;
  unsigned int *timeout_flag;
  // one-launch SVGD iteration (fused.hpp svgd_iter_kernel): the score rows are published to the Stein tiles of the SAME launch -
  // written through (sc1), then one arrival per workgroup on the counter of the key slice its particles belong to
  unsigned int *score_cnt;  // [JS][CNT_STRIDE] or nullptr
  int score_slice;          // keys per slice (a multiple of the particles per workgroup)
  int score_add;            // particles per workgroup
  unsigned long long *tl;   // diagnostic build only: launch timeline words
  // score rows published AS DATA: written through into a buffer that holds a sentinel in every word until then; the Stein
  // tiles poll the rows themselves - no drain, no barrier, no counter between the last store and the consumer (or nullptr)
  float *score_pub;
};

// `tid`/`nt` are the lane index and lane count of the sub-block that owns local particle `nl`; barriers are workgroup
// wide (every sub-block of a workgroup runs the same control flow), reductions are sub-block local.
template <int MODEL, int NB /* staged noise loads in flight per lane: 32 standalone, 12 inside the fused launch (VGPR budget) */,
          bool GROUPS /* lane = (sample, dynamics group): a.G > 1 */,
          bool LEAN /* none of: stored states / actions / omega, injected costs, control cost, sigma-point weights, HBM tile */,
          bool STATES = false /* the stored-states form (HBM-bound): states_out set; no injected costs / sigma-point weights / HBM tile */>
__device__ __forceinline__ void rollout_body(const RolloutArgs &a, float *lds, const int tid, const int nt, const int nl,
                                             const FusedWait *fw) {
  constexpr int DS = MODEL == DUST_MODEL_PENDULUM ? 2 : 4;
  constexpr int DA = MODEL == DUST_MODEL_PENDULUM ? 1 : 2;
  const int n = a.n0 + nl;
  const int S = a.S, D = a.D, H = a.H, N = a.N_total;
  const int Dp = D | 1;
  // Optional features are compiled OUT of the LEAN instances (the host picks them when none is requested): the argument
  // block no longer has to stay live in SGPRs across the hot loops (the full kernel spills ~1000 scalars to VGPR lanes).
  float *const f_states = LEAN ? nullptr : a.states_out;
  float *const f_actions = LEAN ? nullptr : a.actions_out;
  const float *const f_costs_in = (LEAN || STATES) ? nullptr : a.costs_in;
  const float f_a_reg = LEAN ? 0.0f : a.a_reg;
  const float *const f_mw = (LEAN || STATES) ? nullptr : a.mw;
  float *const f_tile_scratch = (LEAN || STATES) ? nullptr : a.tile_scratch;
  float *const f_omegaT = LEAN ? nullptr : a.omegaT;
  // the S x D action tile lives in LDS; when it would not fit (huge S*D) it spills to a per-workgroup HBM scratch slab
  float *tile = f_tile_scratch ? f_tile_scratch + (size_t)nl * S * Dp : lds;  // [S][Dp] actions
  float *cst = f_tile_scratch ? lds : lds + (size_t)S * Dp;                            // [S] costs -> weights
  float *omg = cst + S;    // [S] omega
  float *red = omg + S;    // [96] reduction scratch, flags, prior slice words
  float *part = red + 96;  // [2][nt] partial sums for the weighted reductions

  DUST_STAMP(a.stamps, 0);
  // ---- 1. stage the action tile (a1: actions = theta + L eps) ----
  // theta row of this particle -> LDS once (every later use is an LDS broadcast / lane read, not a global load)
  float *th = part + 2 * nt + 2;  // [D]
  // red[40]: "some action of this particle is NaN" (the branch-free rollout loop clamps with v_med3_f32, which would
  // swallow a NaN that torch.clamp propagates: such particles take the general loop)
  if (tid == 0) red[40] = 0.f;
  float thv = 0.f, amv = 0.f;
  if (tid < D) thv = a.theta[(size_t)n * D + tid];
  if (tid < D && a.update_a_mat) amv = a.a_mat[(size_t)n * D + tid];  // consumed at the very end: no round trip there
  float x0[DS];
#pragma unroll
  for (int k = 0; k < DS; ++k) x0[k] = a.state[k];
  // per-dynamics-sample coefficients once per workgroup (the Python-float / fp32-tensor emulation is long and branchy)
  float *coefs = th + D;  // [M][2]
  for (int m = tid; m < a.M; m += nt) {
    if (a.coef_given) {
      coefs[2 * m] = a.coef_host[0];
      coefs[2 * m + 1] = a.coef_host[1];
    } else {
      const Coef c = make_coef(a.dm, a.params ? a.params + (size_t)m * a.dm.P : nullptr);
      coefs[2 * m] = c.c0;
      coefs[2 * m + 1] = c.c1;
    }
  }
  // caller-supplied noise, row-lane mapping: W = 2^lgW >= D lanes per action row, R = nt / W rows per batch, so a lane keeps
  // ONE column j (theta_j and the Cholesky factor are per-lane constants, no index division) and walks the rows with a
  // fixed stride in HBM and in LDS.  The first NB loads of every lane are issued BEFORE the wait on the theta row, so the
  // two round trips overlap (at S <= NB R - cfg2: 128 rows = 32 batches of 4 - that is the whole tile: one memory round
  // trip for stage 1).  Offsets are clamped, never predicated (a conditional load makes hipcc branch and wait per element).
  // occupancy grid -> LDS (Particle; 6 KB for the demo map), right behind the coefficients
  // (offsets are formed in the index domain: a round trip through an integer would lose the LDS address space and turn every
  // access below into a flat load that waits on vmcnt AND lgkmcnt)
  const int off_grid = ((int)((coefs + 2 * a.M) - lds) + 3) & ~3;
  uint32_t *gridl = reinterpret_cast<uint32_t *>(lds + off_grid);
  // (compile-time in the LEAN and STATES instances - the host gives them the map in LDS whenever there is one: a run-time choice
  // between an LDS and an HBM pointer would make every lookup a flat load, which waits on vmcnt AND lgkmcnt)
  constexpr bool GRID_LDS = (LEAN || STATES) && MODEL == DUST_MODEL_PARTICLE;
  DevModel dml = a.dm;
  if (GRID_LDS) {
    const int words = a.dm.with_obstacle ? (a.dm.nx * a.dm.ny + 31) >> 5 : 0;
    for (int w = tid; w < words; w += nt) gridl[w] = a.dm.grid_bits[w];
    dml.grid_bits = gridl;
  }
  float v[NB];
  const int lgW = a.lgW, R = nt >> lgW;
  const int sr = tid >> lgW, sj = tid & ((1 << lgW) - 1);
  const bool jv = sj < D;
  const int esh = a.noise_f16 ? 1 : 2;  // log2 of the element size in HBM
  const char *nbase = reinterpret_cast<const char *>(a.noise) + (((size_t)n * D) << esh);
  const uint32_t rowbytes = ((uint32_t)N * (uint32_t)D) << esh;  // S*N*D < 2^30 is checked at configuration time
  const uint32_t jbytes = (uint32_t)min(sj, D - 1) << esh;
  const uint32_t off_last = (uint32_t)(S - 1) * rowbytes + jbytes, off_step = (uint32_t)R * rowbytes;
  uint32_t off0 = (uint32_t)sr * rowbytes + jbytes;
  // (the element-size test is wave-uniform and sits OUTSIDE the unrolled load sequences)
  auto issue_loads = [&]() {
    if (a.noise_f16) {
#pragma unroll
      for (int u = 0; u < NB; ++u) v[u] = (float)*reinterpret_cast<const _Float16 *>(nbase + min(off0 + (uint32_t)u * off_step, off_last));
    } else {
#pragma unroll
      for (int u = 0; u < NB; ++u) v[u] = *reinterpret_cast<const float *>(nbase + min(off0 + (uint32_t)u * off_step, off_last));
    }
  };
  if (a.noise_mode != NOISE_PHILOX) issue_loads();
  if (tid < D) th[tid] = thv;
  wg_sync();
  bool nanf = tid < D && thv != thv;
  if (a.noise_mode != NOISE_PHILOX) {
    // (full a_cov: the raw draws are staged and turned into actions by the pass below - a row of L eps needs two of them)
    const bool eps_diag = a.noise_mode == NOISE_EPS && !(DA == 2 && a.full_cov);
    const float thj = eps_diag ? th[min(sj, D - 1)] : 0.f;
    const float lj = eps_diag ? pick_da<DA>(a.chol_a, sj) : 1.f;
    for (int s0 = 0; s0 < S; s0 += NB * R) {
      if (s0) {
        off0 += (uint32_t)NB * off_step;
        issue_loads();
      }
      float *trow = tile + (s0 + sr) * Dp + sj;
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int srow = s0 + u * R + sr;
        if (srow < S && jv) {
          const float av = thj + lj * v[u];  // NOISE_ACTIONS: 0 + 1 * v (exact)
          trow[u * R * Dp] = av;
          nanf |= av != av;
        }
      }
    }
  }
  if (DA == 2 && a.full_cov && a.noise_mode == NOISE_EPS) {
    // actions = theta + L eps, L lower triangular (MultivariateNormal.rsample: loc + scale_tril @ eps, likelihoods.py:85-90)
    wg_sync();
    for (int idx = tid; idx < S * (D >> 1); idx += nt) {
      const int s = idx / (D >> 1), t = idx - s * (D >> 1);
      float *p = tile + s * Dp + 2 * t;
      const float e0 = p[0], e1 = p[1];
      const float a0 = th[2 * t] + a.chol_a[0] * e0;
      const float a1 = th[2 * t + 1] + (a.chol_off * e0 + a.chol_a[1] * e1);
      p[0] = a0;
      p[1] = a1;
      nanf |= (a0 != a0) || (a1 != a1);
    }
  }
  if (nanf) red[40] = 1.f;
  wg_sync();

  DUST_STAMP(a.stamps, 1);
  // ---- 2. rollouts: lane = sample s, dynamics samples m looped in registers (a2-a5) ----
  const long SN = (long)S * N;
  const uint32_t ctr_tick = a.ctr[0], ctr_iter = a.ctr[1];
  // |theta| stays below |theta_0| + max_speed dt H: wave-uniform test for the branch-free trig path
  const bool fast_part = MODEL == DUST_MODEL_PARTICLE && red[40] == 0.f && fabsf(x0[0]) <= 3.0e38f && fabsf(x0[1]) <= 3.0e38f &&
                         fabsf(x0[DS > 2 ? 2 : 0]) <= 3.0e38f && fabsf(x0[DS > 3 ? 3 : 0]) <= 3.0e38f &&
                         // reachable positions keep floor(p / cell + off) far from the int64 edge (collision_pair)
                         (fabsf(x0[0]) + fabsf(x0[1]) + (fabsf(x0[DS > 2 ? 2 : 0]) + fabsf(x0[DS > 3 ? 3 : 0]) + a.dm.max_speed * (float)H) * fabsf((float)a.dm.dt)) *
                                     fabsf(a.dm.inv_cell) + fabsf(a.dm.off_x) + fabsf(a.dm.off_y) < 1.0e17f;
  const bool fast_trig = MODEL == DUST_MODEL_PENDULUM && !f_states && !f_tile_scratch && red[40] == 0.f && !f_mw &&
                         fabsf(x0[1]) <= 3.0e38f && (fabsf(x0[0]) + a.dm.max_speed_pend * (float)a.dm.dt * (float)H < 5.0e4f);
  // lane = (sample s, dynamics group mg): with several dynamics samples per rollout (M > 1) and few action samples the
  // M loop is split over G lane groups that share the action tile - 4x the waves per LDS byte at cfg3 (S = 64, M = 64)
  // (GROUPS is a template parameter - separate kernels - so that the common G == 1 form keeps mg = 0 / G = 1 as constants:
  // block indices, the dynamics-sample index and most of the Philox rounds stay wave-uniform, and the argument block is not
  // kept live across two copies of the loop: with both in one kernel the SGPR spill traffic cost 3.6 us per launch)
  constexpr bool ONE = !GROUPS;
  const int G = ONE ? 1 : a.G, sub = ONE ? nt : nt / a.G;
  const int mg = ONE ? 0 : __builtin_amdgcn_readfirstlane(tid / sub);  // sub is a multiple of 64: a wave belongs to one group
  const int ts = ONE ? tid : tid - mg * sub;
  if (a.noise_mode == NOISE_PHILOX) {  // the lanes of a sample fill ITS row (8-element blocks dealt over the groups)
    for (int s = ts; s < S; s += sub) {
      float *act = tile + s * Dp;
      for (int j8 = mg; j8 * 8 < D; j8 += G) {
        float z[8];
        philox_normal8(a.seed, (uint32_t)j8, (uint32_t)(s * N + n), ctr_iter, ctr_tick, z);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int j = j8 * 8 + q;
          if (DA == 2 && a.full_cov && (q & 1)) {  // (an odd column's partner draw sits in the same block)
            if (j < D) act[j] = th[j] + (a.chol_off * z[q - 1] + a.chol_a[1] * z[q]);
          } else if (j < D) act[j] = th[j] + pick_da<DA>(a.chol_a, j) * z[q];
        }
      }
    }
    if (G > 1) wg_sync();  // G == 1: every lane consumes only the row it filled itself
  }
  // [G][sub] per-group partial sums over m (G > 1); `part` is only 4-byte aligned for odd S: round up (one spare pair of
  // floats is reserved in rollout_lds_bytes)
  double *accp = reinterpret_cast<double *>((reinterpret_cast<uintptr_t>(part) + 7) & ~(uintptr_t)7);
  auto finish_cost = [&](const int s, const double acc, const bool is_sum = false) {
    float cost = (a.M == 1 || is_sum) ? (float)acc : (float)(acc / a.M);
    if (f_a_reg != 0.0f) {  // disco.py:338-346, diagonal of the [S,N,N] tensordot only
      const float *act = tile + s * Dp;
      double cc = 0.0;
      for (int j = 0; j < D; ++j) {
        const float e = act[j] - a.a_seq[j];
        float ap = a.a_mat[(size_t)n * D + j] * pick_da<DA>(a.a_pre, j);
        if (DA == 2 && a.full_cov) {  // (a_mat[n, t, :] @ a_pre)[d], a_pre a full symmetric 2 x 2 (disco.py:341-344)
          const float m0 = a.a_mat[(size_t)n * D + (j & ~1)], m1 = a.a_mat[(size_t)n * D + (j | 1)];
          ap = (j & 1) ? (m0 * a.a_pre_off + m1 * a.a_pre[1]) : (m0 * a.a_pre[0] + m1 * a.a_pre_off);
        }
        cc += (double)(-e) * (double)ap;
      }
      cost = cost + f_a_reg * (float)cc;
    }
    cst[s] = cost;
    a.costsT[(size_t)n * S + s] = cost;
  };
  // Stored states (MultiDISCO.forward returns them, disco.py:394; 11 GB at cfg3): the HBM-bound form of this kernel.  A lane's
  // trajectory is one contiguous (H + 1) * ds run, but the lanes of a wave are N rows apart, so per-lane stores are 4-16-byte
  // scatters (measured 0.43 TB/s).  Instead every wave stages its 64 trajectories in LDS in 128-byte chunks (chunk = 128 /
  // bytes-per-state steps) and writes a chunk out cooperatively: 8 consecutive lanes store the 8 16-byte pieces of ONE
  // trajectory's chunk - full 128-byte runs - 8 trajectories per store instruction.  Whole waves take part (lanes past S roll a
  // clamped duplicate out and store nothing).
  const int bps = DS * (a.store_f16 ? 2 : 4);  // bytes per stored state
  // staged bytes per lane and flush: one 128-byte line; Particle: half a line for each of the pair path's two trajectories (the
  // halves of a line are written 4 steps apart and merge in L2), which keeps the staging at 160 B per lane
  constexpr int CHB = MODEL == DUST_MODEL_PARTICLE ? DUST_STG_CHUNK_PART : DUST_STG_CHUNK;
  constexpr int STG_ROW = CHB + 16;            // LDS bytes per lane (+16: bank spread)
  constexpr int PC = CHB / 16;                 // 16-byte pieces per chunk = lanes that share one trajectory's chunk
  char *stg = nullptr;
  if (f_states) {
    // (Particle: two staging areas per wave - the pair path rolls two trajectories per lane)
    stg = reinterpret_cast<char *>(gridl + (GRID_LDS ? a.grid_words : 0)) + (size_t)(tid >> 6) * (MODEL == DUST_MODEL_PARTICLE ? 2 : 1) * 64 * STG_ROW;
  }
  const int lane64 = tid & 63;
  const int S_loop = f_states ? min(sub, (S + 63) & ~63) : S;
  for (int s_raw = ts; s_raw < S_loop; s_raw += sub) {
    const bool live = s_raw < S;
    const int s = live ? s_raw : S - 1;
    float *act = tile + s * Dp;
    if (f_costs_in) {
      if (mg == 0 && live) cst[s] = f_costs_in[(size_t)s * N + n];
      continue;
    }
    double acc_m = 0.0, ut_term = 0.0;
    int m_begin = mg;
    // ---- stored states: staging row write / cooperative chunk flush, for the trajectory block whose first row is r_first
    const size_t rowb = (size_t)(H + 1) * bps;
    const int n_traj = min(64, S - (s_raw - lane64));  // valid trajectories of this wave
    auto put_state_at = [&](char *sg, const int ph, const int row, const float *xs) {  // this lane's state -> its staging row
      char *dst = sg + lane64 * STG_ROW + ((row * bps + ph) & (CHB - 1));
      if (a.store_f16) {
        _Float16 hx[DS];
#pragma unroll
        for (int k = 0; k < DS; ++k) hx[k] = (_Float16)xs[k];
        if (DS == 2) *reinterpret_cast<uint32_t *>(dst) = *reinterpret_cast<const uint32_t *>(hx);
        else {
          reinterpret_cast<uint32_t *>(dst)[0] = reinterpret_cast<const uint32_t *>(hx)[0];
          reinterpret_cast<uint32_t *>(dst)[1] = reinterpret_cast<const uint32_t *>(hx)[1];
        }
      } else {
#pragma unroll
        for (int k = 0; k < DS; ++k) reinterpret_cast<float *>(dst)[k] = xs[k];  // (4-byte aligned in general; merged when the phase allows)
      }
    };
    auto flush_at = [&](const char *sg, const long r_first, const int ph, const int chunk, const int lo, const int hi) {  // staged bytes [lo, hi) of the wave's chunk -> HBM
      char *gbase = reinterpret_cast<char *>(f_states) + (size_t)chunk * CHB - ph;
      const int piece = lane64 & (PC - 1);
      v4f pv[PC];
#pragma unroll
      for (int i = 0; i < PC; ++i) pv[i] = *reinterpret_cast<const v4f *>(sg + ((lane64 / PC) + (64 / PC) * i) * STG_ROW + piece * 16);  // all reads in flight
      const bool pok = piece * 16 >= lo && piece * 16 + 16 <= hi;
#pragma unroll
      for (int i = 0; i < PC; ++i) {
        const int j = (lane64 / PC) + (64 / PC) * i;
        if (j < n_traj && pok) {
#ifdef DUST_STATES_NT
          __builtin_nontemporal_store(pv[i], reinterpret_cast<v4f *>(gbase + (size_t)(r_first + (long)j * N) * rowb + piece * 16));
#else
          *reinterpret_cast<v4f *>(gbase + (size_t)(r_first + (long)j * N) * rowb + piece * 16) = pv[i];
#endif
        }
      }
      // words of partially covered 16-byte pieces at either end (at most 3 + 3): lane j copies trajectory j's
      const int head_end = min(hi, (lo + 15) & ~15), tail_beg = max(head_end, hi & ~15);
      if (lane64 < n_traj) {
        char *grow = gbase + (size_t)(r_first + (long)lane64 * N) * rowb;
        const char *srow = sg + lane64 * STG_ROW;
        for (int o4 = lo; o4 < head_end; o4 += 4) *reinterpret_cast<float *>(grow + o4) = *reinterpret_cast<const float *>(srow + o4);
        for (int o4 = tail_beg; o4 < hi; o4 += 4) *reinterpret_cast<float *>(grow + o4) = *reinterpret_cast<const float *>(srow + o4);
      }
    };
    if (MODEL == DUST_MODEL_PARTICLE && (LEAN || STATES) && fast_part) {
      // Particle, finite operands, nothing stored: TWO dynamics samples (m, m + G) of this lane's action row roll out side by
      // side - two independent dependency chains per lane, interleaved by the scheduler (the loop is latency-bound: ~150
      // dependent instructions per step at 3-5 waves per SIMD); accumulation order over m is unchanged (m, then m + G, ...)
      const float *actl = lds + s * Dp;
      int m = mg;
      for (; m + G < a.M; m += 2 * G) {
        const int pa = a.dm.interleave ? (int)(((long)m * SN + (long)s * N + n) % a.M) : m;
        const int pb = a.dm.interleave ? (int)(((long)(m + G) * SN + (long)s * N + n) % a.M) : m + G;
        const float ma = coefs[2 * pa], mb = coefs[2 * pb];
        if (!(fabsf(ma) >= 1.0e-30f && fabsf(ma) <= 1.0e30f && fabsf(mb) >= 1.0e-30f && fabsf(mb) <= 1.0e30f)) break;  // general loop
        const float ra = 1.0f / ma, rb = 1.0f / mb;
        const v2f m2 = {ma, mb}, r2 = {ra, rb};
        v2f xp[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) xp[k] = (v2f){x0[k], x0[k]};
        v2f ksum = {0.f, 0.f}, kcomp = {0.f, 0.f};  // PairKahan's arithmetic (plain locals: the struct, captured by the lambda below, kept a dead stack slot alive)
        float tca, tcb;
        // stored states: the two trajectories (rows G S N apart: the same alignment phase) stage side by side and their chunks
        // flush together - 2 PC LDS reads, then 2 PC stores in flight per flush
        const long rA = (long)m * SN + (long)(s_raw - lane64) * N + n, rB = rA + (long)G * SN;
        const int ph = (STATES && ((size_t)N * rowb) % CHB == 0) ? (int)(((size_t)rA * rowb) % CHB) : 0;
        char *stgB = stg + 64 * STG_ROW;
        auto put_pair = [&](const int row) {
          if (!STATES) return;
          const float sa[4] = {xp[0].x, xp[1].x, xp[2].x, xp[3].x}, sb[4] = {xp[0].y, xp[1].y, xp[2].y, xp[3].y};
          put_state_at(stg, ph, row, sa);
          put_state_at(stgB, ph, row, sb);
          const int endb = (row + 1) * bps + ph;  // staged bytes so far, counted from the start of chunk 0's line
          if ((endb & (CHB - 1)) == 0) {
            flush_at(stg, rA, ph, endb / CHB - 1, endb == CHB ? ph : 0, CHB);
            flush_at(stgB, rB, ph, endb / CHB - 1, endb == CHB ? ph : 0, CHB);
          } else if (row == H) {
            flush_at(stg, rA, ph, endb / CHB, endb < CHB ? ph : 0, endb & (CHB - 1));
            flush_at(stgB, rB, ph, endb / CHB, endb < CHB ? ph : 0, endb & (CHB - 1));
          }
        };
        PairK pk;
        pk.load(a.dm);
        pk.pin();  // (VGPR pairs: with the packed cost sums the compiler runs out of SGPR pairs and parks odd-indexed splats in a stack slot)
        auto pair_loop = [&](auto obst, auto crash, auto noisy) {
          constexpr bool OB = decltype(obst)::value, CR = decltype(crash)::value, NZ = decltype(noisy)::value;
          put_pair(0);
          if constexpr (NZ) {
            // control noise: ONE block of eight normals serves two steps of the lane's two rollouts (two channels each); the block is
            // keyed by the first rollout of the pair (key word "ctrp"), the step pair, the iteration and the tick
            const long rN = (long)m * SN + (long)s * N + n;
            const v2f sd0 = {a.dyn_std[0], a.dyn_std[0]}, sd1 = {a.dyn_std[1], a.dyn_std[1]};
            for (int t = 0; t < H; t += 2) {
              float z[8];
              philox_normal8(a.seed ^ 0x6374727000000000ull, (uint32_t)rN, (uint32_t)((unsigned long long)rN >> 32) ^ ((uint32_t)(t >> 1) << 8), ctr_iter, ctr_tick, z);
#pragma unroll
              for (int q = 0; q < 2; ++q) {
                if (t + q < H) {
                  const float a0 = actl[2 * (t + q)], a1 = actl[2 * (t + q) + 1];
                  const v2f un[2] = {v2f{a0, a0} + sd0 * v2f{z[4 * q], z[4 * q + 2]}, v2f{a1, a1} + sd1 * v2f{z[4 * q + 1], z[4 * q + 3]}};
                  const v2f c = particle_pair_step<OB, CR>(a.dm, pk, dml.grid_bits, m2, r2, xp, a0, a1, particle_ctrl_cost(a.dm, a0, a1), nullptr, un);
                  const v2f ky = c - kcomp;
                  const v2f kt = ksum + ky;
                  kcomp = (kt - ksum) - ky;
                  ksum = kt;
                  put_pair(t + q + 1);
                }
              }
            }
          } else {
#pragma unroll 2  // (two steps per trip: the loop-carried register copies of a single-step body - 4 v_mov of 86 instructions - go away)
            for (int t = 0; t < H; ++t) {
              const float a0 = actl[2 * t], a1 = actl[2 * t + 1];
              const v2f c = particle_pair_step<OB, CR>(a.dm, pk, dml.grid_bits, m2, r2, xp, a0, a1, particle_ctrl_cost(a.dm, a0, a1));
              {
                const v2f ky = c - kcomp;
                const v2f kt = ksum + ky;
                kcomp = (kt - ksum) - ky;
                ksum = kt;
              }
              put_pair(t + 1);
            }
          }
          const v2f tc = particle_pair_term<OB>(a.dm, dml.grid_bits, xp);
          tca = tc.x;
          tcb = tc.y;
        };
        if (a.ctrl_noise) {
          if (!dml.with_obstacle) pair_loop(std::false_type{}, std::false_type{}, std::true_type{});
          else if (dml.can_crash) pair_loop(std::true_type{}, std::true_type{}, std::true_type{});
          else pair_loop(std::true_type{}, std::false_type{}, std::true_type{});
        } else if (!dml.with_obstacle) pair_loop(std::false_type{}, std::false_type{}, std::false_type{});
        else if (dml.can_crash) pair_loop(std::true_type{}, std::true_type{}, std::false_type{});
        else pair_loop(std::true_type{}, std::false_type{}, std::false_type{});
        acc_m += (double)(ksum.x + tca);
        acc_m += (double)(ksum.y + tcb);
      }
      m_begin = m;
    }
    if (MODEL == DUST_MODEL_PENDULUM && fast_trig) {
      // Round 6: TWO of the lane's dynamics samples side by side in packed fp32 (cfg5: M = 8) - the step is nearly all FMAs, and a
      // packed FMA is one lane-op for two of them: 45 -> ~30 instructions per sample and step.  The same operations in the same order
      // as the one-sample loop below, and the samples' trajectory costs join acc_m in its order: bit-identical results.
      const float dt = (float)a.dm.dt, mt = a.dm.max_torque, ms = a.dm.max_speed_pend;
      const float *actl = lds + s * Dp;
      int m = m_begin;
      for (; m + G < a.M; m += 2 * G) {
        const long rA = (long)m * SN + (long)s * N + n, rB = rA + (long)G * SN;
        const int pa = a.dm.interleave ? (int)(rA % a.M) : m, pb = a.dm.interleave ? (int)(rB % a.M) : m + G;
        const v2f c0 = {coefs[2 * pa], coefs[2 * pb]}, c1 = {coefs[2 * pa + 1], coefs[2 * pb + 1]};
        if (!(fabsf(c0.x) <= 3.0e38f && fabsf(c0.y) <= 3.0e38f && fabsf(c1.x) <= 3.0e38f && fabsf(c1.y) <= 3.0e38f)) break;  // (the general loop)
        v2f th = {x0[0], x0[0]}, thd = {x0[1], x0[1]};
        const v2f wc = {a.dm.w_cos, a.dm.w_cos}, wv = {a.dm.w_vel, a.dm.w_vel}, dt2 = {dt, dt}, one = {1.0f, 1.0f};
        double totA = 0.0, totB = 0.0;
        v2f part = {0.f, 0.f};
        v2f sn, cs;
#pragma unroll 4
        for (int t = 0; t < H; ++t) {
          pendulum_trig2(th, &sn, &cs);
          const v2f cm = cs - one;
          part += wc * (cm * cm) + wv * (thd * thd);  // (q = W (q q); q.x + q.y of the one-sample loop, per sample)
          if ((t & 3) == 3) {  // CostSum<PENDULUM>: groups of four steps, then double
            totA += (double)part.x;
            totB += (double)part.y;
            part = v2f{0.f, 0.f};
          }
          const float u = __builtin_amdgcn_fmed3f(actl[t], -mt, mt);
          thd = thd + dt2 * (c0 * sn + c1 * v2f{u, u});
          thd.x = __builtin_amdgcn_fmed3f(thd.x, -ms, ms);
          thd.y = __builtin_amdgcn_fmed3f(thd.y, -ms, ms);
          th = th + thd * dt2;
        }
        pendulum_trig2(th, &sn, &cs);
        const v2f cm = cs - one;
        const v2f tq = wc * (cm * cm) + wv * (thd * thd);
        acc_m += (double)((float)(totA + (double)part.x) + tq.x);
        acc_m += (double)((float)(totB + (double)part.y) + tq.y);
      }
      m_begin = m;
    }
    for (int m = m_begin; m < a.M; m += G) {
      const long r = (long)m * SN + (long)s * N + n;
      const int pidx = a.dm.interleave ? (int)(r % a.M) : m;
      Coef cf;
      cf.c0 = coefs[2 * pidx];
      cf.c1 = coefs[2 * pidx + 1];
      float x[DS];
#pragma unroll
      for (int k = 0; k < DS; ++k) x[k] = x0[k];
      CostSum<MODEL> tot;
      float traj;
      if (fast_trig && fabsf(cf.c0) <= 3.0e38f && fabsf(cf.c1) <= 3.0e38f) {
        // PendulumModel.step + demo cost, same fp32 operation order as model_step / inst_cost.  All operands are finite
        // here, so the clamps are single v_med3_f32; the action row is addressed as LDS (ds_read, not flat).
        const float dt = (float)a.dm.dt, mt = a.dm.max_torque, ms = a.dm.max_speed_pend;
        const v2f W = {a.dm.w_cos, a.dm.w_vel};
        const float *actl = lds + s * Dp;
        float sn, cs;
#pragma unroll 4
        for (int t = 0; t < H; ++t) {
          pendulum_trig(x[0], &sn, &cs);
          v2f q = {cs - 1.0f, x[1]};
          q = W * (q * q);
          tot.add(q.x + q.y, t);
          const float u = __builtin_amdgcn_fmed3f(actl[t], -mt, mt);
          float thd = x[1] + dt * (cf.c0 * sn + cf.c1 * u);
          thd = __builtin_amdgcn_fmed3f(thd, -ms, ms);
          x[0] = x[0] + thd * dt;
          x[1] = thd;
        }
        pendulum_trig(x[0], &sn, &cs);
        v2f q = {cs - 1.0f, x[1]};
        q = W * (q * q);
        traj = (float)tot.total() + (q.x + q.y);
      } else {
        const bool so = f_states != nullptr;
        // trajectory j of this wave (lane j's) is row r_first + j N of the [M][S][N] rollout index
        const long r_first = ((long)m * SN + (long)(s_raw - lane64) * N + n);
        // chunk boundaries sit on 128-byte LINES of the output when every trajectory of the wave has the same alignment (N rows
        // apart: N * rowb a multiple of 128 - cfg3: 4096 * 656): a store instruction then writes whole lines, and only a
        // trajectory's first / last partial chunk shares its line with the neighbouring row
        const int ph = ((size_t)N * rowb) % CHB == 0 ? (int)(((size_t)r_first * rowb) % CHB) : 0;  // bytes; a multiple of 4
        auto put_state = [&](const int row) {  // this lane's state -> its staging row
          put_state_at(stg, ph, row, x);
        };
        auto flush = [&](const int chunk, const int lo, const int hi) { flush_at(stg, r_first, ph, chunk, lo, hi); };
        auto put_and_flush = [&](const int row) {
          put_state(row);
          const int endb = (row + 1) * bps + ph;  // staged bytes so far, counted from the start of chunk 0's line
          if ((endb & (CHB - 1)) == 0) flush(endb / CHB - 1, endb == CHB ? ph : 0, CHB);
          else if (row == H) flush(endb / CHB, endb < CHB ? ph : 0, endb & (CHB - 1));
        };
        if (so) put_and_flush(0);
        float zc0 = 0.f, zc1 = 0.f, zc2 = 0.f, zc3 = 0.f;  // control noise of this rollout (key word "ctrd": one block of four normals per two steps;
                                                            // four scalars - an array read with (t & 1) became a scratch slot with a dynamic index)
        for (int t = 0; t < H; ++t) {
          float at[DA];
#pragma unroll
          for (int k = 0; k < DA; ++k) at[k] = act[t * DA + k];
          float ci;
          if (MODEL == DUST_MODEL_PARTICLE && a.ctrl_noise) {
            if ((t & 1) == 0) {
              float zn[4];
              philox_normal4(a.seed ^ 0x6374726400000000ull, (uint32_t)r, (uint32_t)((unsigned long long)r >> 32) ^ ((uint32_t)(t >> 1) << 8), ctr_iter, ctr_tick, zn);
              zc0 = zn[0];
              zc1 = zn[1];
              zc2 = zn[2];
              zc3 = zn[3];
            }
            const float zq[2] = {(t & 1) ? zc2 : zc0, (t & 1) ? zc3 : zc1};
            float un[DA];
#pragma unroll
            for (int k = 0; k < DA; ++k) un[k] = at[k] + (k < 2 ? a.dyn_std[k & 1] * zq[k & 1] : 0.f);
            ci = step_with_cost<MODEL>(dml, cf, x, at, un);
          } else
          ci = step_with_cost<MODEL>(dml, cf, x, at);  // cost of the state BEFORE the action (disco.py:306)
          // sigma-point rollouts: the reference pairs entry (m, t) of its flat [sigma][step] block with w[(m H + t) mod M]
          if (f_mw) tot.add_weighted((double)f_mw[((long)m * H + t) % a.M], ci);
          else tot.add(ci, t);
          if (so) put_and_flush(t + 1);
        }
        if (f_mw) {  // weighted instantaneous and terminal parts are summed separately over the sigma points (disco.py:314-321)
          ut_term += (double)f_mw[m] * (double)term_cost<MODEL>(dml, x);
          traj = (float)tot.total();  // unused
          acc_m += tot.total();
          continue;
        }
        traj = (float)tot.total() + term_cost<MODEL>(dml, x);
      }
      acc_m += (double)traj;
    }
    if (G > 1) {  // every lane has at most one sample here: park the group partial, combine after the barrier below
      accp[mg * sub + ts] = acc_m;
      continue;
    }
    if (!live) continue;
    if (f_mw) finish_cost(s, (double)((float)acc_m + (float)ut_term), true);  // weighted sum over sigma points, not a mean
    else finish_cost(s, acc_m);
  }
  if (G > 1) {  // fixed-order sum of the group partials (the barrier sits outside the sample loop: lanes without a sample reach it too)
    wg_sync();
    if (mg == 0 && ts < S && !f_costs_in) {
      double acc_m = accp[ts];
      for (int g = 1; g < G; ++g) acc_m += accp[g * sub + ts];
      finish_cost(ts, acc_m);
    }
  }
  wg_sync();
  if (f_actions) {
    for (int idx = tid; idx < S * D; idx += nt) {
      const int s = (int)__umulhi((uint32_t)idx, a.magicD), j = idx - s * D;
      if (a.store_f16) reinterpret_cast<_Float16 *>(f_actions)[((size_t)s * N + n) * D + j] = (_Float16)tile[s * Dp + j];
      else f_actions[((size_t)s * N + n) * D + j] = tile[s * Dp + j];
    }
  }

  DUST_TL(fw ? fw->tl : nullptr, 1);
  if (fw) {
    // fused launch: the prior-pass workgroups of this launch publish the partials of query tile nl/32 with an agent-scope
    // release + counter add; poll relaxed from ONE lane, acquire once, then the barrier admits the other lanes
    // (cdna_hip_programming.md Guideline 16, counter form).  The spin is bounded.
    if (threadIdx.x == 0) {
      const unsigned int *cp = fw->cnt + (nl / PAIR_TI) * CNT_STRIDE;
      unsigned int spins = 0;
      while (__hip_atomic_load(cp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < fw->target) {
        __builtin_amdgcn_s_sleep(4);
        if (++spins > (1u << 24)) {
          *fw->timeout_flag = 1u;
          break;
        }
      }
    }
    wg_sync();  // the partials below are read with sc1 loads only (they were stored sc1): no acquire fence needed
  }
  DUST_STAMP(a.stamps, 2);
  // ---- 3. softmax over samples: likelihood weights w (alpha) and MPPI weights omega (1/temp) ----
  float cmin = INFINITY, csum = 0.f;
  for (int s = tid; s < S; s += nt) {
    cmin = fminf(cmin, cst[s]);
    csum += cst[s];
  }
  {
    const int lane = tid & 63, wid = tid >> 6, nw = (nt + 63) >> 6;
    cmin = wave_min(cmin);
    csum = wave_sum(csum);
    if (lane == 0) {
      red[wid] = cmin;
      red[8 + wid] = csum;
    }
    lds_barrier();
    cmin = red[0];
    csum = red[8];
    for (int w = 1; w < nw; ++w) {
      cmin = fminf(cmin, red[w]);
      csum += red[8 + w];
    }
  }
  // un-normalised weights stay in LDS; the 1/Z factors are applied once to the reduced sums in stage 4.  With
  // temperature = 1/alpha (every demo) omega and w are the same softmax: one exp and one accumulation are skipped.
  const bool same_w = (a.alpha * a.temp == 1.0f);
  float zw = 0.f, zo = 0.f;
  for (int s = tid; s < S; s += nt) {
    const float c = cst[s];
    const float ew = expf(-c * a.alpha - (-cmin * a.alpha));  // svmpc.py:51 softmax(-costs * alpha)
    cst[s] = ew;
    zw += ew;
    if (!same_w) {
      const float lo = (-1.0f * (c - cmin)) / a.temp;  // disco.py:381 with beta := per-policy min (beta cancels in omega)
      const float eo = expf(lo);
      omg[s] = eo;
      zo += eo;
    }
  }
  {
    const int lane = tid & 63, wid = tid >> 6, nw = (nt + 63) >> 6;
    zw = wave_sum(zw);
    zo = wave_sum(zo);
    if (lane == 0) {
      red[16 + wid] = zw;
      red[24 + wid] = zo;
    }
    lds_barrier();  // also publishes every lane's cst[] / omg[] to the stage-4 readers
    zw = red[16];
    zo = red[24];
    for (int w = 1; w < nw; ++w) {
      zw += red[16 + w];
      zo += red[24 + w];
    }
    if (same_w) zo = zw;
  }
  const float *wom = same_w ? cst : omg;
  if (f_omegaT)
    for (int s = tid; s < S; s += nt) f_omegaT[(size_t)n * S + s] = wom[s] / zo;
  if (tid == 0 && (!f_costs_in || a.costs_own)) {
    if (a.lik == DUST_LIK_EXP_UTILITY)  // likelihoods.py:127-135
      a.logl[n] = ((-cmin * a.alpha) + logf(zw)) - logf((float)S);
    else  // likelihoods.py:113-119
      a.logl[n] = -a.alpha * (csum / (float)S);
    a.eta[n] = (-cmin / a.temp) + logf(zo);
  }

  DUST_STAMP(a.stamps, 3);
  if (a.bump_adam && nl == 0 && tid == 0) {
    // one-launch iteration: the update role of the same launch reads adam_step (sc1 load) - a device-scope RMW, issued HERE,
    // ahead of the prior-partial loads below: vector-memory operations retire in order, so the wait on those loads (which the
    // score needs) also covers this add, and the score rows - the update role's only way to learn that the rollouts are done -
    // cannot become visible before it
    if (fw && (fw->score_cnt || fw->score_pub)) __hip_atomic_fetch_add(a.ctr + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else a.ctr[2] += 1u;
  }
  // prior partials of this row: the (independent) loads are issued HERE - after the softmax, whose wave reductions would
  // otherwise run into waits on them (in-order vmcnt + register reuse) - and land under the weighted reductions below.
  // (the per-slice max / mass words are the same for every column: lanes 0..15 fetch one slice each and hand the combine
  // weights to the other lanes through LDS (red[64..95]) - 2 registers instead of 32 in every lane; slices past JS re-read
  // slice 0, clamped and never predicated, and are masked at the point of use: a select right after the load would make
  // the compiler wait for it on the spot)
  float pmA[16], pmM1 = -INFINITY, pmL1 = 0.f;
  const bool merger = a.merge_prior && tid < D;
  const int JS = a.pm.JS;
  if (a.merge_prior && tid < 16) {
    const size_t rowi = (size_t)min(tid, JS - 1) * a.pm.n_local + nl;
    if (fw) {
      pmM1 = __hip_atomic_load(a.pm.pM + rowi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      pmL1 = __hip_atomic_load(a.pm.pL + rowi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      pmM1 = a.pm.pM[rowi];
      pmL1 = a.pm.pL[rowi];
    }
  }
  if (merger) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const size_t rowi = (size_t)(u < JS ? u : 0) * a.pm.n_local + nl;  // combine weight of slices past JS is 0
      if (fw) pmA[u] = __hip_atomic_load(a.pm.pA + rowi * a.pm.ldp + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else pmA[u] = a.pm.pA[rowi * a.pm.ldp + tid];
    }
  }
  // ---- 4. weighted reductions over s: grad_lik (svmpc.py:52-54) and a_mat += sum_s omega eps (disco.py:387-392) ----
  const int Q = nt / D > 0 ? nt / D : 1;
  float g = 0.f, am = 0.f;
  const int q = (int)__umulhi((uint32_t)tid, a.magicD), j = tid - q * D;
  if (q < Q) {
    const float thj = th[j];
    const float is2 = 1.0f / (pick_da<DA>(a.sigma_a, j) * pick_da<DA>(a.sigma_a, j));  // (a - x) / sigma^2 as a multiply: <= 1 ulp apart
    // running pointers and a host-computed, wave-uniform trip count (a.wq_iters = ceil(S / Q)): no index division, no
    // remainder loop; the mode tests are hoisted out of the loop
    // (rows past S are clamped to row S-1 and given weight 0: a PREDICATED LDS load makes hipcc branch and wait per element -
    // this loop ran at 176 cycles per iteration that way)
    const float *tp = tile + j;
    int so = q, to = q * Dp;  // running row index / tile offset
    const int tstep = Q * Dp, tmax = (S - 1) * Dp;
    if (same_w && a.eps_base_mode) {  // SVMPC: omega == w and eps = a - theta: the a_mat sum is g * sigma^2
#pragma unroll 4
      for (int it = 0; it < a.wq_iters; ++it) {
        const float av = tp[min(to, tmax)], wl = cst[min(so, S - 1)];
        const float w = so < S ? wl : 0.f;
        g = fmaf(w, (av - thj) * is2, g);
        so += Q;
        to += tstep;
      }
    } else {
      const float base = a.eps_base_mode ? thj : a.a_seq[j];
      const float *op = same_w ? cst : omg;
#pragma unroll 4
      for (int it = 0; it < a.wq_iters; ++it) {
        const int sc = min(so, S - 1);
        const float av = tp[min(to, tmax)], wl = cst[sc], wol = op[sc];
        const float w = so < S ? wl : 0.f, wo = so < S ? wol : 0.f;
        g = fmaf(w, (av - thj) * is2, g);
        am = fmaf(wo, av - base, am);
        so += Q;
        to += tstep;
      }
    }
  }
  part[tid] = g;
  part[nt + tid] = am;
  if (a.merge_prior && tid < 16) {
    // combine weights of the JS <= 16 prior slices, once per particle: lane u holds slice u's (max, mass); 16-lane DPP
    // reductions give the overall max and the total mass; every column then only needs sum_u pA[u] w_u
    if (tid >= JS) {  // clamped duplicates of slice JS-1
      pmM1 = -INFINITY;
      pmL1 = 0.f;
    }
    const float m = row16_reduce(pmM1, -INFINITY, [](float x, float y) { return fmaxf(x, y); });
    const float w = (pmM1 == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((pmM1 - m) * 1.44269504088896340736f);
    const float l = row16_reduce(pmL1 * w, 0.f, [](float x, float y) { return x + y; });
    red[64 + tid] = w;
    red[80 + tid] = l;
  }
  lds_barrier();
  DUST_STAMP(a.stamps, 4);
  if (tid < D) {
    float gs = 0.f, as = 0.f;
    for (int qq = 0; qq < Q; ++qq) {
      gs += part[qq * D + tid];
      as += part[nt + qq * D + tid];
    }
    const size_t o = (size_t)n * D + tid;
    gs = gs / zw;
    if (same_w && a.eps_base_mode) as = gs * (pick_da<DA>(a.sigma_a, tid) * pick_da<DA>(a.sigma_a, tid));  // sum_s w (a - theta)
    else as = as / zo;
    a.grad_lik[o] = gs;
    if (a.update_a_mat) a.a_mat[o] = amv + as;
    if (a.merge_prior) {  // prior half of the score (svmpc.py:38-41,56) from the pairwise kernel's slice partials
      float gp;
      if (JS <= 16) {
        float acc = 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = fmaf(pmA[u], red[64 + u], acc);
        gp = (acc / red[80]) * pick_da<DA>(a.pm.inv_s2, tid);
      } else {
        float m, l;
        prior_merge_row(a.pm, nl, &m, &l);
        gp = prior_merge_col(a.pm, nl, D, tid, DA, m, l);
      }
      a.grad_pri[o] = gp;
      if (fw && fw->score_pub) {
        a.score[o] = gs + gp;
        __hip_atomic_store(fw->score_pub + o, gs + gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else if (fw && fw->score_cnt) {
        __hip_atomic_store(a.score + o, gs + gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        a.score[o] = gs + gp;
      }
    }
  }
  if (a.rearm && nl == 0)
    for (int t = tid; t < a.rearm_n; t += nt) a.rearm[t * CNT_STRIDE] = 0u;
  DUST_STAMP(a.stamps, 5);
  DUST_TL(fw ? fw->tl : nullptr, 2);
  if (fw && fw->score_cnt && !fw->score_pub) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its write-through stores ...
    wg_sync();                                    // ... before the one lane that signals for the workgroup
    if (threadIdx.x == 0)
      __hip_atomic_fetch_add(fw->score_cnt + (nl / fw->score_slice) * CNT_STRIDE, (unsigned int)fw->score_add, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
  }
}

// XCD-aware block -> work-item map.  Consecutive workgroup ids are dealt round-robin to the 8 XCDs (each with its own
// L2): give every XCD one CONTIGUOUS range of particles, so rows that share a 128-byte line (120-byte noise rows, the
// per-slice pM / pL words of 32 neighbours) are fetched into one L2 once instead of into several (measured with
// FETCH_SIZE: 2.2x the algorithmic bytes before).  Affinity only: results do not depend on the actual placement.
__device__ __forceinline__ int xcd_contiguous(int b, int nblocks) {
  return (nblocks & 7) ? b : (b & 7) * (nblocks >> 3) + (b >> 3);
}

template <int MODEL, bool GROUPS, bool LEAN, bool STATES = false>
__global__ __launch_bounds__(256) void rollout_kernel(const RolloutArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  rollout_body<MODEL, 32, GROUPS, LEAN, STATES>(a, lds, threadIdx.x, blockDim.x, xcd_contiguous(blockIdx.x, gridDim.x), nullptr);
}

// Same body under its own symbol for the HBM-streaming form (caller-supplied eps resident in HBM): profiles and PMC
// passes then attribute it separately from the Philox form.
template <int MODEL, bool GROUPS, bool LEAN, bool STATES = false>
__global__ __launch_bounds__(256) void rollout_stream_kernel(const RolloutArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  rollout_body<MODEL, 32, GROUPS, LEAN, STATES>(a, lds, threadIdx.x, blockDim.x, xcd_contiguous(blockIdx.x, gridDim.x), nullptr);
}

static inline int rollout_stage_bytes_per_lane(int model) {
  return model == DUST_MODEL_PARTICLE ? 2 * (DUST_STG_CHUNK_PART + 16) : DUST_STG_CHUNK + 16;
}
static inline size_t rollout_lds_bytes(int S, int D, int M, int nt, bool tile_in_lds, int stage_bytes_per_lane = 0, int grid_words = 0) {
  return sizeof(float) * ((tile_in_lds ? (size_t)S * (D | 1) : 0) + 2 * (size_t)S + 96 + 2 * (size_t)nt + 2 + (size_t)D + 2 * (size_t)M) + 16 +
         sizeof(uint32_t) * (size_t)grid_words +                     // occupancy grid (Particle)
         (size_t)(nt / 64) * 64 * stage_bytes_per_lane;  // + staging rows (chunk + 16 B) per lane when states are stored
}

}  // namespace dust
