// Stored-states rollouts, whole-line form (Particle family): the HBM-bound kernel of the path.
//
// MultiDISCO.forward returns every rollout's state trajectory (disco.py:394): [M][S][N][(H+1)][ds] floats, 11 GB at cfg3.  A
// trajectory is one (H+1)*16-byte run (656 B at H = 40) - NOT a multiple of the 128-byte line, so 2 of a row's ~6 lines are
// shared with the neighbouring rows n-1 / n+1.  Measured on MI355X (tools/store_pattern_bench.hip): a kernel that writes such
// rows from one workgroup per particle - whatever the staging - tops out at 2.6-3.2 TB/s, because the shared lines reach HBM as
// two partial writes from two different XCDs; the same pattern with 640-byte rows reaches 5.4 TB/s, the flat-fill rate.
//
// So here a wave owns GROUPS OF 8 ADJACENT particles: 8 rows = 8*16*(H+1) bytes = (H+1) whole lines.  lane = (sample s_sub = lane/8,
// particle n_sub = lane%8); a workgroup = 8 samples x 8 particles, its waves split the M dynamics samples and share the 64 action
// rows.  Only whole lines are ever stored:
//   - a lane keeps the last 8 states of its trajectory in registers (a ring indexed by row % 8, static after 8x unrolling); row
//     n_sub starts 16*p bytes into a line, p = ((H+1)*n_sub) % 8 - a different phase per lane (H+1 odd: p is a permutation) - so
//     at every time step exactly ONE particle per group completes a line: its 8 lanes (one per sample) dump their rings into a
//     128-byte transient line in LDS and the whole wave writes these 8 lines, 8 lanes x 16 bytes per line;
//   - the first line of a row with p != 0 is a HEAD shared with the previous row's TAIL: the head states (rows 0..6) stay in
//     registers until the rollout ends, then tails and heads are assembled in LDS into the 7 straddling lines per group.
// Two dynamics samples (m, m + GW) roll out side by side per lane in packed fp32 (common.hpp particle_pair_step), as in the
// lean kernel; costs go to a [S][N] buffer that the regular kernel then consumes (its injected-costs mode) for the softmax /
// weights / score stage.  Any non-finite operand or abnormal mass sends the workgroup down the general (reference-order,
// branchy) step functions - same storage scheme, bit-identical results to rollout_body's general loop.
#pragma once
#include "rollout.hpp"

namespace dust {

enum { SG_ROW = 144 };  // LDS bytes per staged 128-byte line (+16: bank spread)
enum { SP_FAST_FREE = 0, SP_FAST_OBST = 1, SP_FAST_CRASH = 2, SP_GENERAL = 3 };

template <int MODE>
__device__ __forceinline__ v2f sg_pair_step(const DevModel &dmk, const PairK &pk, const uint32_t *grid, const v2f m2, const v2f r2, v2f *xp, const float a0, const float a1,
                                            v2f *coll_io) {
  if (MODE == SP_GENERAL) {
    float xa[4], xb[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      xa[k] = xp[k].x;
      xb[k] = xp[k].y;
    }
    const float at[2] = {a0, a1};
    Coef ca, cb;
    ca.c0 = m2.x;
    cb.c0 = m2.y;
    ca.c1 = cb.c1 = 0.f;
    v2f c;
    c.x = step_with_cost<DUST_MODEL_PARTICLE>(dmk, ca, xa, at);  // (the rare path looks the map up in HBM: no private copy of the model)
    c.y = step_with_cost<DUST_MODEL_PARTICLE>(dmk, cb, xb, at);
#pragma unroll
    for (int k = 0; k < 4; ++k) xp[k] = (v2f){xa[k], xb[k]};
    return c;
  } else {
    return particle_pair_step<MODE != SP_FAST_FREE, MODE == SP_FAST_CRASH>(dmk, pk, grid, m2, r2, xp, a0, a1, particle_ctrl_cost(dmk, a0, a1), coll_io);
  }
}
template <int MODE>
__device__ __forceinline__ v2f sg_pair_term(const DevModel &dmk, const uint32_t *grid, const v2f *xp, const v2f *coll_in) {
  if (MODE == SP_GENERAL) {
    float xa[4], xb[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      xa[k] = xp[k].x;
      xb[k] = xp[k].y;
    }
    return (v2f){term_cost<DUST_MODEL_PARTICLE>(dmk, xa), term_cost<DUST_MODEL_PARTICLE>(dmk, xb)};
  } else {
    return particle_pair_term<MODE != SP_FAST_FREE>(dmk, grid, xp, coll_in);
  }
}

// One lane's two trajectories (m, m + GW) for every pair of this wave's share of the dynamics samples.
template <int MODE>
__device__ __forceinline__ double sg_roll_pairs(const RolloutArgs &a, const uint32_t *grid, const float *actl, const float *coefs, char *area, const int lane,
                                                const int w, const int GW, const bool live, char *gsg /* this lane's s-group, m = 0, + 16 * (lane % 8) */,
                                                const float *x0) {
  const int H = a.H, Hp1 = H + 1;
  const int j = lane & 7, sgrp = lane >> 3;
  const int pj = (Hp1 * j) & 7;  // this row's phase: it starts 16 * pj bytes into a line
  const int inv8 = Hp1 & 7;      // x * x = 1 (mod 8) for odd x: the particle whose phase is ph is (ph * inv8) % 8
  const uint32_t rowb = 16u * (uint32_t)Hp1;
  const size_t mstride = (size_t)a.S * a.N_total * rowb;  // bytes between consecutive dynamics samples
  char *const my_row = area + lane * SG_ROW;
  char *const prev_row = area + (lane - 1) * SG_ROW;
  char *const tr_w = area + (64 + sgrp) * SG_ROW;                 // transient line of this lane's s-group (sample B's dumps)
  const char *const tr_r = area + (64 + sgrp) * SG_ROW + j * 16;  // flusher role: piece j of it
  const char *const end_r = area + (lane & ~7) * SG_ROW + j * 16; // flusher role at the end: + it * SG_ROW
  // ring entry q (row % 8) sits in 16-byte slot (q + pj) % 8 of its line: the 8 slot addresses of the transient line (B's dumps);
  // the lane's own line in the staging area (A's rows, the tails at the end) is a constant `own` bytes below
  char *tw[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) tw[q] = tr_w + ((q + pj) & 7) * 16;
  const int own = (int)(my_row - tr_w);
  PairK pk;
  pk.load(a.dm);
  pk.pin();
  double acc = 0.0;
  for (int m = w; m + GW < a.M; m += 2 * GW) {
    const float ma = coefs[2 * m], mb = coefs[2 * (m + GW)];
    const v2f m2 = {ma, mb}, r2 = {1.0f / ma, 1.0f / mb};
    char *const gA = gsg + (size_t)m * mstride, *const gB = gA + (size_t)GW * mstride;
    v2f xp[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) xp[k] = (v2f){x0[k], x0[k]};
    v4f cB[8];  // sample B's state ring (row % 8); sample A's lives in the staging area (slot = (row + pj) % 8 of the lane's line)
    PairKahan tk2;
    // one row of both trajectories.  From row 7 on, the particle whose line completes at this row: A's line is read straight
    // from the area, B's 8 lanes dump their rings into the transient lines first; the wave stores 8 + 8 whole lines.
    auto emit = [&](const int q /* row % 8, static */, const int row) {
      *reinterpret_cast<v4f *>(tw[q] + own) = (v4f){xp[0].x, xp[1].x, xp[2].x, xp[3].x};
      cB[q] = (v4f){xp[0].y, xp[1].y, xp[2].y, xp[3].y};
      if (row < 7) return;  // (lines completing before row 7 are heads: assembled at the end)
      const int ph = (7 - q) & 7;
      const int jn = (ph * inv8) & 7;
      const uint32_t line = (rowb * (uint32_t)jn + 16u * (uint32_t)(row + 1)) / 128u - 1u;  // of the group, wave-uniform
      __builtin_amdgcn_wave_barrier();
      const v4f pa = *reinterpret_cast<const v4f *>(end_r + jn * SG_ROW);
      if (pj == ph) {
#pragma unroll
        for (int u = 0; u < 8; ++u) *reinterpret_cast<v4f *>(tw[u]) = cB[u];
      }
      __builtin_amdgcn_wave_barrier();
      const v4f pb = *reinterpret_cast<const v4f *>(tr_r);
      __builtin_amdgcn_wave_barrier();
      if (live) {
        *reinterpret_cast<v4f *>(gA + (size_t)line * 128u) = pa;
        *reinterpret_cast<v4f *>(gB + (size_t)line * 128u) = pb;
      }
    };
    // software pipeline across the steps: the next step's two actions and the occupancy of the state just produced are read
    // from LDS BEFORE the row is emitted, so their latency hides under the dump / store sequence (2 waves per SIMD only)
    float a0 = actl[0], a1 = actl[1];
    v2f coll = {0.f, 0.f};
    if (MODE == SP_FAST_OBST || MODE == SP_FAST_CRASH) coll = collision_pair(a.dm, grid, xp[0], xp[1]);
    auto step = [&](const int i /* static */, const int t) {
      const v2f c = sg_pair_step<MODE>(a.dm, pk, grid, m2, r2, xp, a0, a1, &coll);
      const int tn = min(t + 1, H - 1);
      a0 = actl[2 * tn];
      a1 = actl[2 * tn + 1];
      tk2.add(c);
      emit((i + 1) & 7, t + 1);
    };
    emit(0, 0);
    // (8x unrolled for the static ring index; the per-step test keeps every step its own scheduling region: with whole blocks as
    // one region the scheduler spends 60 more VGPRs - spills - for no measurable gain)
    for (int base = 0; base < H; base += 8) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (base + i < H) step(i, base + i);
    }
    const v2f tc = sg_pair_term<MODE>(a.dm, grid, xp, &coll);
    acc += (double)(tk2.sum.x + tc.x);
    acc += (double)(tk2.sum.y + tc.y);
    // ---- the 7 straddling lines of each group: tail of row j (slots 0 .. t-1) + head of row j+1 (slots t .. 7) ----
    // head rows 1..6 are rolled out AGAIN (same instructions, same operands: the same bits) rather than held in 48 registers
    // through the whole rollout
    v4f hA[6], hB[6];
    {
      v2f xh[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) xh[k] = (v2f){x0[k], x0[k]};
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        (void)sg_pair_step<MODE>(a.dm, pk, grid, m2, r2, xh, actl[2 * q], actl[2 * q + 1], nullptr);
        hA[q] = (v4f){xh[0].x, xh[1].x, xh[2].x, xh[3].x};
        hB[q] = (v4f){xh[0].y, xh[1].y, xh[2].y, xh[3].y};
      }
    }
    const v4f first = {x0[0], x0[1], x0[2], x0[3]};
    auto assemble = [&](const v4f *ring /* or nullptr: the tails are in the area already */, const v4f *head, char *g) {
      if (ring != nullptr && j < 7) {  // (row 7 of a group ends on a line boundary: no tail)
#pragma unroll
        for (int u = 0; u < 8; ++u) *reinterpret_cast<v4f *>(tw[u] + own) = ring[u];  // tail slots valid, the rest overwritten below
      }
      __builtin_amdgcn_wave_barrier();
      if (j >= 1) {  // pj >= 1: rows 0 .. 7-pj of this trajectory complete the previous row's last line
        *reinterpret_cast<v4f *>(prev_row + pj * 16) = first;
#pragma unroll
        for (int q = 1; q <= 6; ++q)
          if (q + pj <= 7) *reinterpret_cast<v4f *>(prev_row + (q + pj) * 16) = head[q - 1];
      }
      __builtin_amdgcn_wave_barrier();
      v4f pv[7];
#pragma unroll
      for (int it = 0; it < 7; ++it) pv[it] = *reinterpret_cast<const v4f *>(end_r + it * SG_ROW);
      __builtin_amdgcn_wave_barrier();
      if (live) {
#pragma unroll
        for (int it = 0; it < 7; ++it) *reinterpret_cast<v4f *>(g + (size_t)((rowb * (uint32_t)(it + 1)) / 128u) * 128u) = pv[it];
      }
    };
    assemble(nullptr, hA, gA);
    assemble(cB, hB, gB);
  }
  return acc;
}

// grid = (n_local / 8) * ceil(S / 8) workgroups of 64 * GW lanes.  Preconditions (checked by the host): Particle, fp32 noise
// or actions (NOISE_EPS / NOISE_ACTIONS) and fp32 states, H + 1 odd and >= 9, N_total, n0, n_local multiples of 8, M a multiple of 2 * GW, no parameter
// interleave, no sigma-point weights, a_reg == 0 (the second pass adds nothing to the costs)
// MODE: SP_FAST_* as the host's configuration says (obstacle map / crash semantics); a workgroup whose operands fail the fast
// path's preconditions raises its word of wg_flags and leaves - particle_states_kernel<SP_GENERAL>, launched right behind, rolls
// exactly those workgroups out with the general step functions (and returns at once everywhere else).  One kernel holding both
// paths would carry the general path's register pressure into the fast one (measured: 39 spilled VGPRs, +8 % time).
template <int MODE>
__global__ void __launch_bounds__(256, 2) particle_states_kernel(const RolloutArgs a, float *costs_sn, const int GW, unsigned int *wg_flags) {
  if (MODE == SP_GENERAL && wg_flags[blockIdx.x] == 0u) return;
  extern __shared__ float lds[];
  const int S = a.S, D = a.D, H = a.H, N = a.N_total, M = a.M;
  const int Dp = D | 1;
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, w = tid >> 6;
  const int nblk = (int)gridDim.x / ((S + 7) >> 3);  // particle groups
  const int nb = blockIdx.x % nblk, sb = blockIdx.x / nblk;
  const int n_first = a.n0 + nb * 8;
  float *tile = lds;                         // [64][Dp] action rows, row = lane
  float *coefs = tile + 64 * Dp;             // [M][2]
  float *flags = coefs + 2 * M;              // [4]
  const int off_grid = ((int)((flags + 4) - lds) + 3) & ~3;
  uint32_t *gridl = reinterpret_cast<uint32_t *>(lds + off_grid);
  const int off_acc = (off_grid + a.grid_words + 3) & ~3;
  double *accp = reinterpret_cast<double *>(lds + off_acc);                  // [GW][64]
  char *areas = reinterpret_cast<char *>(lds + off_acc + 2 * GW * 64);        // [GW][72][SG_ROW]
  if (tid == 0) flags[0] = 0.f;
  float x0[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) x0[k] = a.state[k];
  wg_sync();
  bool bad = false;
  for (int m = tid; m < M; m += nt) {
    float c0;
    if (a.coef_given) c0 = a.coef_host[0];
    else c0 = make_coef(a.dm, a.params ? a.params + (size_t)m * a.dm.P : nullptr).c0;
    coefs[2 * m] = c0;
    coefs[2 * m + 1] = 0.f;
    bad |= !(fabsf(c0) >= 1.0e-30f && fabsf(c0) <= 1.0e30f);
  }
  {
    const int words = a.dm.with_obstacle ? (a.dm.nx * a.dm.ny + 31) >> 5 : 0;
    for (int i = tid; i < words; i += nt) gridl[i] = a.dm.grid_bits[i];
  }
  // action rows: actions[s][n][:] = theta[n][:] + chol_a * eps[s][n][:] (or the caller's actions); the 8 particles of a sample are
  // one contiguous 8 D run
  for (int idx = tid; idx < 64 * D; idx += nt) {
    const int row = (int)__umulhi((uint32_t)idx, a.magicD), k = idx - row * D;
    const int s = min(sb * 8 + (row >> 3), S - 1), n = n_first + (row & 7);
    const float e = a.noise[((size_t)s * N + n) * D + k];
    const float thk = a.noise_mode == NOISE_EPS ? a.theta[(size_t)n * D + k] : 0.f;
    const float lk = a.noise_mode == NOISE_EPS ? ((k & 1) ? a.chol_a[1] : a.chol_a[0]) : 1.f;
    const float av = thk + lk * e;  // caller-supplied actions: 0 + 1 * v (exact), as rollout_body stages them
    tile[row * Dp + k] = av;
    bad |= av != av;
  }
  if (bad) flags[0] = 1.f;
  wg_sync();
  const bool fast = flags[0] == 0.f && fabsf(x0[0]) <= 3.0e38f && fabsf(x0[1]) <= 3.0e38f && fabsf(x0[2]) <= 3.0e38f && fabsf(x0[3]) <= 3.0e38f &&
                    (fabsf(x0[0]) + fabsf(x0[1]) + (fabsf(x0[2]) + fabsf(x0[3]) + a.dm.max_speed * (float)H) * fabsf((float)a.dm.dt)) * fabsf(a.dm.inv_cell) +
                                fabsf(a.dm.off_x) + fabsf(a.dm.off_y) <
                        1.0e17f;
  const int s = sb * 8 + (lane >> 3);
  const bool live = s < S;
  const int sc = live ? s : S - 1;
  const uint32_t rowb = 16u * (uint32_t)(H + 1);
  char *gsg = reinterpret_cast<char *>(a.states_out) + ((size_t)sc * N + n_first) * rowb + (lane & 7) * 16;
  const float *actl = tile + lane * Dp;
  char *area = areas + (size_t)w * 72 * SG_ROW;  // 64 staging lines + 8 transient lines per wave
  if (MODE != SP_GENERAL) {
    if (tid == 0) wg_flags[blockIdx.x] = fast ? 0u : 1u;
    if (!fast) return;
  }
  const double acc = sg_roll_pairs<MODE>(a, gridl, actl, coefs, area, lane, w, GW, live, gsg, x0);
  accp[w * 64 + lane] = acc;
  wg_sync();
  if (w == 0 && live) {  // fixed-order sum of the wave partials, then the mean over the dynamics samples (rollout_body finish_cost)
    double t = accp[lane];
    for (int g = 1; g < GW; ++g) t += accp[g * 64 + lane];
    const float cost = M == 1 ? (float)t : (float)(t / M);
    costs_sn[(size_t)s * N + n_first + (lane & 7)] = cost;
    a.costsT[(size_t)(n_first + (lane & 7)) * S + s] = cost;
  }
}

// =====================================================================================================================
// Pendulum family: states are 8 bytes, a trajectory is 8 (H+1) bytes (248 at H = 30) - groups of 16 ADJACENT particles are H+1 whole
// lines.  Same scheme as above with 16 slots per line: lane = (sample s_sub = lane / 16, particle n_sub = lane % 16), a workgroup =
// 16 samples x 16 particles (its four waves take four samples each), ONE trajectory per lane at a time (the M dynamics samples in
// sequence); every lane writes its state into slot (row + p) % 16 of its own line in the staging area, phase p = ((H+1) n_sub) % 16
// (H+1 odd: a permutation), so at every step one particle per group completes a line: half a wave reads these 4 lines, and every
// second step the whole wave stores 8 whole lines.  A head can be 15 of the 31 states here, so heads stay in registers (30 VGPRs;
// rolling them out again would add half the rollout).
// GENERAL = false: branch-free trig path (finite operands); a workgroup that cannot take it raises its flag and the GENERAL = true
// instance, launched right behind, rolls it out with the reference-order step functions.
template <bool GENERAL>
__global__ void __launch_bounds__(256, 2) pendulum_states_kernel(const RolloutArgs a, float *costs_sn, unsigned int *wg_flags) {
  if (GENERAL && wg_flags[blockIdx.x] == 0u) return;
  extern __shared__ float lds[];
  const int S = a.S, D = a.D, H = a.H, N = a.N_total, M = a.M, Hp1 = H + 1;
  const int Dp = D | 1;
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, w = tid >> 6;
  const int nblk = (int)gridDim.x / ((S + 15) >> 4);
  const int nb = blockIdx.x % nblk, sb = blockIdx.x / nblk;
  const int n_first = a.n0 + nb * 16;
  float *tile = lds;               // [256][Dp] action rows, row = tid
  float *coefs = tile + 256 * Dp;  // [M][2]
  float *flags = coefs + 2 * M;    // [4]
  const int off_area = ((int)((flags + 4) - lds) + 3) & ~3;
  char *areas = reinterpret_cast<char *>(lds + off_area);  // [4 waves][64][SG_ROW]: one staged line per lane
  if (tid == 0) flags[0] = 0.f;
  const float x0[2] = {a.state[0], a.state[1]};
  wg_sync();
  bool bad = false;
  for (int m = tid; m < M; m += nt) {
    Coef cf;
    if (a.coef_given) {
      cf.c0 = a.coef_host[0];
      cf.c1 = a.coef_host[1];
    } else {
      cf = make_coef(a.dm, a.params ? a.params + (size_t)m * a.dm.P : nullptr);
    }
    coefs[2 * m] = cf.c0;
    coefs[2 * m + 1] = cf.c1;
    bad |= !(fabsf(cf.c0) <= 3.0e38f && fabsf(cf.c1) <= 3.0e38f);
  }
  for (int idx = tid; idx < 256 * D; idx += nt) {
    const int row = (int)__umulhi((uint32_t)idx, a.magicD), k = idx - row * D;
    const int s = min(sb * 16 + (row >> 6) * 4 + ((row & 63) >> 4), S - 1), n = n_first + (row & 15);
    const float e = a.noise[((size_t)s * N + n) * D + k];
    const float thk = a.noise_mode == NOISE_EPS ? a.theta[(size_t)n * D + k] : 0.f;
    const float lk = a.noise_mode == NOISE_EPS ? a.chol_a[0] : 1.f;
    const float av = thk + lk * e;
    tile[row * Dp + k] = av;
    bad |= av != av;
  }
  if (bad) flags[0] = 1.f;
  wg_sync();
  const bool fast = flags[0] == 0.f && fabsf(x0[1]) <= 3.0e38f && (fabsf(x0[0]) + a.dm.max_speed_pend * (float)a.dm.dt * (float)H < 5.0e4f);
  if (!GENERAL) {
    if (tid == 0) wg_flags[blockIdx.x] = fast ? 0u : 1u;
    if (!fast) return;
  }
  const int j = lane & 15, ssub = lane >> 4;
  const int s = sb * 16 + w * 4 + ssub;
  const bool live = s < S;
  const int sc = live ? s : S - 1;
  // flusher role: each half of the wave owns (line of sample sf, 16-byte piece); the halves take alternate steps
  const int sf_sub = (lane >> 3) & 3, piece = lane & 7;
  const int sf = sb * 16 + w * 4 + sf_sub;
  const bool flive = sf < S;
  const bool half1 = lane >= 32;
  const uint32_t rowb = 8u * (uint32_t)Hp1;
  const size_t mstride = (size_t)S * N * rowb;
  char *const gF = reinterpret_cast<char *>(a.states_out) + ((size_t)min(sf, S - 1) * N + n_first) * rowb + piece * 16;
  const int pj = (Hp1 * j) & 15;
  int inv16 = 1;
  for (int c = 1; c < 16; c += 2)
    if (((Hp1 * c) & 15) == 1) inv16 = c;
  char *const area = areas + (size_t)w * 64 * SG_ROW;
  char *const my_row = area + lane * SG_ROW;
  char *const prev_row = area + (lane - 1) * SG_ROW;
  const char *const fl_r = area + (sf_sub * 16) * SG_ROW + piece * 16;  // flusher role: + jn * SG_ROW
  const float *actl = tile + tid * Dp;
  const float dt = (float)a.dm.dt, mt = a.dm.max_torque, ms = a.dm.max_speed_pend;
  const v2f W = {a.dm.w_cos, a.dm.w_vel};
  double acc = 0.0;
  for (int m = 0; m < M; ++m) {
    Coef cf;
    cf.c0 = coefs[2 * m];
    cf.c1 = coefs[2 * m + 1];
    char *const g = gF + (size_t)m * mstride;
    float x[2] = {x0[0], x0[1]};
    v2f head[15];  // rows 0 .. 14
    v4f pv = {0.f, 0.f, 0.f, 0.f};  // a line read by the first half of the wave, stored together with the second half's next step
    uint32_t pline = 0u;
    CostSum<DUST_MODEL_PENDULUM> tot;
    auto emit = [&](const int q /* row % 16, static */, const int row) {
      *reinterpret_cast<v2f *>(my_row + (((q + pj) & 15) << 3)) = (v2f){x[0], x[1]};
      if (row < 15) {  // (lines completing before row 15 are heads)
        head[q < 15 ? q : 0] = (v2f){x[0], x[1]};
        return;
      }
      const int ph = (15 - q) & 15;
      const int jn = (ph * inv16) & 15;
      const uint32_t line = (rowb * (uint32_t)jn + 8u * (uint32_t)(row + 1)) / 128u - 1u;
      __builtin_amdgcn_wave_barrier();
      if (((row - 15) & 1) == 0) {  // first of a pair of steps: lanes 0..31 take this line
        if (!half1) {
          pv = *reinterpret_cast<const v4f *>(fl_r + jn * SG_ROW);
          pline = line;
        }
      } else {                      // second: lanes 32..63 take this one, the whole wave stores
        if (half1) {
          pv = *reinterpret_cast<const v4f *>(fl_r + jn * SG_ROW);
          pline = line;
        }
        if (flive) *reinterpret_cast<v4f *>(g + (size_t)pline * 128u) = pv;
      }
      __builtin_amdgcn_wave_barrier();
    };
    auto step = [&](const int t) {
      const float at[1] = {actl[t]};
      if (GENERAL) {
        tot.add(step_with_cost<DUST_MODEL_PENDULUM>(a.dm, cf, x, at), t);  // cost of the state BEFORE the action (disco.py:306)
      } else {
        float sn, cs;
        pendulum_trig(x[0], &sn, &cs);
        v2f qv = {cs - 1.0f, x[1]};
        qv = W * (qv * qv);
        tot.add(qv.x + qv.y, t);
        const float u = __builtin_amdgcn_fmed3f(at[0], -mt, mt);
        float thd = x[1] + dt * (cf.c0 * sn + cf.c1 * u);
        thd = __builtin_amdgcn_fmed3f(thd, -ms, ms);
        x[0] = x[0] + thd * dt;
        x[1] = thd;
      }
    };
    emit(0, 0);
    for (int base = 0; base < H; base += 16) {
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (base + i < H) {
          step(base + i);
          emit((i + 1) & 15, base + i + 1);
        }
    }
    float traj;
    if (GENERAL) {
      traj = (float)tot.total() + term_cost<DUST_MODEL_PENDULUM>(a.dm, x);
    } else {
      float sn, cs;
      pendulum_trig(x[0], &sn, &cs);
      v2f qv = {cs - 1.0f, x[1]};
      qv = W * (qv * qv);
      traj = (float)tot.total() + (qv.x + qv.y);
    }
    acc += (double)traj;
    if (((H - 15) & 1) == 0 && !half1 && flive) *reinterpret_cast<v4f *>(g + (size_t)pline * 128u) = pv;  // an unpaired last line
    // ---- the 15 straddling lines of each group: tail of row j (slots 0 .. t-1, in the area already) + head of row j+1 ----
    __builtin_amdgcn_wave_barrier();
    if (j >= 1) {  // pj >= 1: rows 0 .. 15-pj of this trajectory complete the previous row's last line
#pragma unroll
      for (int q = 0; q < 15; ++q)
        if (q + pj <= 15) *reinterpret_cast<v2f *>(prev_row + (q + pj) * 8) = head[q];
    }
    __builtin_amdgcn_wave_barrier();
    // 4 samples x 15 lines x 8 pieces = 480 pieces over the wave's 64 lanes
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int P = it * 64 + lane, ln = P >> 3, pc = P & 7;
      const int ss = ln / 15, jb = ln - ss * 15;
      if (ln < 60) {
        const v4f pv = *reinterpret_cast<const v4f *>(area + (ss * 16 + jb) * SG_ROW + pc * 16);
        const int sg = sb * 16 + w * 4 + ss;
        if (sg < S)
          *reinterpret_cast<v4f *>(reinterpret_cast<char *>(a.states_out) + ((size_t)m * S * N + (size_t)sg * N + n_first) * rowb +
                                   (size_t)((rowb * (uint32_t)(jb + 1)) / 128u) * 128u + pc * 16) = pv;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (live) {
    const float cost = M == 1 ? (float)acc : (float)(acc / M);
    costs_sn[(size_t)s * N + n_first + j] = cost;
    a.costsT[(size_t)(n_first + j) * S + s] = cost;
  }
}

// =====================================================================================================================
// Pendulum family, binary16 states (DUST_STORE_F16; the arithmetic stays fp32): a state is 4 bytes and a trajectory 4 (H+1) bytes
// (124 at H = 30) - SHORTER than a line, so no line of the output is complete before the trajectories end and there is nothing to
// stream.  A wave rolls out 2 samples x 32 ADJACENT particles: per sample 32 rows of 4 (H+1) bytes = H+1 whole lines, line-aligned
// because N and the group's first particle are multiples of 32.  Every lane writes its rows into an LDS IMAGE of that output block
// (row stride H+1 words: conflict-free when H+1 is odd) and the wave copies the image out in whole-wave 16-byte stores: 2 (H+1)
// whole lines per dynamics sample, nothing partial, nothing written twice.  A workgroup = 8 samples x 32 particles.
// GENERAL as above.
template <bool GENERAL>
__global__ void __launch_bounds__(256, 2) pendulum_states_f16_kernel(const RolloutArgs a, float *costs_sn, unsigned int *wg_flags) {
  if (GENERAL && wg_flags[blockIdx.x] == 0u) return;
  extern __shared__ float lds[];
  const int S = a.S, D = a.D, H = a.H, N = a.N_total, M = a.M, Hp1 = H + 1;
  const int Dp = D | 1;
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, w = tid >> 6;
  const int nblk = (int)gridDim.x / ((S + 7) >> 3);
  const int nb = blockIdx.x % nblk, sb = blockIdx.x / nblk;
  const int n_first = a.n0 + nb * 32;
  float *tile = lds;               // [256][Dp] action rows, row = tid
  float *coefs = tile + 256 * Dp;  // [M][2]
  float *flags = coefs + 2 * M;    // [4]
  const int off_img = ((int)((flags + 4) - lds) + 3) & ~3;
  uint32_t *const img = reinterpret_cast<uint32_t *>(lds + off_img) + (size_t)w * 64 * Hp1;  // [2 samples][32 particles][H+1] packed states
  if (tid == 0) flags[0] = 0.f;
  const float x0[2] = {a.state[0], a.state[1]};
  wg_sync();
  bool bad = false;
  for (int m = tid; m < M; m += nt) {
    Coef cf;
    if (a.coef_given) {
      cf.c0 = a.coef_host[0];
      cf.c1 = a.coef_host[1];
    } else {
      cf = make_coef(a.dm, a.params ? a.params + (size_t)m * a.dm.P : nullptr);
    }
    coefs[2 * m] = cf.c0;
    coefs[2 * m + 1] = cf.c1;
    bad |= !(fabsf(cf.c0) <= 3.0e38f && fabsf(cf.c1) <= 3.0e38f);
  }
  for (int idx = tid; idx < 256 * D; idx += nt) {
    const int row = (int)__umulhi((uint32_t)idx, a.magicD), k = idx - row * D;
    const int s = min(sb * 8 + (row >> 6) * 2 + ((row & 63) >> 5), S - 1), n = n_first + (row & 31);
    const float e = a.noise[((size_t)s * N + n) * D + k];
    const float thk = a.noise_mode == NOISE_EPS ? a.theta[(size_t)n * D + k] : 0.f;
    const float lk = a.noise_mode == NOISE_EPS ? a.chol_a[0] : 1.f;
    const float av = thk + lk * e;
    tile[row * Dp + k] = av;
    bad |= av != av;
  }
  if (bad) flags[0] = 1.f;
  wg_sync();
  const bool fast = flags[0] == 0.f && fabsf(x0[1]) <= 3.0e38f && (fabsf(x0[0]) + a.dm.max_speed_pend * (float)a.dm.dt * (float)H < 5.0e4f);
  if (!GENERAL) {
    if (tid == 0) wg_flags[blockIdx.x] = fast ? 0u : 1u;
    if (!fast) return;
  }
  const int j = lane & 31, ssub = lane >> 5;
  const int s = sb * 8 + w * 2 + ssub;
  const bool live = s < S;
  uint32_t *const my = img + lane * Hp1;
  const float *actl = tile + tid * Dp;
  const float dt = (float)a.dm.dt, mt = a.dm.max_torque, ms = a.dm.max_speed_pend;
  const v2f W = {a.dm.w_cos, a.dm.w_vel};
  const uint32_t rowb = 4u * (uint32_t)Hp1;
  const int pieces = 16 * Hp1;  // 16-byte pieces of the wave's image: 2 samples x (H+1) lines x 8
  double acc = 0.0;
  for (int m = 0; m < M; ++m) {
    Coef cf;
    cf.c0 = coefs[2 * m];
    cf.c1 = coefs[2 * m + 1];
    float x[2] = {x0[0], x0[1]};
    CostSum<DUST_MODEL_PENDULUM> tot;
    auto emit = [&](const int row) {
      const _Float16 h[2] = {(_Float16)x[0], (_Float16)x[1]};
      my[row] = *reinterpret_cast<const uint32_t *>(h);
    };
    emit(0);
#pragma unroll 2
    for (int t = 0; t < H; ++t) {
      const float at[1] = {actl[t]};
      if (GENERAL) {
        tot.add(step_with_cost<DUST_MODEL_PENDULUM>(a.dm, cf, x, at), t);  // cost of the state BEFORE the action (disco.py:306)
      } else {
        float sn, cs;
        pendulum_trig(x[0], &sn, &cs);
        v2f qv = {cs - 1.0f, x[1]};
        qv = W * (qv * qv);
        tot.add(qv.x + qv.y, t);
        const float u = __builtin_amdgcn_fmed3f(at[0], -mt, mt);
        float thd = x[1] + dt * (cf.c0 * sn + cf.c1 * u);
        thd = __builtin_amdgcn_fmed3f(thd, -ms, ms);
        x[0] = x[0] + thd * dt;
        x[1] = thd;
      }
      emit(t + 1);
    }
    float traj;
    if (GENERAL) {
      traj = (float)tot.total() + term_cost<DUST_MODEL_PENDULUM>(a.dm, x);
    } else {
      float sn, cs;
      pendulum_trig(x[0], &sn, &cs);
      v2f qv = {cs - 1.0f, x[1]};
      qv = W * (qv * qv);
      traj = (float)tot.total() + (qv.x + qv.y);
    }
    acc += (double)traj;
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int P = lane; P < pieces; P += 64) {
      const int ss = P >= 8 * Hp1 ? 1 : 0, pc = P - ss * 8 * Hp1;
      const int sg = sb * 8 + w * 2 + ss;
      const v4f pv = *reinterpret_cast<const v4f *>(img + 4 * P);
      if (sg < S)
        *reinterpret_cast<v4f *>(reinterpret_cast<char *>(a.states_out) + (((size_t)m * S + sg) * N + n_first) * rowb + (size_t)pc * 16) = pv;
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (live) {
    const float cost = M == 1 ? (float)acc : (float)(acc / M);
    costs_sn[(size_t)s * N + n_first + j] = cost;
    a.costsT[(size_t)(n_first + j) * S + s] = cost;
  }
}

// =====================================================================================================================
// Particle family, binary16 states (DUST_STORE_F16; the arithmetic stays fp32): a state is 8 bytes, a trajectory 8 (H+1) bytes (328 at
// H = 40) - the geometry of the Pendulum's fp32 rows above: groups of 16 ADJACENT particles are H+1 whole lines, 16 slots per line.
// Rollouts as in particle_states_kernel (two dynamics samples (m, m + GW) side by side per lane in packed fp32, the workgroup's waves
// split the dynamics samples and share 64 action rows), storage as in pendulum_states_kernel: lane = (sample lane / 16, particle
// lane % 16), a workgroup = 4 samples x 16 particles; trajectory A is staged in the lane's own LDS line (slot (row + p) % 16),
// trajectory B's last 16 states ride in a register ring and are dumped into a transient line when a line of its group completes; at
// every step from row 15 on lanes 0-31 fetch the 4 completed lines of A, lanes 32-63 those of B, and the wave stores 8 whole lines.
// Heads (rows 0 .. 14 of both trajectories) stay in registers; the 15 straddling lines per group are assembled at the end.
// MODE as particle_states_kernel (SP_GENERAL: the flagged workgroups, reference-order step functions).
template <int MODE>
__global__ void __launch_bounds__(256, 2) particle_states_f16_kernel(const RolloutArgs a, float *costs_sn, const int GW, unsigned int *wg_flags) {
  if (MODE == SP_GENERAL && wg_flags[blockIdx.x] == 0u) return;
  extern __shared__ float lds[];
  const int S = a.S, D = a.D, H = a.H, N = a.N_total, M = a.M, Hp1 = H + 1;
  const int Dp = D | 1;
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, w = tid >> 6;
  const int nblk = (int)gridDim.x / ((S + 3) >> 2);  // particle groups of 16
  const int nb = blockIdx.x % nblk, sb = blockIdx.x / nblk;
  const int n_first = a.n0 + nb * 16;
  float *tile = lds;                         // [64][Dp] action rows, row = lane
  float *coefs = tile + 64 * Dp;             // [M][2]
  float *flags = coefs + 2 * M;              // [4]
  const int off_grid = ((int)((flags + 4) - lds) + 3) & ~3;
  uint32_t *gridl = reinterpret_cast<uint32_t *>(lds + off_grid);
  const int off_acc = (off_grid + a.grid_words + 3) & ~3;
  double *accp = reinterpret_cast<double *>(lds + off_acc);                  // [GW][64]
  char *areas = reinterpret_cast<char *>(lds + off_acc + 2 * GW * 64);        // [GW][68][SG_ROW]: 64 staged lines + 4 transient lines per wave
  if (tid == 0) flags[0] = 0.f;
  float x0[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) x0[k] = a.state[k];
  wg_sync();
  bool bad = false;
  for (int m = tid; m < M; m += nt) {
    float c0;
    if (a.coef_given) c0 = a.coef_host[0];
    else c0 = make_coef(a.dm, a.params ? a.params + (size_t)m * a.dm.P : nullptr).c0;
    coefs[2 * m] = c0;
    coefs[2 * m + 1] = 0.f;
    bad |= !(fabsf(c0) >= 1.0e-30f && fabsf(c0) <= 1.0e30f);
  }
  {
    const int words = a.dm.with_obstacle ? (a.dm.nx * a.dm.ny + 31) >> 5 : 0;
    for (int i = tid; i < words; i += nt) gridl[i] = a.dm.grid_bits[i];
  }
  for (int idx = tid; idx < 64 * D; idx += nt) {  // the 16 particles of a sample are one contiguous 16 D run
    const int row = (int)__umulhi((uint32_t)idx, a.magicD), k = idx - row * D;
    const int s = min(sb * 4 + (row >> 4), S - 1), n = n_first + (row & 15);
    const float e = a.noise[((size_t)s * N + n) * D + k];
    const float thk = a.noise_mode == NOISE_EPS ? a.theta[(size_t)n * D + k] : 0.f;
    const float lk = a.noise_mode == NOISE_EPS ? ((k & 1) ? a.chol_a[1] : a.chol_a[0]) : 1.f;
    const float av = thk + lk * e;
    tile[row * Dp + k] = av;
    bad |= av != av;
  }
  if (bad) flags[0] = 1.f;
  wg_sync();
  const bool fast = flags[0] == 0.f && fabsf(x0[0]) <= 3.0e38f && fabsf(x0[1]) <= 3.0e38f && fabsf(x0[2]) <= 3.0e38f && fabsf(x0[3]) <= 3.0e38f &&
                    (fabsf(x0[0]) + fabsf(x0[1]) + (fabsf(x0[2]) + fabsf(x0[3]) + a.dm.max_speed * (float)H) * fabsf((float)a.dm.dt)) * fabsf(a.dm.inv_cell) +
                                fabsf(a.dm.off_x) + fabsf(a.dm.off_y) <
                        1.0e17f;
  if (MODE != SP_GENERAL) {
    if (tid == 0) wg_flags[blockIdx.x] = fast ? 0u : 1u;
    if (!fast) return;
  }
  const int j = lane & 15, ssub = lane >> 4;
  const int s = sb * 4 + ssub;
  const bool live = s < S;
  // flusher role: lanes 0-31 take trajectory A's lines, lanes 32-63 B's: (sample sf_sub, 16-byte piece)
  const int sf_sub = (lane >> 3) & 3, piece = lane & 7;
  const bool halfB = lane >= 32;
  const bool flive = sb * 4 + sf_sub < S;
  const uint32_t rowb = 8u * (uint32_t)Hp1;
  const size_t mstride = (size_t)S * N * rowb;  // bytes between consecutive dynamics samples
  char *const gF = reinterpret_cast<char *>(a.states_out) + ((size_t)min(sb * 4 + sf_sub, S - 1) * N + n_first) * rowb + piece * 16;
  const int pj = (Hp1 * j) & 15;
  int inv16 = 1;
  for (int c = 1; c < 16; c += 2)
    if (((Hp1 * c) & 15) == 1) inv16 = c;
  char *const area = areas + (size_t)w * 68 * SG_ROW;
  char *const my_row = area + lane * SG_ROW;
  char *const prev_row = area + (lane - 1) * SG_ROW;
  char *const tr_w = area + (64 + ssub) * SG_ROW;  // transient line of this lane's sample (B's dumps)
  const char *const fl_r = halfB ? area + (64 + sf_sub) * SG_ROW + piece * 16 : area + (sf_sub * 16) * SG_ROW + piece * 16;  // (A: + jn * SG_ROW)
  const float *actl = tile + lane * Dp;
  typedef unsigned int u2 __attribute__((ext_vector_type(2)));
  auto pack = [](const float x, const float y, const float z, const float v) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 lo = {(_Float16)x, (_Float16)y}, hi = {(_Float16)z, (_Float16)v};
    return u2{__builtin_bit_cast(unsigned int, lo), __builtin_bit_cast(unsigned int, hi)};
  };
  PairK pk;
  pk.load(a.dm);
  pk.pin();
  double acc = 0.0;
  for (int m = w; m + GW < M; m += 2 * GW) {
    const float ma = coefs[2 * m], mb = coefs[2 * (m + GW)];
    const v2f m2 = {ma, mb}, r2 = {1.0f / ma, 1.0f / mb};
    char *const gA = gF + (size_t)m * mstride, *const gB = gA + (size_t)GW * mstride;
    char *const gX = halfB ? gB : gA;
    v2f xp[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) xp[k] = (v2f){x0[k], x0[k]};
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    u4 cB2[8];            // trajectory B's ring (row % 16): entries (2 v, 2 v + 1) share a register quad - half of the dumps are 16-byte writes
    u2 hA[15], hB[15];    // rows 0 .. 14 of both
    PairKahan tk2;
    auto emit = [&](const int q /* row % 16, static */, const int row) {
      const u2 sa = pack(xp[0].x, xp[1].x, xp[2].x, xp[3].x), sbv = pack(xp[0].y, xp[1].y, xp[2].y, xp[3].y);
      *reinterpret_cast<u2 *>(my_row + (((q + pj) & 15) << 3)) = sa;
      if (q & 1) {
        cB2[q >> 1].z = sbv.x;
        cB2[q >> 1].w = sbv.y;
      } else {
        cB2[q >> 1].x = sbv.x;
        cB2[q >> 1].y = sbv.y;
      }
      if (row < 15) {  // (lines completing before row 15 are heads)
        hA[q < 15 ? q : 0] = sa;
        hB[q < 15 ? q : 0] = sbv;
        return;
      }
      const int ph = (15 - q) & 15;
      const int jn = (ph * inv16) & 15;
      const uint32_t line = (rowb * (uint32_t)jn + 8u * (uint32_t)(row + 1)) / 128u - 1u;
      __builtin_amdgcn_wave_barrier();
      if (pj == ph) {  // (ph is static: ring entry u goes to the static slot (u + ph) % 16; with ph even the entry pairs of a register quad are
                       //  aligned 16-byte pairs of slots - 8 writes -, with ph odd they straddle: 16 writes of 8 bytes)
        if ((ph & 1) == 0) {
#pragma unroll
          for (int v = 0; v < 8; ++v) *reinterpret_cast<u4 *>(tr_w + (((2 * v + ph) & 15) << 3)) = cB2[v];
        } else {
#pragma unroll
          for (int v = 0; v < 8; ++v) {
            *reinterpret_cast<u2 *>(tr_w + (((2 * v + ph) & 15) << 3)) = u2{cB2[v].x, cB2[v].y};
            *reinterpret_cast<u2 *>(tr_w + (((2 * v + 1 + ph) & 15) << 3)) = u2{cB2[v].z, cB2[v].w};
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
      const v4f pv = *reinterpret_cast<const v4f *>(fl_r + (halfB ? 0 : jn * SG_ROW));
      __builtin_amdgcn_wave_barrier();
      if (flive) *reinterpret_cast<v4f *>(gX + (size_t)line * 128u) = pv;
    };
    float a0 = actl[0], a1 = actl[1];
    v2f coll = {0.f, 0.f};
    if (MODE == SP_FAST_OBST || MODE == SP_FAST_CRASH) coll = collision_pair(a.dm, gridl, xp[0], xp[1]);
    auto step = [&](const int i /* static */, const int t) {
      const v2f c = sg_pair_step<MODE>(a.dm, pk, gridl, m2, r2, xp, a0, a1, &coll);
      const int tn = min(t + 1, H - 1);
      a0 = actl[2 * tn];
      a1 = actl[2 * tn + 1];
      tk2.add(c);
      emit((i + 1) & 15, t + 1);
    };
    emit(0, 0);
    for (int base = 0; base < H; base += 16) {
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (base + i < H) step(i, base + i);
    }
    const v2f tc = sg_pair_term<MODE>(a.dm, gridl, xp, &coll);
    acc += (double)(tk2.sum.x + tc.x);
    acc += (double)(tk2.sum.y + tc.y);
    // ---- the 15 straddling lines of each group: tail of row j (slots 0 .. t-1) + head of row j+1 (slots t .. 15) ----
    auto assemble = [&](const bool ring /* B: the tails are in the register ring */, const u2 *head, char *g0 /* + m stride, piece 0 of sample 0's group */) {
      __builtin_amdgcn_wave_barrier();
      if (ring && j < 15) {  // (row 15 of a group ends on a line boundary: no tail)
#pragma unroll
        for (int u = 0; u < 16; ++u)  // tail slots valid, the rest overwritten below
          *reinterpret_cast<u2 *>(my_row + (((u + pj) & 15) << 3)) = (u & 1) ? u2{cB2[u >> 1].z, cB2[u >> 1].w} : u2{cB2[u >> 1].x, cB2[u >> 1].y};
      }
      __builtin_amdgcn_wave_barrier();
      if (j >= 1) {  // pj >= 1: rows 0 .. 15-pj of this trajectory complete the previous row's last line
#pragma unroll
        for (int q = 0; q < 15; ++q)
          if (q + pj <= 15) *reinterpret_cast<u2 *>(prev_row + (q + pj) * 8) = head[q];
      }
      __builtin_amdgcn_wave_barrier();
      // 4 samples x 15 lines x 8 pieces = 480 pieces over the wave's 64 lanes
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int P = it * 64 + lane, ln = P >> 3, pc = P & 7;
        const int ss = ln / 15, jb = ln - ss * 15;
        if (ln < 60) {
          const v4f pv = *reinterpret_cast<const v4f *>(area + (ss * 16 + jb) * SG_ROW + pc * 16);
          const int sg = sb * 4 + ss;
          if (sg < S)
            *reinterpret_cast<v4f *>(g0 + ((size_t)sg * N + n_first) * rowb + (size_t)((rowb * (uint32_t)(jb + 1)) / 128u) * 128u + pc * 16) = pv;
        }
      }
      __builtin_amdgcn_wave_barrier();
    };
    char *const base_out = reinterpret_cast<char *>(a.states_out);
    assemble(false, hA, base_out + (size_t)m * mstride);
    assemble(true, hB, base_out + (size_t)(m + GW) * mstride);
  }
  accp[w * 64 + lane] = acc;
  wg_sync();
  if (w == 0 && live) {  // fixed-order sum of the wave partials, then the mean over the dynamics samples (rollout_body finish_cost)
    double t = accp[lane];
    for (int g = 1; g < GW; ++g) t += accp[g * 64 + lane];
    const float cost = M == 1 ? (float)t : (float)(t / M);
    costs_sn[(size_t)s * N + n_first + j] = cost;
    a.costsT[(size_t)(n_first + j) * S + s] = cost;
  }
}

static inline size_t particle_states_f16_lds_bytes(int D, int M, int grid_words, int GW) {
  const size_t floats = (size_t)64 * (D | 1) + 2 * (size_t)M + 4 + 4 + (size_t)grid_words + 4 + 2 * (size_t)GW * 64;
  return floats * sizeof(float) + (size_t)GW * 68 * SG_ROW + 16;
}

static inline size_t pendulum_states_f16_lds_bytes(int D, int M, int H) {
  const size_t floats = (size_t)256 * (D | 1) + 2 * (size_t)M + 4 + 4;
  return floats * sizeof(float) + (size_t)4 * 64 * (H + 1) * 4 + 16;
}

static inline size_t pendulum_states_lds_bytes(int D, int M) {
  const size_t floats = (size_t)256 * (D | 1) + 2 * (size_t)M + 4 + 4;
  return floats * sizeof(float) + (size_t)4 * 64 * SG_ROW + 16;
}

static inline size_t particle_states_lds_bytes(int D, int M, int grid_words, int GW) {
  const size_t floats = (size_t)64 * (D | 1) + 2 * (size_t)M + 4 + 4 + (size_t)grid_words + 4 + 2 * (size_t)GW * 64;
  return floats * sizeof(float) + (size_t)GW * 72 * SG_ROW + 16;
}

}  // namespace dust
