// persist.hpp - ONE launch per control tick: SVMPC.optimize (n_iters SVGD iterations) + SVMPC.forward as a persistent kernel.
//
// Replaces (reference file:line): the loop of SVMPC.optimize svmpc.py:97-126 around SVMPC.step svmpc.py:87-95 (sample ->
// phi -> optimiser step) and SVMPC.forward svmpc.py:172-200 (get_weights, argmax, roll, update_prior), i.e. the call pair
// `optimize(); forward()` of dust/utils/simulations.py:104-123, for the shapes whose whole tick fits on the chip at once.
//
// Why (MI355X): at cfg2 sizes (N = 1024 particles, S = 128 samples, H = 30) an SVGD iteration is ~5 us of issue time spread
// over a chain of four all-to-all hand-offs; one launch per iteration (fused.hpp svgd_iter_kernel) spends 25 us on it because
// every launch regenerates the policy noise (Philox + Box-Muller: ~40 % of the rollout role's instructions) at the head of
// its critical path, refills every LDS tile, and pays a launch ramp.  Here the grid is resident for the whole tick:
//   * OWNER workgroups (256 / nt particles each) keep their particles' state on the CU across iterations - theta row, a_mat
//     row, Adam moments in registers, the S x D action tile in LDS - and draw the NEXT iteration's noise into that tile while
//     they wait for the Stein partials (the tile is free from the weighted reductions on): noise generation leaves the
//     critical path;
//   * PAIR workgroups (32-query x 64-key tiles, stein.hpp) run prior tile -> Stein tile per iteration, then the log-density
//     tile of forward();
//   * the plant state is a by-value kernel argument: no set-state launch, no graph, one hipLaunchKernel per tick.
// Hand-offs are the write-through form of cdna_hip_programming.md Guideline 16 (sc1 stores -> every storing wave drains
// vmcnt -> workgroup barrier -> one lane's agent-scope add; consumers poll with ONE lane per line, barrier, then only sc1
// loads).  Counters are monotonic within a tick (target = iterations so far x arrivals per iteration), kept in two sets:
// a tick uses one and zeroes the other.  Arrival lines, one 128-byte line each, per 32-particle group g (= query tile):
//   cnt_theta[g]  theta(k) rows of group g written         (owners -> pair tiles that use g as queries or keys)
//   cnt_prior[g]  prior / log-density partials of tile g   (pair -> owners of g), JS arrivals per pass
//   cnt_score[g]  score rows of group g                    (owners -> pair tiles with keys in g)
//   cnt_stein[g]  Stein partials of tile g                 (pair -> owners of g), JS arrivals per iteration
//   cnt_lw[g]     log-weights of group g                   (owners -> every owner: forward's softmax over all particles)
// theta is double-buffered (iteration k reads buf[k & 1], its update writes buf[(k + 1) & 1]): a writer of theta(k + 2) has
// waited on Stein partials that needed every theta(k + 1) row, hence every reader of theta(k) has finished.  One score buffer
// and one set of partial buffers suffice by the same argument.
// Every workgroup must be RESIDENT (they wait on each other): the host launches this kernel only when the grid fits the
// occupancy the runtime reports (minus nothing else running on the device is NOT assumed: every wait is bounded by
// wall-clock time and a time-out is reported by the host as an error, never a hang).
#pragma once
#include "forward.hpp"
#include "fused.hpp"

namespace dust {

#ifdef DUST_STAMPS
#define DUST_TLK(p, k)                                                                                                 \
  do {                                                                                                                 \
    if ((p) && threadIdx.x == 0 && (k) < 128) (p)[128 * blockIdx.x + (k)] = (unsigned long long)__builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define DUST_TLK(p, k) \
  do {                 \
  } while (0)
#endif

struct TickArgs {
  PairArgs prior, stein;  // geometry and constants; X / Y / V are set per iteration inside the kernel
  RolloutArgs ra;         // LEAN rollout configuration; theta / noise / params are set per iteration inside the kernel
  UpdateArgs ua;
  PriorMerge pm;
  int n_iters, do_forward;
  int tiles, JS, n_pair_blocks, n_own_blocks;
  int sub_nt, per_block, lds_roll_floats;
  int mu_aliased;       // the prior means are the current particles (every tick after the first forward)
  int share_pair;       // mu_aliased and isotropic scales: prior tile and Stein tile share staging and distances (tick_pair_shared)
  float stein_ratio;    // (sigma_p / ell)^2
  int roll_strategy, weighted_prior;
  float *theta_buf0, *theta_buf1;  // buf0: theta at the start of the tick and, after forward's roll, at its end (two members, not an
                                   // array: a run-time index into the by-value argument block would send the whole block to scratch)
  const float *mu;      // prior means when they do not alias theta
  float x0[4];          // plant state (by value)
  const float *eps;     // device [n_iters][S][N][D] standard normals, or nullptr: Philox stream in registers
  size_t eps_stride;    // floats between the slices of two iterations
  const float *params;  // device [n_iters][M][P] dynamics samples or nullptr
  unsigned int *cnt_theta, *cnt_prior, *cnt_score, *cnt_stein, *cnt_lw;  // this tick's set: [tiles] lines each
  unsigned int *zero_base;  // the other set ...
  int zero_lines;           // ... of this many lines
  unsigned int *timeout_flag;
  // start barrier (round 3; as tick2.hpp's): every workgroup arrives on start_cnt (monotonic: start_target = seq x grid), workgroup 0 waits
  // a bounded time for all of them and publishes go (seq) or abort (seq | 2^31) in *go; nothing is written before go, so a launch whose
  // workgroups cannot all be resident leaves the state untouched and the host replays the tick on plain kernels.  nullptr: no barrier.
  unsigned int *start_cnt, *go, *abort_cnt;  // start_cnt: 16 lines of THIS tick's counter set (workgroup b arrives on line b % 16; re-armed
                                             // with the set by the next tick) - no running totals, no per-lane index into this block
  unsigned int seq;
  int test_abort;
  unsigned int expect_aborts;  // *abort_cnt as the host knew it at launch: a larger value = an earlier one-launch tick awaits its replay, abort too
  // forward outputs
  float *logp, *lw, *pw, *a_seq_out, *logmix, *mixw;
  int *istar;
  unsigned long long *tl;  // diagnostic build only: [grid][64] wall-clock stamps
};

// publish: every storing wave drains its write-through stores, barrier, one lane signals for the workgroup
__device__ __forceinline__ void arrive(unsigned int *line) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  wg_sync();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(line, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned int group_arrivals(const int g, const int N, const int per_block) {
  return (unsigned int)(min(PAIR_TI, N - g * PAIR_TI) / per_block);
}

// ---------------------------------------------------------------------------------------------------------------
// Prior tile + Stein tile of ONE (query tile, key slice) pair when the prior means alias theta (every tick after the first
// forward; svgd.py:87) and the prior / kernel scales are isotropic: both passes see the same particles, so the tiles are
// staged once (in x / sigma_p) and the pair distances are computed once; the Stein Gram value follows from the same
// distance with the ratio (sigma_p / ell)^2 folded into the exponent.  Arithmetic per element is that of pairwise_body<PRIOR>
// and stein_split_body up to the rounding of that rescaling (<= 2 ulp on the distance): the persistent tick then agrees with
// the launch-per-stage path to ~1e-6 instead of bit for bit (tests/test_gpu_parity.py checks both against the oracle).
// LDS: Xs[TI][DP] | Ys[JC][YS] | Vs[JC][YS] | kvP[TI][JC+1] | kvS[TI][JC+1] | mrow[TI]   (39.3 KB at CPT = 4)
template <int MODE, int CPT>
__device__ __forceinline__ void tick_pair_shared(const TickArgs &f, float *lds, const int tile_x, const int js, const float *th, const int k,
                                                 const LineGate &gate, const int tid, unsigned long long *tlp) {
  static_assert(MODE == PAIR_K1 || MODE == PAIR_IMQ, "Stein modes only");
  constexpr int TI = PAIR_TI, JC = PAIR_JC, NT = PAIR_NT;
  constexpr int DP = 8 * CPT, YS = DP + 4, QG = NT / JC, QPG = TI / QG, TPW = DP / 32;
  const PairArgs &a = f.prior;
  float *Xs = lds;
  float *Ys = Xs + TI * DP;
  float *Vs = Ys + JC * YS;
  float *kvP = Vs + JC * YS;
  float *kvS = kvP + TI * (JC + 1);
  float *mrow = kvS + TI * (JC + 1);
  const int D = a.D, da = a.da, N = a.N;
  const int ib = tile_x * TI;
  const int jbeg = js * a.slice, jend = min(N, jbeg + a.slice);
  const int nq = min(TI, a.n_local - ib), jc = min(JC, jend - jbeg);
  DUST_PRIO(DUST_PRIO_PRIOR);
  {
    float vx[RowLane<TI, DP, NT>::NB], vy[RowLane<JC, DP, NT>::NB];
    rowlane_issue<TI, DP, NT, true>(th, ib, nq, D, vx, tid);
    rowlane_issue<JC, DP, NT, true>(th, jbeg, jc, D, vy, tid);
    if (tid < TI) mrow[tid] = -INFINITY;
    rowlane_commit<TI, DP, DP, NT, true>(vx, nq, D, da, a.inv_s, Xs, tid);
    rowlane_commit<JC, DP, YS, NT, true>(vy, jc, D, da, a.inv_s, Ys, tid);
  }
  const int iB = tid >> 3, cB = (tid & 7) * CPT;
  v2f accA[CPT / 2], accB[CPT / 2];
#pragma unroll
  for (int c = 0; c < CPT / 2; ++c) accA[c] = accB[c] = v2f{0.f, 0.f};
  float accL = 0.f;
  const int mw = tid >> 6, ml = tid & 63, mqh = mw >> 1, mct0 = (mw & 1) * TPW;
  const int jA = tid & (JC - 1), igA = tid / JC;
  const float lm = a.logmix[jbeg + min(jA, jc - 1)];
  wg_sync();
  DUST_TLP(tlp, 3);
  v2f xB[CPT / 2];
#pragma unroll
  for (int c = 0; c < CPT / 2; ++c) xB[c] = *reinterpret_cast<const v2f *>(&Xs[iB * DP + cB + 2 * c]);
  {  // pass A (pairwise_body): lane = key, QPG queries per lane; both kernels' values from one distance
    v2f d2[QPG];
#pragma unroll
    for (int ii = 0; ii < QPG; ++ii) d2[ii] = v2f{0.f, 0.f};
#pragma unroll
    for (int d = 0; d < DP; d += 4) {
      const float4 yv = *reinterpret_cast<const float4 *>(&Ys[jA * YS + d]);
      const v2f y01 = {yv.x, yv.y}, y23 = {yv.z, yv.w};
#pragma unroll
      for (int ii = 0; ii < QPG; ++ii) {
        const float4 xv = *reinterpret_cast<const float4 *>(&Xs[(igA * QPG + ii) * DP + d]);
        const v2f z01 = v2f{xv.x, xv.y} - y01, z23 = v2f{xv.z, xv.w} - y23;
        d2[ii] = __builtin_elementwise_fma(z01, z01, d2[ii]);
        d2[ii] = __builtin_elementwise_fma(z23, z23, d2[ii]);
      }
    }
    const float rs = f.stein_ratio;  // (sigma_p / ell)^2
#pragma unroll
    for (int ii = 0; ii < QPG; ++ii) {
      const float dd = d2[ii].x + d2[ii].y;
      const float ds = dd * rs;
      kvP[(igA * QPG + ii) * (JC + 1) + jA] = (jA < jc) ? lm - 0.5f * dd : -INFINITY;
      float v;
      if (MODE == PAIR_K1) v = (jA < jc) ? __builtin_amdgcn_exp2f(-0.72134752044448170f * ds) : 0.f;
      else v = (jA < jc) ? __builtin_amdgcn_rsqf(1.0f + ds) : 0.f;
      kvS[(igA * QPG + ii) * (JC + 1) + jA] = v;
    }
  }
  wg_sync();
  DUST_TLP(tlp, 4);
  {  // prior softmax over the chunk (single chunk: no running rescale)
    float m = -INFINITY;
#pragma unroll
    for (int q = 0; q < JC / 8; ++q) m = fmaxf(m, kvP[iB * (JC + 1) + (tid & 7) + 8 * q]);
    m = oct_max(m);
    if ((tid & 7) == 0) mrow[iB] = m;
#pragma unroll
    for (int q = 0; q < JC / 8; ++q) {
      const int jj = (tid & 7) + 8 * q;
      const float l = kvP[iB * (JC + 1) + jj];
      kvP[iB * (JC + 1) + jj] = (m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((l - m) * 1.44269504088896340736f);
    }
    wg_sync();
  }
  DUST_TLP(tlp, 5);
  // pass B, prior half first (its partials are what the owners wait for)
#pragma unroll 4
  for (int jj = 0; jj < JC; ++jj) {
    const float kq = kvP[iB * (JC + 1) + jj];
    const v2f kk = {kq, kq};
#pragma unroll
    for (int c = 0; c < CPT; c += 4) {
      const float4 yv = *reinterpret_cast<const float4 *>(&Ys[jj * YS + cB + c]);
      const v2f y01 = {yv.x, yv.y}, y23 = {yv.z, yv.w};
      accA[c / 2] = __builtin_elementwise_fma(kk, y01 - xB[c / 2], accA[c / 2]);
      accA[c / 2 + 1] = __builtin_elementwise_fma(kk, y23 - xB[c / 2 + 1], accA[c / 2 + 1]);
    }
    accL += kq;
  }
  DUST_TLP(tlp, 6);
  const int il = tile_x * TI + iB;
  const float un = 1.0f / a.inv_s[0];  // isotropic scales (host check)
  if (il < a.n_local) {
    const size_t row = ((size_t)js * a.n_local + il) * DP;
#pragma unroll
    for (int c = 0; c < CPT; c += 4) {
      v4f oa;
#pragma unroll
      for (int q = 0; q < 4; ++q) oa[q] = (((c + q) & 1) ? accA[(c + q) / 2].y : accA[(c + q) / 2].x) * un;
      store16(a.pA + row + cB + c, oa, true);
    }
    if ((tid & 7) == 0) {
      st_sc1(a.pM + (size_t)js * a.n_local + il, mrow[iB]);
      st_sc1(a.pL + (size_t)js * a.n_local + il, accL);
    }
  }
  DUST_TLP(tlp, 7);
  arrive(f.cnt_prior + (size_t)tile_x * CNT_STRIDE);
  DUST_PRIO(0);
  DUST_TLK(f.tl, 16 * k + 1);
  // repulsive term (stein_split_body), in x / sigma_p coordinates: sum_j k'_ij (x_i - x_j)
#pragma unroll 4
  for (int jj = 0; jj < JC; ++jj) {
    const float kq = kvS[iB * (JC + 1) + jj];
    const float kp = MODE == PAIR_K1 ? -kq : -(kq * kq) * kq;
    const v2f kpp = {kp, kp};
#pragma unroll
    for (int c = 0; c < CPT; c += 4) {
      const float4 yv = *reinterpret_cast<const float4 *>(&Ys[jj * YS + cB + c]);
      const v2f y01 = {yv.x, yv.y}, y23 = {yv.z, yv.w};
      accB[c / 2] = __builtin_elementwise_fma(kpp, xB[c / 2] - y01, accB[c / 2]);
      accB[c / 2 + 1] = __builtin_elementwise_fma(kpp, xB[c / 2 + 1] - y23, accB[c / 2 + 1]);
    }
  }
  if (il < a.n_local) {
    const size_t row = ((size_t)js * a.n_local + il) * DP;
#pragma unroll
    for (int c = 0; c < CPT; c += 4) {
      v4f ob;
#pragma unroll
      for (int q = 0; q < 4; ++q) ob[q] = (((c + q) & 1) ? accB[(c + q) / 2].y : accB[(c + q) / 2].x) * un;
      store16(f.stein.pB + row + cB + c, ob, true);
    }
  }
  DUST_TLP(tlp, 10);
  // score rows of the key slice (owners, this iteration)
  if (tid < gate.nlines) spin_until(gate.cnt + (size_t)(gate.line0 + tid) * CNT_STRIDE, tid == 0 ? gate.target0 : gate.target1, f.timeout_flag);
  wg_sync();
  DUST_PRIO(3);
  DUST_TLP(tlp, 11);
  using RLV = RowLane<JC, DP, NT>;
  {
    float vv[RLV::NB];
    rowlane_issue<JC, DP, NT, true>(f.stein.V, jbeg, jc, D, vv, tid);
    rowlane_commit<JC, DP, YS, NT, false>(vv, jc, D, da, a.inv_s, Vs, tid);
  }
  wg_sync();
  v4f accM[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) accM[t] = v4f{0.f, 0.f, 0.f, 0.f};
  {
    const float *kb = kvS + (mqh * 16 + (ml & 15)) * (JC + 1) + (ml >> 4);
    const float *sb = Vs + (ml >> 4) * YS + mct0 * 16 + (ml & 15);
#pragma unroll
    for (int k4 = 0; k4 < JC / 4; ++k4) {
      const float bq = kb[4 * k4];
#pragma unroll
      for (int t = 0; t < TPW; ++t) accM[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(sb[4 * k4 * YS + 16 * t], bq, accM[t], 0, 0, 0);
    }
  }
  const int ilm = tile_x * TI + mqh * 16 + (ml & 15);
  if (ilm < a.n_local) {
    const size_t rowm = ((size_t)js * a.n_local + ilm) * DP;
#pragma unroll
    for (int t = 0; t < TPW; ++t) store16(f.stein.pA + rowm + (mct0 + t) * 16 + 4 * (ml >> 4), accM[t], true);
  }
  arrive(f.cnt_stein + (size_t)tile_x * CNT_STRIDE);
  DUST_PRIO(0);
  DUST_TLK(f.tl, 16 * k + 2);
}

// ---------------------------------------------------------------------------------------------------------------
// PAIR role: workgroup (tile_x, js) - prior tile, then Stein tile, per iteration; the log-density tile of forward at the end
template <int MODE, int CPT>
__device__ __forceinline__ void tick_pair(const TickArgs &f, float *lds, const int pb) {
  const int tile_x = pb % f.tiles, js = pb / f.tiles;
  const int N = f.prior.N;
  const int jbeg = js * f.prior.slice, jend = min(N, jbeg + f.prior.slice);
  const int g0 = jbeg / PAIR_TI, g1 = (jend - 1) / PAIR_TI;  // key groups (one chunk of <= 64 keys: one or two groups)
  const unsigned int aq = group_arrivals(tile_x, N, f.per_block), a0 = group_arrivals(g0, N, f.per_block), a1 = group_arrivals(g1, N, f.per_block);
  const int passes = f.n_iters + (f.do_forward ? 1 : 0);
  for (int k = 0; k < passes; ++k) {
    const float *th = (k & 1) ? f.theta_buf1 : f.theta_buf0;
    if (k > 0) {  // theta(k): rows of the query group and of the key group(s)
      if (threadIdx.x == 0) spin_until(f.cnt_theta + (size_t)tile_x * CNT_STRIDE, (unsigned int)k * aq, f.timeout_flag);
      if (threadIdx.x == 1) spin_until(f.cnt_theta + (size_t)g0 * CNT_STRIDE, (unsigned int)k * a0, f.timeout_flag);
      if (threadIdx.x == 2 && g1 != g0) spin_until(f.cnt_theta + (size_t)g1 * CNT_STRIDE, (unsigned int)k * a1, f.timeout_flag);
      wg_sync();
    }
    DUST_TLK(f.tl, 16 * k + 0);
    const float *ky = f.mu_aliased ? th : f.mu;
    const int tx = opaque((int)threadIdx.x);
#ifdef DUST_STAMPS
    unsigned long long *tlp = f.tl ? f.tl + 128 * blockIdx.x + 16 * k : nullptr;
#else
    unsigned long long *tlp = nullptr;
#endif
    if (k == f.n_iters) {  // SVMPC.forward: log p(theta) under the tick's prior (svmpc.py:137), no gradient
      pairwise_body<PAIR_LOGP, CPT, true>(f.prior, lds, tile_x, js, /*write_through=*/true, th, ky, tx, tlp);
      arrive(f.cnt_prior + (size_t)tile_x * CNT_STRIDE);
      DUST_TLK(f.tl, 16 * k + 1);
      break;
    }
    if (f.share_pair) {
      const LineGate gate{f.cnt_score, g0, g1 - g0 + 1, (unsigned int)(k + 1) * a0, (unsigned int)(k + 1) * a1};
      tick_pair_shared<MODE, CPT>(f, lds, tile_x, js, th, k, gate, tx, tlp);
      continue;
    }
    DUST_PRIO(DUST_PRIO_PRIOR);
    pairwise_body<PAIR_PRIOR, CPT, true>(f.prior, lds, tile_x, js, /*write_through=*/true, th, ky, tx, tlp);
    arrive(f.cnt_prior + (size_t)tile_x * CNT_STRIDE);
    DUST_PRIO(0);  // the theta-only half of the Stein tile is background work until the score rows arrive
    DUST_TLK(f.tl, 16 * k + 1);
    const LineGate gate{f.cnt_score, g0, g1 - g0 + 1, (unsigned int)(k + 1) * a0, (unsigned int)(k + 1) * a1};
    stein_split_body<MODE, CPT, true>(f.stein, lds, tile_x, js, nullptr, nullptr, f.timeout_flag, nullptr, &gate, th, opaque((int)threadIdx.x), tlp);
    arrive(f.cnt_stein + (size_t)tile_x * CNT_STRIDE);
    DUST_TLK(f.tl, 16 * k + 2);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// OWNER role: `per_block` particles per workgroup, `nt` lanes each (lane = action sample in the rollouts).  The arithmetic of
// every stage is that of rollout_body (LEAN form) / update_body / finalize_body / roll_kernel - same operations in the same
// order, so a tick equals the launch-per-stage path bit for bit (tests/test_gpu_parity.py).
template <int MODEL>
__device__ __forceinline__ void tick_owner(const TickArgs &f, float *lds_all, const int ob) {
  constexpr int DS = MODEL == DUST_MODEL_PENDULUM ? 2 : 4;
  constexpr int DA = MODEL == DUST_MODEL_PENDULUM ? 1 : 2;
  const RolloutArgs &a = f.ra;
  const int nt = f.sub_nt;
  const int sub = __builtin_amdgcn_readfirstlane((int)threadIdx.x / nt);  // nt is a multiple of 64: a wave belongs to one particle
  const int tid0 = (int)threadIdx.x - sub * nt;
  const int tid = tid0;
  float *lds = lds_all + (size_t)sub * f.lds_roll_floats;
  const int n0 = ob * f.per_block + sub;
  const int n = n0;
  const int S = a.S, D = a.D, H = a.H, N = a.N_total, Dp = D | 1, M = a.M;
  float *tile = lds;            // [S][Dp] noise, then actions
  float *cst = tile + (size_t)S * Dp;  // [S] costs -> weights
  float *omg = cst + S;         // [S] omega
  float *red = omg + S;         // [96]
  float *part = red + 96;       // [2][nt]
  float *th = part + 2 * nt + 2;  // [D] theta row of this particle
  float *coefs = th + D;        // [M][2]
  const int grp = (ob * f.per_block) / PAIR_TI;  // the workgroup's particles share one 32-particle group
  const unsigned int arr = group_arrivals(grp, N, f.per_block);
  const bool own = tid < D;
  const size_t o = (size_t)n * D + (own ? tid : 0);
  const UpdateArgs &ua = f.ua;
  const bool adam = ua.optimizer == DUST_OPT_ADAM;

  float thv = own ? f.theta_buf0[o] : 0.f;
  float amv = (own && a.update_a_mat) ? a.a_mat[o] : 0.f;
  float adm = 0.f, adv = 0.f;
  if (own && adam) {
    adm = ua.adam_m[o];
    adv = ua.adam_v[o];
  }
  const uint32_t ctr_tick = a.ctr[0], ctr_iter0 = a.ctr[1], adam0 = a.ctr[2];
  float x0[DS];
#pragma unroll
  for (int k = 0; k < DS; ++k) x0[k] = f.x0[k];
  float last_logl = (tid == 0) ? a.logl[n] : 0.f;  // forward without an iteration in this launch: the last sample's
  float *wred = lds_all + (red - lds);  // reduction scratch common to the whole workgroup (sub-block 0's)

  // noise of iteration k into the (free) tile, raw: standard normals.  Philox: a lane fills ITS row; caller-supplied eps: row-lane
  // staging (rollout_body), 12 loads in flight per lane.
  const int lgW = a.lgW, R = nt >> lgW;
  const int sr = tid >> lgW, sj = tid & ((1 << lgW) - 1);
  bool eps_bad = false;  // caller-supplied noise of the pending iteration holds a non-finite value (Philox never does)
  auto draw_noise = [&](const int k) {
    eps_bad = false;
    if (f.eps == nullptr) {
      for (int s = tid; s < S; s += nt) {
        float *row = tile + s * Dp;
        for (int j8 = 0; j8 * 8 < D; ++j8) {
          float z[8];
          philox_normal8(a.seed, (uint32_t)j8, (uint32_t)(s * N + n), ctr_iter0 + (uint32_t)k, ctr_tick, z);
#pragma unroll
          for (int q = 0; q < 8; ++q)
            if (j8 * 8 + q < D) row[j8 * 8 + q] = z[q];
        }
      }
    } else {
      constexpr int NB = 12;
      const float *nbase = f.eps + (size_t)k * f.eps_stride + (size_t)n * D + min(sj, D - 1);
      const size_t rowf = (size_t)N * D;
      for (int s0 = 0; s0 < S; s0 += NB * R) {
        float v[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) v[u] = nbase[(size_t)min(s0 + u * R + sr, S - 1) * rowf];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
          const int srow = s0 + u * R + sr;
          if (srow < S && sj < D) {
            tile[srow * Dp + sj] = v[u];
            eps_bad |= !(fabsf(v[u]) <= 3.0e38f);
          }
        }
      }
    }
  };
  draw_noise(0);

  for (int k = 0; k < f.n_iters; ++k) {
    DUST_TLK(f.tl, 16 * k + 0);
    const int tid = opaque(tid0);  // (these shadow the outer ones: see opaque())
    const int n = opaque_s(n0);
    const bool own = tid < D;
    const size_t o = (size_t)n * D + (own ? tid : 0);
    // ---- 1. actions = theta + L eps (likelihoods.py:81-101), in place ----
    for (int m = tid; m < M; m += nt) {
      if (a.coef_given) {
        coefs[2 * m] = a.coef_host[0];
        coefs[2 * m + 1] = a.coef_host[1];
      } else {
        const Coef c = make_coef(a.dm, f.params ? f.params + ((size_t)k * M + m) * a.dm.P : nullptr);
        coefs[2 * m] = c.c0;
        coefs[2 * m + 1] = c.c1;
      }
    }
    if (own) th[tid] = thv;
    if (tid == 0) red[40] = 0.f;
    wg_sync();
    // the tile keeps the raw noise; actions = theta + L eps (likelihoods.py:81-101) are formed where they are used (same two
    // operations, same rounding).  red[40]: some action may be NaN (non-finite theta / noise): the general loop, whose
    // clamps propagate NaN as torch.clamp does, instead of the v_med3 fast path
    if (eps_bad || (own && !(fabsf(thv) <= 3.0e38f))) red[40] = 1.f;
    wg_sync();
    DUST_TLK(f.tl, 16 * k + 1);

    // ---- 2. rollouts: lane = sample, dynamics samples looped in registers (rollout_body stage 2, LEAN, G == 1) ----
    DUST_PRIO(DUST_PRIO_OWNER);  // the owners' chain is the tick's critical path; background work fills the gaps
    const long SN = (long)S * N;
    const bool fast_trig = MODEL == DUST_MODEL_PENDULUM && red[40] == 0.f && fabsf(x0[1]) <= 3.0e38f &&
                           (fabsf(x0[0]) + a.dm.max_speed_pend * (float)a.dm.dt * (float)H < 5.0e4f);
    for (int s = tid; s < S; s += nt) {
      float *act = tile + s * Dp;
      double acc_m = 0.0;
      for (int m = 0; m < M; ++m) {
        const long r = (long)m * SN + (long)s * N + n;
        const int pidx = a.dm.interleave ? (int)(r % M) : m;
        Coef cf;
        cf.c0 = coefs[2 * pidx];
        cf.c1 = coefs[2 * pidx + 1];
        float x[DS];
#pragma unroll
        for (int q = 0; q < DS; ++q) x[q] = x0[q];
        CostSum<MODEL> tot;
        float traj;
        if (fast_trig && fabsf(cf.c0) <= 3.0e38f && fabsf(cf.c1) <= 3.0e38f) {
          const float dt = (float)a.dm.dt, mt = a.dm.max_torque, ms = a.dm.max_speed_pend, chol0 = a.chol_a[0];
          const v2f W = {a.dm.w_cos, a.dm.w_vel};
          float sn, cs;
#pragma unroll 4
          for (int t = 0; t < H; ++t) {
            pendulum_trig(x[0], &sn, &cs);
            v2f q = {cs - 1.0f, x[1]};
            q = W * (q * q);
            tot.add(q.x + q.y, t);
            const float u = __builtin_amdgcn_fmed3f(th[t] + chol0 * act[t], -mt, mt);
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
          for (int t = 0; t < H; ++t) {
            float at[DA];
#pragma unroll
            for (int q = 0; q < DA; ++q) at[q] = th[t * DA + q] + a.chol_a[q] * act[t * DA + q];
            const float ci = step_with_cost<MODEL>(a.dm, cf, x, at);
            tot.add(ci, t);
          }
          traj = (float)tot.total() + term_cost<MODEL>(a.dm, x);
        }
        acc_m += (double)traj;
      }
      const float cost = (M == 1) ? (float)acc_m : (float)(acc_m / M);
      cst[s] = cost;
      a.costsT[(size_t)n * S + s] = cost;
    }
    wg_sync();
    DUST_TLK(f.tl, 16 * k + 2);

    // ---- 3. softmax over samples (rollout_body stage 3) ----
    float cmin = INFINITY, csum = 0.f;
    for (int s = tid; s < S; s += nt) {
      cmin = fminf(cmin, cst[s]);
      csum += cst[s];
    }
    const int lane = tid & 63, wid = tid >> 6, nw = (nt + 63) >> 6;
    {
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
    const bool same_w = (a.alpha * a.temp == 1.0f);
    float zw = 0.f, zo = 0.f;
    for (int s = tid; s < S; s += nt) {
      const float c = cst[s];
      const float ew = expf(-c * a.alpha - (-cmin * a.alpha));
      cst[s] = ew;
      zw += ew;
      if (!same_w) {
        const float lo = (-1.0f * (c - cmin)) / a.temp;
        const float eo = expf(lo);
        omg[s] = eo;
        zo += eo;
      }
    }
    {
      zw = wave_sum(zw);
      zo = wave_sum(zo);
      if (lane == 0) {
        red[16 + wid] = zw;
        red[24 + wid] = zo;
      }
      lds_barrier();
      zw = red[16];
      zo = red[24];
      for (int w = 1; w < nw; ++w) {
        zw += red[16 + w];
        zo += red[24 + w];
      }
      if (same_w) zo = zw;
    }
    DUST_TLK(f.tl, 16 * k + 8);
    float eta_n = 0.f;
    if (tid == 0) {
      if (a.lik == DUST_LIK_EXP_UTILITY) last_logl = ((-cmin * a.alpha) + logf(zw)) - logf((float)S);
      else last_logl = -a.alpha * (csum / (float)S);
      eta_n = (-cmin / a.temp) + logf(zo);
    }
    // ---- 4. weighted reductions over s (rollout_body stage 4) ----
    const int Q = nt / D > 0 ? nt / D : 1;
    float g = 0.f, am = 0.f;
    const int q = (int)__umulhi((uint32_t)tid, a.magicD), j = tid - q * D;
    if (q < Q) {
      const float thj = th[j], lj = pick_da<DA>(a.chol_a, j);
      const float is2 = 1.0f / (pick_da<DA>(a.sigma_a, j) * pick_da<DA>(a.sigma_a, j));
      const float *tp = tile + j;
      int so = q, to = q * Dp;
      const int tstep = Q * Dp, tmax = (S - 1) * Dp;
      if (same_w && a.eps_base_mode) {
#pragma unroll 4
        for (int it = 0; it < a.wq_iters; ++it) {
          const float av = thj + lj * tp[min(to, tmax)], wl = cst[min(so, S - 1)];
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
          const float av = thj + lj * tp[min(to, tmax)], wl = cst[sc], wol = op[sc];
          const float w = so < S ? wl : 0.f, wo = so < S ? wol : 0.f;
          g = fmaf(w, (av - thj) * is2, g);
          am = fmaf(wo, av - base, am);
          so += Q;
          to += tstep;
        }
      }
    }
    DUST_TLK(f.tl, 16 * k + 9);
    // ---- prior partials of this group's query tile (pair role, this iteration) ----
    if (threadIdx.x == 0) spin_until(f.cnt_prior + (size_t)grp * CNT_STRIDE, (unsigned int)(k + 1) * (unsigned int)f.JS, f.timeout_flag);
    wg_sync();
    DUST_TLK(f.tl, 16 * k + 3);

    // prior partials of this row (write-through by the pair role: sc1 loads), issued as soon as the tile has arrived
    float pmA[16], pmM1 = -INFINITY, pmL1 = 0.f;
    const int JS = f.JS;
    if (tid < 16) {
      const size_t rowi = (size_t)min(tid, JS - 1) * f.pm.n_local + n;
      pmM1 = ld_sc1(f.pm.pM + rowi);
      pmL1 = ld_sc1(f.pm.pL + rowi);
    }
    if (own) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const size_t rowi = (size_t)(u < JS ? u : 0) * f.pm.n_local + n;
        pmA[u] = ld_sc1(f.pm.pA + rowi * f.pm.ldp + tid);
      }
    }
    part[tid] = g;
    part[nt + tid] = am;
    if (tid < 16) {
      if (tid >= JS) {
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
    DUST_TLK(f.tl, 16 * k + 10);
    float gs_keep = 0.f, gp_keep = 0.f;
    if (own) {
      float gs = 0.f, as = 0.f;
      for (int qq = 0; qq < Q; ++qq) {
        gs += part[qq * D + tid];
        as += part[nt + qq * D + tid];
      }
      gs = gs / zw;
      if (same_w && a.eps_base_mode) as = gs * (pick_da<DA>(a.sigma_a, tid) * pick_da<DA>(a.sigma_a, tid));
      else as = as / zo;
      if (a.update_a_mat) amv = amv + as;
      float acc = 0.f;
#pragma unroll
      for (int u = 0; u < 16; ++u) acc = fmaf(pmA[u], red[64 + u], acc);
      gp_keep = (acc / red[80]) * pick_da<DA>(f.pm.inv_s2, tid);
      gs_keep = gs;
      st_sc1(a.score + o, gs + gp_keep);
    }
    DUST_TLK(f.tl, 16 * k + 11);
    arrive(f.cnt_score + (size_t)grp * CNT_STRIDE);  // (the drain inside waits for the score rows only: the record stores follow)
    DUST_PRIO(0);
    if (own) {
      a.grad_lik[o] = gs_keep;
      a.grad_pri[o] = gp_keep;
      if (a.update_a_mat) a.a_mat[o] = amv;
    }
    if (tid == 0) {
      a.logl[n] = last_logl;
      a.eta[n] = eta_n;
    }
    DUST_TLK(f.tl, 16 * k + 4);

    // ---- next iteration's noise, underneath the wait for the Stein partials (the tile is free now) ----
    if (k + 1 < f.n_iters) draw_noise(k + 1);
    DUST_TLK(f.tl, 16 * k + 5);

    // ---- Stein partials of this group's query tile -> phi -> optimiser step (update_body) ----
    if (threadIdx.x == 0) spin_until(f.cnt_stein + (size_t)grp * CNT_STRIDE, (unsigned int)(k + 1) * (unsigned int)f.JS, f.timeout_flag);
    wg_sync();
    DUST_PRIO(DUST_PRIO_OWNER);
    DUST_TLK(f.tl, 16 * k + 6);
    if (own) {
      float sa = 0.f, sb = 0.f;
      for (int q0 = 0; q0 < JS; q0 += 16) {
        float va[16], vb[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const size_t p = ((size_t)min(q0 + u, JS - 1) * ua.n_local + n) * ua.ldp + tid;
          va[u] = ld_sc1(ua.pA + p);
          vb[u] = ld_sc1(ua.pB + p);
        }
#pragma unroll
        for (int u = 0; u < 16; ++u)
          if (q0 + u < JS) {
            sa += va[u];
            sb += vb[u];
          }
      }
      DUST_TLK(f.tl, 16 * k + 12);
      const float phi = sb * ua.inv_l2 + sa * ua.inv_n;
      ua.phi[o] = phi;
      const float gr = -phi;
      if (!adam) {
        thv = fmaf(-ua.lr, gr, thv);
      } else {
        thv = adam_step(thv, gr, adm, adv, ua.lr, ua.beta1, ua.beta2, ua.eps, (float)(adam0 + (uint32_t)k + 1u));
      }
      st_sc1(((k + 1) & 1 ? f.theta_buf1 : f.theta_buf0) + o, thv);
    }
    arrive(f.cnt_theta + (size_t)grp * CNT_STRIDE);
    DUST_TLK(f.tl, 16 * k + 7);
  }

  const int kf = f.n_iters;
  if (!f.do_forward) {  // SVMPC.optimize alone: the optimiser state stays (svmpc.py:97-126)
    if (own && adam) {
      ua.adam_m[o] = adm;
      ua.adam_v[o] = adv;
    }
    if (ob == 0 && threadIdx.x == 0) {
      a.ctr[1] = ctr_iter0 + (uint32_t)kf;
      a.ctr[2] = adam0 + (uint32_t)kf;
    }
    return;
  }

  // ---- SVMPC.forward (svmpc.py:172-200), fast_pred: the last iteration's costs ----
  if (own) th[tid] = thv;
  if (threadIdx.x == 0) spin_until(f.cnt_prior + (size_t)grp * CNT_STRIDE, (unsigned int)(kf + 1) * (unsigned int)f.JS, f.timeout_flag);
  wg_sync();
  DUST_TLK(f.tl, 16 * kf + 0);
  if (tid == 0) {  // log p(theta_n) from the slice partials (finalize_body), log_w = log_l + log_p (svmpc.py:137-138)
    float pmx, pl;
    prior_merge_row<true>(f.pm, n, &pmx, &pl);
    const float lp = (pmx + logf(pl)) + f.pm.log_norm;
    f.logp[n] = lp;
    st_sc1(f.lw + n, last_logl + lp);
  }
  // two-level arrival: the last owner workgroup of a group bumps the global line (cnt_lw[tiles]); every owner polls that one
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  wg_sync();
  if (threadIdx.x == 0) {
    const unsigned int prev = __hip_atomic_fetch_add(f.cnt_lw + (size_t)grp * CNT_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev + 1u == arr) __hip_atomic_fetch_add(f.cnt_lw + (size_t)f.tiles * CNT_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    spin_until(f.cnt_lw + (size_t)f.tiles * CNT_STRIDE, (unsigned int)f.tiles, f.timeout_flag);
  }
  wg_sync();
  DUST_TLK(f.tl, 16 * kf + 1);
  // softmax over all particles, first-index argmax (finalize_body), computed by every owner workgroup for itself
  {
    constexpr int RR = 16;  // N <= 16 * 256 on this path (host check)
    const int t = (int)threadIdx.x;
    float lwr[RR];
    float m = -INFINITY;
#pragma unroll
    for (int r = 0; r < RR; ++r) {
      const int i = t + r * PAIR_NT;
      lwr[r] = i < N ? ld_sc1(f.lw + i) : -INFINITY;
      m = fmaxf(m, lwr[r]);
    }
    m = block_reduce<RED_MAX>(m, wred);
    float z = 0.f;
#pragma unroll
    for (int r = 0; r < RR; ++r) {
      const int i = t + r * PAIR_NT;
      if (i < N) z += expf(lwr[r] - m);
    }
    z = block_reduce<RED_SUM>(z, wred);
    const float lz = m + logf(z);
    float best = -INFINITY, psum = 0.f;
    int bi = 0x7fffffff;
    const int n_first = ob * f.per_block;
#pragma unroll
    for (int r = 0; r < RR; ++r) {
      const int i = t + r * PAIR_NT;
      if (i >= N) continue;
      const float p = expf(lwr[r] - lz);
      if (i >= n_first && i < n_first + f.per_block) f.pw[i] = p;
      lwr[r] = p;
      psum += p;
      if (p > best) {
        best = p;
        bi = i;
      }
    }
    const int lane2 = t & 63, wid2 = t >> 6;
    for (int of = 32; of > 0; of >>= 1) {
      const float obv = __shfl_xor(best, of, 64);
      const int oi = __shfl_xor(bi, of, 64);
      if (obv > best || (obv == best && oi < bi)) {
        best = obv;
        bi = oi;
      }
    }
    wg_sync();
    int *redi = reinterpret_cast<int *>(wred + 32);
    if (lane2 == 0) {
      wred[wid2] = best;
      redi[wid2] = bi;
    }
    wg_sync();
    best = wred[0];
    bi = redi[0];
    for (int w = 1; w < PAIR_NT / 64; ++w)
      if (wred[w] > best || (wred[w] == best && redi[w] < bi)) {
        best = wred[w];
        bi = redi[w];
      }
    wg_sync();
    if (bi >= n_first && bi < n_first + f.per_block) {  // the owner of the best particle hands out its action sequence
      const float *thb = lds_all + (size_t)(bi - n_first) * f.lds_roll_floats + (th - lds);
      if (t == 0) *f.istar = bi;
      for (int d = t; d < D; d += PAIR_NT) f.a_seq_out[d] = thb[d];
    }
    // new prior mixture (finalize_body): Categorical(probs) clamps, then log_softmax
    if (!f.weighted_prior) {
      const float l = logf(fminf(fmaxf(1.0f / (float)N, 1.1920929e-07f), 1.0f - 1.1920929e-07f));
      const float lzz = l + logf((float)N);
      if (t < f.per_block) {
        f.mixw[n_first + t] = 1.0f;
        f.logmix[n_first + t] = l - lzz;
      }
    } else {
      psum = block_reduce<RED_SUM>(psum, wred);
      float lm = -INFINITY;
#pragma unroll
      for (int r = 0; r < RR; ++r) {
        const int i = t + r * PAIR_NT;
        if (i >= N) continue;
        const float w = lwr[r];
        if (i >= n_first && i < n_first + f.per_block) f.mixw[i] = w;
        float p = w / psum;
        p = fminf(fmaxf(p, 1.1920929e-07f), 1.0f - 1.1920929e-07f);
        const float l = logf(p);
        lwr[r] = l;
        lm = fmaxf(lm, l);
      }
      lm = block_reduce<RED_MAX>(lm, wred);
      float zs = 0.f;
#pragma unroll
      for (int r = 0; r < RR; ++r) {
        const int i = t + r * PAIR_NT;
        if (i < N) zs += expf(lwr[r] - lm);
      }
      zs = block_reduce<RED_SUM>(zs, wred);
      const float lzz = lm + logf(zs);
#pragma unroll
      for (int r = 0; r < RR; ++r) {
        const int i = t + r * PAIR_NT;
        if (i >= n_first && i < n_first + f.per_block) f.logmix[i] = lwr[r] - lzz;
      }
    }
    wg_sync();
  }
  // roll (svmpc.py:142-158): shift left along H, last row per strategy; into the home buffer.  Every reader of theta(kf) in
  // either buffer has finished: the log-weights of ALL particles needed every log-density tile.
  float outv = 0.f;
  if (f.roll_strategy == DUST_ROLL_MEAN) {
    // roll_kernel: block_reduce over a 128-lane block = wave sums of lanes [0, 64) and [64, 128), added
    for (int c = 0; c < DA; ++c) {
      const float v = (own && tid % DA == c) ? thv : 0.f;
      const float ws = wave_sum(v);
      wg_sync();
      if ((tid & 63) == 0 && (tid >> 6) < 2) part[tid >> 6] = ws;
      wg_sync();
      const float s = nt >= 128 ? part[0] + part[1] : part[0];
      if (own && tid + DA >= D && tid % DA == c) outv = s / (float)H;
    }
  }
  if (own) {
    float out = (tid + DA < D) ? th[tid + DA] : thv;
    if (f.roll_strategy == DUST_ROLL_MEAN && tid + DA >= D) out = outv;
    f.theta_buf0[o] = out;
    if (adam) {  // SVMPC.roll makes a NEW parameter tensor: torch's optimiser state (exp_avg, exp_avg_sq, step) restarts
      ua.adam_m[o] = 0.f;
      ua.adam_v[o] = 0.f;
    }
  }
  if (ob == 0 && threadIdx.x == 0) {  // next tick: new Philox sub-stream (every owner read the counters before its first wait)
    a.ctr[0] = ctr_tick + 1u;
    a.ctr[1] = 0u;
    a.ctr[2] = 0u;
  }
  DUST_TLK(f.tl, 16 * kf + 2);
}

template <int MODEL, int MODE, int CPT>
__global__ __launch_bounds__(PAIR_NT, (CPT <= 4 ? 4 : 2)) void svmpc_tick_kernel(const TickArgs f) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int b0 = (int)blockIdx.x;
#ifdef DUST_STAMPS
  if (f.tl && threadIdx.x == 0) {  // placement census: raw HW_ID (wave / simd / cu / sh / se fields) and XCC_ID of every workgroup
    f.tl[128 * blockIdx.x + 126] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    f.tl[128 * blockIdx.x + 127] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);
  }
#endif
  // A hand-off wait of an EARLIER launch timed out and the host has not seen it yet (open-loop callers read no outputs): every wait
  // of this launch would leave after 256 spins and compute on stale data.  Do nothing instead - the particles, the optimiser state
  // and the noise counters stay where the failed tick left them; the host reports the error at its next synchronisation and clears
  // the flag (dust_sync / tick_outputs).  (The flag is only ever set inside a launch: uniform for this one.)
  if (__hip_atomic_load(f.timeout_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
  if (b0 == 0)
    for (int t = threadIdx.x; t < f.zero_lines; t += PAIR_NT) f.zero_base[t * CNT_STRIDE] = 0u;
  if (f.start_cnt) {  // residency proof (see TickArgs): the other counter set is re-armed above whether or not the tick starts
    // (arrivals are spread over 16 lines - 1 024 same-address atomics take ~9 us - and the waiters poll `go` sparsely: a thousand lanes
    //  polling one line every 60 ns starve the arrivals on their way to the same L2)
    unsigned int &s_go = *reinterpret_cast<unsigned int *>(lds);  // (the roles stage into the dynamic region only after this barrier)
    if (threadIdx.x < 64) {
      const int lane = threadIdx.x;
      if (lane == 0) __hip_atomic_fetch_add(f.start_cnt + (size_t)(b0 & 15) * CNT_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned int g = 0u;
      if (b0 == 0) {
        bool ok = true;
        if (lane < 16) {
          const unsigned int G = gridDim.x, mine = G / 16u + ((unsigned int)lane < G % 16u ? 1u : 0u);
          const unsigned int target = mine;
          const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
          while ((int)(__hip_atomic_load(f.start_cnt + (size_t)lane * CNT_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
            __builtin_amdgcn_s_sleep(2);
            if (__builtin_amdgcn_s_memrealtime() - t_start > 20000ull) {  // 200 us
              ok = false;
              break;
            }
          }
        }
        const bool all_ok = __all(ok ? 1 : 0) && !f.test_abort &&
                            __hip_atomic_load(f.abort_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == f.expect_aborts;
        g = all_ok ? f.seq : (f.seq | 0x80000000u);
        if (lane == 0) {
          if (!all_ok) __hip_atomic_fetch_add(f.abort_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(f.go, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      } else if (lane == 0) {
        unsigned int spins = 0u;
        unsigned long long t0 = 0ull;
        for (;;) {
          g = __hip_atomic_load(f.go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((g & 0x7fffffffu) == f.seq) break;
          __builtin_amdgcn_s_sleep(12);
          if ((++spins & 63u) == 0u) {
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if (!t0) t0 = now;
            else if (now - t0 > DUST_SPIN_TIMEOUT_TICKS) {  // workgroup 0 never came: give up (reported as a time-out)
              __hip_atomic_store(f.timeout_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              g = f.seq | 0x80000000u;
              break;
            }
          }
        }
      }
      if (lane == 0) s_go = g == f.seq ? 1u : 0u;
    }
    wg_sync();
    if (s_go == 0u) return;
  }
  if (b0 < f.n_pair_blocks) {
#ifndef DUST_X_NOPAIR
    tick_pair<MODE, CPT>(f, lds, b0);
#endif
  } else {
#ifndef DUST_X_NOOWN
    const int br = b0 - f.n_pair_blocks;
    const int ob = ((f.n_pair_blocks | f.n_own_blocks) & 7) ? br : xcd_contiguous(br, f.n_own_blocks);
    tick_owner<MODEL>(f, lds, ob);
#endif
  }
}

}  // namespace dust
