// stein.hpp - pairwise (N x N x D) passes: GMM prior score / log-density, Stein kernel Gram + phi, optimiser update.
//
// Replaces (reference file:line): the prior half of SVMPC.phi svmpc.py:38-41 (autograd through
// MixtureSameFamily.log_prob -> closed form sum_k r_ik (mu_k - x_i)/sigma^2), prior.log_prob in SVMPC.get_weights
// svmpc.py:128-140, the K1 kernel branch svmpc.py:76-83 (gpytorch RBFKernel semantics, lengthscale ln 2), the new IMQ
// kernel, and SVMPC.step's optimiser update svmpc.py:87-95.
//
// Mapping (MI355X).  The pairwise work is cut into 2-D tiles so that a launch has >= 256 workgroups even at N = 1024
// and every CU reads only ITS tile of the particle set (not all of it): workgroup (it, js) owns TI = 32 query particles
// and the js-th slice of the key particles, streamed through LDS in chunks of JC = 64 rows (row-lane staging: a lane keeps
// one column and walks the rows, every load of a tile in flight before the first wait).  Per chunk:
//   pass A  lane = key j, the wave walks 8 queries: d2 = sum_d ((x_i[d] - y_j[d]) / s_d)^2 with y_j in registers and
//           x_i[d] LDS-broadcast (b128); kernel value / softmax logit -> LDS kv[TI][JC+1];
//   pass B  lane = (query i, column group): acc[i][c] += kv[i][j] * V[j][c], a [32 x 64] . [64 x D] product out of
//           LDS with b128 reads, differences (x_i - y_j) formed BEFORE the multiply (collapsed particle sets must not
//           cancel catastrophically).
// Each workgroup writes a partial row block; the js partials are combined by the NEXT kernel in the chain (prior ->
// rollout_kernel, Stein -> update_kernel) in a fixed order, so results are bitwise reproducible (no float atomics).
// Nothing N x N ever reaches HBM (the reference materialises [N,N,H,da] through autograd).
#pragma once
#include "common.hpp"
#include "handoff.hpp"

namespace dust {

enum { PAIR_TI = 32, PAIR_JC = 64, PAIR_NT = 256 };
// a word no arithmetic produces (hardware NaNs are the canonical quiet NaN, inputs never carry this payload): "not written yet"
static constexpr unsigned int SCORE_SENTINEL = 0xFFFFFFFFu;

struct PairArgs {
  int N, D, da, H;
  int i0, n_local;     // query rows [i0, i0 + n_local)
  int JS;              // number of key slices (gridDim.y)
  int slice;           // keys per slice (multiple of PAIR_JC except possibly the last)
  uint32_t magicD;     // floor(2^32 / D) + 1
  const float *X;      // [N][D] queries (theta)
  const float *Y;      // [N][D] keys (mu for the prior, theta for Stein), row-major
  const float *V;      // [N][D] score (Stein only)
  const float *logmix; // [N] prior mixture log-weights
  float inv_s[4];      // 1/sigma_p[d % da] (prior) or 1/ell (Stein)
  // partial outputs, indexed [js][i_local]
  float *pA;           // [JS][n_local][DP] prior: sum_k p (mu - x)      Stein: sum_j k s_j
  float *pB;           // [JS][n_local][DP]                              Stein: sum_j k' (x_i - x_j)
  float *pM;           // [JS][n_local]     prior: slice max of the logits
  float *pL;           // [JS][n_local]     prior: sum exp(logit - max)
  unsigned long long *stamps;  // diagnostic build only
};

// 16-byte global store; `wt` = write-through to memory (sc1), readable by sc1 loads from any CU of the device in the same launch
__device__ __forceinline__ void store16(float *p, v4f v, bool wt) {
  // s_nop AFTER the store: a > 8-byte VMEM store reads its data registers late, and a VALU write of them in the next
  // instruction needs a wait state that the compiler's hazard recogniser inserts for its own stores but not around
  // inline asm (without it the register allocator's immediate reuse of v[8:9] zeroed two columns at CPT = 8)
  if (wt) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else *reinterpret_cast<v4f *>(p) = v;
}

// Row-lane tile staging: W = 2^k >= DP lanes per row, RB = NT / W rows per batch; a lane keeps ONE column (its scale is a
// per-lane constant, no index division) and walks the rows with a fixed stride in HBM and LDS.  Rows / columns outside the
// source are written as zeros, so no separate zero-fill pass (and no barrier for it) is needed.  Loads are clamped, never
// predicated.  issue and commit are separate: several tiles' loads are put in flight before the first wait.
template <int TR, int DP, int NT>
struct RowLane {
  static constexpr int W = DP <= 32 ? 32 : (DP <= 64 ? 64 : 128);
  static constexpr int RB = NT / W;
  static constexpr int NB = TR / RB;
};
// SC1: the rows were written (write-through) by other workgroups of the SAME launch: every load bypasses this CU's L1
template <int TR, int DP, int NT, bool SC1 = false>
__device__ __forceinline__ void rowlane_issue(const float *__restrict__ src, int r0, int nrows, int D, float (&v)[RowLane<TR, DP, NT>::NB],
                                              const int tx = (int)threadIdx.x) {
  using RL = RowLane<TR, DP, NT>;
  const int lr = tx / RL::W, lc = min(tx % RL::W, D - 1);
  const float *base = src + (size_t)r0 * D + lc;
#pragma unroll
  for (int u = 0; u < RL::NB; ++u) {
    const float *p = base + (size_t)min(u * RL::RB + lr, nrows - 1) * D;
    v[u] = SC1 ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
  }
}
template <int TR, int DP, int LS, int NT, bool SCALE>
__device__ __forceinline__ void rowlane_commit(const float (&v)[RowLane<TR, DP, NT>::NB], int nrows, int D, int da, const float *colscale,
                                               float *dst, const int tx = (int)threadIdx.x) {
  using RL = RowLane<TR, DP, NT>;
  const int lr = tx / RL::W, lc = tx % RL::W;
  if (lc >= DP) return;  // W > DP (DP = 96): idle lanes
  const float sc = SCALE ? colscale[da == 1 ? 0 : (da == 2 ? (lc & 1) : lc % da)] : 1.0f;
  const bool cv = lc < D;
#pragma unroll
  for (int u = 0; u < RL::NB; ++u) {
    const int r = u * RL::RB + lr;
    dst[r * LS + lc] = (cv && r < nrows) ? (SCALE ? v[u] * sc : v[u]) : 0.f;
  }
}

template <int MODE, int CPT /* columns per lane in pass B: multiple of 4, 8*CPT >= D */, bool SC1 = false /* X / Y were written inside this launch */>
__device__ __forceinline__ void pairwise_body(const PairArgs &a, float *lds, const int tile_x, const int js, const bool write_through = false,
                                              const float *Xo = nullptr, const float *Yo = nullptr /* override a.X / a.Y (persistent tick) */,
                                              const int tx = (int)threadIdx.x /* lane index; the persistent tick passes an opaque copy per
                                              iteration so that the lane-derived addresses of every body are not all hoisted out of its loop */,
                                              unsigned long long *tlp = nullptr /* diagnostic build: timeline slots */) {
  constexpr int TI = PAIR_TI, JC = PAIR_JC, NT = PAIR_NT;
  constexpr bool PRI = MODE == PAIR_PRIOR || MODE == PAIR_LOGP;  // softmax-weighted passes over the prior mixture
  constexpr bool LOGP = MODE == PAIR_LOGP;                        // ... of which forward needs only the log-density
  constexpr int DP = 8 * CPT;  // padded row length in LDS (multiple of 4 -> b128 reads)
  constexpr int YS = DP + 4;
  constexpr int QG = NT / JC;      // query groups in pass A (4)
  constexpr int QPG = TI / QG;     // queries per group (8)
  float *Xs = lds;                  // [TI][DP]   queries, pre-scaled by 1/s_d
  float *Ys = Xs + TI * DP;         // [JC][YS]   keys, pre-scaled by 1/s_d
  float *Vs = Ys + JC * YS;         // [JC][YS]   score (Stein), unscaled
  float *kv = Vs + (PRI ? 0 : JC * YS);  // [TI][JC + 1]
  float *mrow = kv + TI * (JC + 1); // [TI] running max
  const int tid = tx;
  const int D = a.D, da = a.da, N = a.N;
  const int ib = a.i0 + tile_x * TI;  // first query (global index)
  const int jbeg = js * a.slice, jend = min(N, jbeg + a.slice);

  DUST_STAMP(a.stamps, 0);
  // ---- query tile + first key chunk -> LDS: all first-batch loads are in flight before the first wait ----
  const int nq = min(TI, a.i0 + a.n_local - ib), jc0 = min(JC, jend - jbeg);
  {
    float vx[RowLane<TI, DP, NT>::NB], vy[RowLane<JC, DP, NT>::NB], vv[RowLane<JC, DP, NT>::NB];
    rowlane_issue<TI, DP, NT, SC1>(Xo ? Xo : a.X, ib, nq, D, vx, tid);
    rowlane_issue<JC, DP, NT, SC1>(Yo ? Yo : a.Y, jbeg, jc0, D, vy, tid);
    if (!PRI) rowlane_issue<JC, DP, NT, SC1>(a.V, jbeg, jc0, D, vv, tid);
    if (tid < TI) mrow[tid] = -INFINITY;
    rowlane_commit<TI, DP, DP, NT, true>(vx, nq, D, da, a.inv_s, Xs, tid);
    rowlane_commit<JC, DP, YS, NT, true>(vy, jc0, D, da, a.inv_s, Ys, tid);
    if (!PRI) rowlane_commit<JC, DP, YS, NT, false>(vv, jc0, D, da, a.inv_s, Vs, tid);
  }

  // pass-B ownership: query iB, columns [cB, cB + CPT)
  const int iB = tid >> 3, cB = (tid & 7) * CPT;
  v2f accA[CPT / 2], accB[CPT / 2];
#pragma unroll
  for (int c = 0; c < CPT / 2; ++c) accA[c] = accB[c] = v2f{0.f, 0.f};
  float accL = 0.f;  // prior: sum of weights (same in the 8 lanes of a query)
  // Stein modes, D <= 64: the Gram x score product (sum_j k_ij s_j: a [32 x 64].[64 x DP] GEMM per chunk) runs on the matrix
  // cores - v_mfma_f32_16x16x4_f32, fp32 in and out - computed transposed (D'[col][query]) so that a lane ends up with 4
  // CONSECUTIVE columns of one query, i.e. one 16-byte store of the partial row.  Wave w owns query half w >> 1 and
  // TPW = DP / 32 column tiles.  The repulsive term keeps exact differences on the VALU and overlaps with the MFMAs.
  constexpr bool MFMA_A = !PRI && CPT <= 8;
  constexpr int TPW = MFMA_A ? DP / 32 : 1;
  v4f accM[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) accM[t] = v4f{0.f, 0.f, 0.f, 0.f};
  const int mw = tid >> 6, ml = tid & 63, mqh = mw >> 1, mct0 = (mw & 1) * TPW;  // MFMA ownership
  wg_sync();
  v2f xB[CPT / 2];
#pragma unroll
  for (int c = 0; c < CPT / 2; ++c) xB[c] = *reinterpret_cast<const v2f *>(&Xs[iB * DP + cB + 2 * c]);

  for (int j0 = jbeg; j0 < jend; j0 += JC) {
    const int jc = min(JC, jend - j0);
    DUST_STAMP(a.stamps, 1);
    // ---- key chunk -> LDS: rows j0..j0+jc-1 are contiguous in HBM ----
    const int jA = tid & (JC - 1), igA = tid / JC;
    const float lm = (PRI) ? a.logmix[j0 + min(jA, jc - 1)] : 0.f;  // issued with the tile loads
    if (j0 != jbeg) {  // later chunks of a long slice (the first one was staged with the query tile)
      float vy[RowLane<JC, DP, NT>::NB], vv[RowLane<JC, DP, NT>::NB];
      rowlane_issue<JC, DP, NT, SC1>(Yo ? Yo : a.Y, j0, jc, D, vy, tid);
      if (!PRI) rowlane_issue<JC, DP, NT, SC1>(a.V, j0, jc, D, vv, tid);
      rowlane_commit<JC, DP, YS, NT, true>(vy, jc, D, da, a.inv_s, Ys, tid);  // the barrier that ended the previous chunk's pass B
      if (!PRI) rowlane_commit<JC, DP, YS, NT, false>(vv, jc, D, da, a.inv_s, Vs, tid);  // protects these writes
      wg_sync();
    }
    DUST_STAMP(a.stamps, 2);
    DUST_TLP(tlp, 3);
    // ---- pass A: lane = key j, QPG queries per lane; packed math: 2 dims per v_pk_add / v_pk_fma ----
    {
      // packed math: 2 dims per v_pk_add / v_pk_fma.  (The Gram value uses the bare v_exp_f32: inlining ocml expf eight
      // times here made hipcc (ROCm 7.2) allocate 256 VGPRs + scratch and pass A ran 2-8x slower.)
      v2f d2[QPG];
#pragma unroll
      for (int ii = 0; ii < QPG; ++ii) d2[ii] = v2f{0.f, 0.f};
#pragma unroll
      for (int d = 0; d < DP; d += 4) {
        const float4 yv = *reinterpret_cast<const float4 *>(&Ys[jA * YS + d]);
        const v2f y01 = {yv.x, yv.y}, y23 = {yv.z, yv.w};
#pragma unroll
        for (int ii = 0; ii < QPG; ++ii) {
          const float4 xv = *reinterpret_cast<const float4 *>(&Xs[(igA * QPG + ii) * DP + d]);  // wave-uniform: LDS broadcast
          const v2f z01 = v2f{xv.x, xv.y} - y01, z23 = v2f{xv.z, xv.w} - y23;
          d2[ii] = __builtin_elementwise_fma(z01, z01, d2[ii]);
          d2[ii] = __builtin_elementwise_fma(z23, z23, d2[ii]);
        }
      }
#pragma unroll
      for (int ii = 0; ii < QPG; ++ii) {
        const float dd = d2[ii].x + d2[ii].y;
        float v;
        if (PRI) v = (jA < jc) ? lm - 0.5f * dd : -INFINITY;
        else if (MODE == PAIR_K1) v = (jA < jc) ? __builtin_amdgcn_exp2f(-0.72134752044448170f * dd) : 0.f;  // exp(-dd/2) = 2^(-dd/(2 ln 2)); bare v_exp_f32, rel. error ~|x| 2^-24
        else v = (jA < jc) ? __builtin_amdgcn_rsqf(1.0f + dd) : 0.f;  // IMQ: k = (1 + d^2/l^2)^(-1/2) once per pair (bare v_rsq_f32, 1 ulp); k' = -k^3 in pass B
        kv[(igA * QPG + ii) * (JC + 1) + jA] = v;
      }
    }
    wg_sync();
    DUST_TLP(tlp, 4);
    if (PRI) {
      // online softmax over key chunks: row max (8 lanes per query), rescale, exponentiate in place
      float m = -INFINITY;
#pragma unroll
      for (int q = 0; q < JC / 8; ++q) m = fmaxf(m, kv[iB * (JC + 1) + (tid & 7) + 8 * q]);
      m = oct_max(m);  // the 8 lanes of a query are 8 consecutive lanes of one wave: DPP, no LDS crossbar
      const float mo = mrow[iB];
      const float mn = fmaxf(mo, m);
      // bare v_exp_f32 (relative error ~|x| 2^-24, as for the K1 Gram value): weights near 1 are exact to 1 ulp and the
      // ones it perturbs are negligible in the sums
      const float sc = (mo == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((mo - mn) * 1.44269504088896340736f);
#pragma unroll
      for (int c = 0; c < CPT / 2; ++c) accA[c] *= sc;
      accL *= sc;
      // no barrier here: mrow[iB] and the kv entries below are touched only by the 8 lanes of query iB, which run in
      // lockstep (every lane has read mrow[iB] before lane 0 of the group overwrites it)
      if ((tid & 7) == 0) mrow[iB] = mn;
      float rs = 0.f;
#pragma unroll
      for (int q = 0; q < JC / 8; ++q) {
        const int jj = (tid & 7) + 8 * q;
        const float l = kv[iB * (JC + 1) + jj];
        const float e = (mn == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((l - mn) * 1.44269504088896340736f);
        kv[iB * (JC + 1) + jj] = e;
        if (LOGP) rs += e;
      }
      if (LOGP) accL += oct_sum(rs);  // mass of this chunk (pass B, which would add it up, is skipped)
      wg_sync();
    }
    DUST_STAMP(a.stamps, 3);
    DUST_TLP(tlp, 5);
    // ---- pass B: lane = (query, 8 column groups); packed math ----
    if (MFMA_A) {
      // A[i = col][k = key] = S[key][col] (lane: i = l % 16, k = l / 16), B[k = key][j = query] = K[query][key] (lane:
      // j = l % 16, k = l / 16), D'[col][query]: lane holds cols 4 (l / 16) + r of query l % 16
      const float *kb = kv + (mqh * 16 + (ml & 15)) * (JC + 1) + (ml >> 4);
      const float *sb = Vs + (ml >> 4) * YS + mct0 * 16 + (ml & 15);
#pragma unroll
      for (int k4 = 0; k4 < JC / 4; ++k4) {
        const float bq = kb[4 * k4];
#pragma unroll
        for (int t = 0; t < TPW; ++t) accM[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(sb[4 * k4 * YS + 16 * t], bq, accM[t], 0, 0, 0);
      }
    }
#pragma unroll 4
    for (int jj = 0; jj < (LOGP ? 0 : JC); ++jj) {
      const float kq = kv[iB * (JC + 1) + jj];
      float k = kq, kp = 0.f;
      if (MODE == PAIR_K1) kp = -kq;  // d k / d x_i = -k (x_i - x_j) / ell^2
      if (MODE == PAIR_IMQ) kp = -(kq * kq) * kq;  // d k / d x_i = -(1 + d^2/l^2)^(-3/2) (x_i - x_j) / l^2
      const v2f kk = {k, k}, kpp = {kp, kp};
#pragma unroll
      for (int c = 0; c < CPT; c += 4) {
        const float4 yv = *reinterpret_cast<const float4 *>(&Ys[jj * YS + cB + c]);
        const v2f y01 = {yv.x, yv.y}, y23 = {yv.z, yv.w};
        if (PRI) {
          accA[c / 2] = __builtin_elementwise_fma(kk, y01 - xB[c / 2], accA[c / 2]);
          accA[c / 2 + 1] = __builtin_elementwise_fma(kk, y23 - xB[c / 2 + 1], accA[c / 2 + 1]);
        } else {
          if (!MFMA_A) {
            const float4 sv = *reinterpret_cast<const float4 *>(&Vs[jj * YS + cB + c]);
            accA[c / 2] = __builtin_elementwise_fma(kk, v2f{sv.x, sv.y}, accA[c / 2]);
            accA[c / 2 + 1] = __builtin_elementwise_fma(kk, v2f{sv.z, sv.w}, accA[c / 2 + 1]);
          }
          accB[c / 2] = __builtin_elementwise_fma(kpp, xB[c / 2] - y01, accB[c / 2]);
          accB[c / 2 + 1] = __builtin_elementwise_fma(kpp, xB[c / 2 + 1] - y23, accB[c / 2 + 1]);
        }
      }
      if (PRI) accL += k;
    }
    wg_sync();
  }

  DUST_STAMP(a.stamps, 4);
  DUST_TLP(tlp, 6);
  // ---- partial outputs (differences were accumulated in scaled coordinates: undo the 1/s_d) ----
  const int il = tile_x * TI + iB;  // local row
  if (il < a.n_local) {
    // partial rows are padded to DP floats: every lane stores whole 16-byte groups, the 8 lanes of a row one contiguous
    // DP*4-byte run (pad columns hold zeros).  Full-line stores matter for the write-through form: dword-granular sc1
    // stores cost a memory transaction each.
    const size_t row = ((size_t)js * a.n_local + il) * DP;
#pragma unroll
    for (int c = 0; c < CPT; c += 4) {
      v4f oa, ob;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int col = cB + c + k;
        const float un = 1.0f / a.inv_s[da == 1 ? 0 : (da == 2 ? (col & 1) : col % da)];
        const float va = ((c + k) & 1) ? accA[(c + k) / 2].y : accA[(c + k) / 2].x;
        const float vb = ((c + k) & 1) ? accB[(c + k) / 2].y : accB[(c + k) / 2].x;
        oa[k] = (PRI) ? va * un : va;
        ob[k] = vb * un;
      }
      // write_through: sc1 stores so an in-launch consumer on another CU can read them with sc1 loads after the arrival
      // counter, with no release / acquire fence (Guideline 16, R1 form)
      if (!MFMA_A && !LOGP) store16(a.pA + row + cB + c, oa, write_through);
      if (!PRI) store16(a.pB + row + cB + c, ob, write_through);
    }
    if (PRI && (tid & 7) == 0) {
      if (write_through) {
        __hip_atomic_store(a.pM + (size_t)js * a.n_local + il, mrow[iB], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.pL + (size_t)js * a.n_local + il, accL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        a.pM[(size_t)js * a.n_local + il] = mrow[iB];
        a.pL[(size_t)js * a.n_local + il] = accL;
      }
    }
  }
  if (MFMA_A) {  // Gram x score partial rows from the MFMA accumulators: query = l % 16 of the wave's half, 4 consecutive columns
    const int ilm = tile_x * TI + mqh * 16 + (ml & 15);
    if (ilm < a.n_local) {
      const size_t rowm = ((size_t)js * a.n_local + ilm) * DP;
#pragma unroll
      for (int t = 0; t < TPW; ++t) store16(a.pA + rowm + (mct0 + t) * 16 + 4 * (ml >> 4), accM[t], write_through);
    }
  }
  DUST_STAMP(a.stamps, 5);
  DUST_TLP(tlp, 7);
}

// Stein tile of the one-launch SVGD iteration (fused.hpp svgd_iter_kernel): the same arithmetic as pairwise_body<K1|IMQ> on a
// single key chunk (slice <= PAIR_JC), split at the point where the score is first needed.  Everything that reads only
// theta - the Gram values (pass A) and the repulsive term sum_j k'_ij (x_i - x_j) - runs BEFORE the wait on the score rows,
// i.e. underneath the rollouts of the same launch; after the wait only the score tile load, the Gram x score MFMAs and the
// 16-byte partial stores remain on the critical path.  Accumulation orders are those of pairwise_body (bitwise equal).
// arrival lines [line0, line0 + nlines) of `cnt` (one counter per 128-byte line), each with its own target
struct LineGate {
  const unsigned int *cnt;
  int line0, nlines;
  unsigned int target0, target1;
};

template <int MODE, int CPT, bool SC1 = false>
__device__ __forceinline__ void stein_split_body(const PairArgs &a, float *lds, const int tile_x, const int js, const unsigned int *score_cnt,
                                                 const float *score_pub, unsigned int *timeout_flag, unsigned long long *tl,
                                                 const LineGate *gate = nullptr, const float *XYo = nullptr /* override a.X = a.Y */,
                                                 const int tx = (int)threadIdx.x, unsigned long long *tlp = nullptr) {
  static_assert(MODE == PAIR_K1 || MODE == PAIR_IMQ, "Stein modes only");
  static_assert(CPT <= 8, "the Gram x score product runs on the matrix cores (D <= 64)");
  constexpr int TI = PAIR_TI, JC = PAIR_JC, NT = PAIR_NT;
  constexpr int DP = 8 * CPT, YS = DP + 4, QG = NT / JC, QPG = TI / QG, TPW = DP / 32;
  float *Xs = lds;                   // [TI][DP]  queries / ell
  float *Ys = Xs + TI * DP;          // [JC][YS]  keys / ell
  float *Vs = Ys + JC * YS;          // [JC][YS]  score rows
  float *kv = Vs + JC * YS;          // [TI][JC + 1] Gram values
  const int tid = tx;
  const int D = a.D, da = a.da, N = a.N;
  const int ib = a.i0 + tile_x * TI;
  const int jbeg = js * a.slice, jend = min(N, jbeg + a.slice);
  const int nq = min(TI, a.i0 + a.n_local - ib), jc = min(JC, jend - jbeg);
  {
    float vx[RowLane<TI, DP, NT>::NB], vy[RowLane<JC, DP, NT>::NB];
    rowlane_issue<TI, DP, NT, SC1>(XYo ? XYo : a.X, ib, nq, D, vx, tid);
    rowlane_issue<JC, DP, NT, SC1>(XYo ? XYo : a.Y, jbeg, jc, D, vy, tid);
    rowlane_commit<TI, DP, DP, NT, true>(vx, nq, D, da, a.inv_s, Xs, tid);
    rowlane_commit<JC, DP, YS, NT, true>(vy, jc, D, da, a.inv_s, Ys, tid);
  }
  const int iB = tid >> 3, cB = (tid & 7) * CPT;
  v2f accB[CPT / 2];
#pragma unroll
  for (int c = 0; c < CPT / 2; ++c) accB[c] = v2f{0.f, 0.f};
  const int mw = tid >> 6, ml = tid & 63, mqh = mw >> 1, mct0 = (mw & 1) * TPW;
  wg_sync();
  DUST_TLP(tlp, 8);
  v2f xB[CPT / 2];
#pragma unroll
  for (int c = 0; c < CPT / 2; ++c) xB[c] = *reinterpret_cast<const v2f *>(&Xs[iB * DP + cB + 2 * c]);
  {  // pass A: lane = key, QPG queries per lane (pairwise_body)
    const int jA = tid & (JC - 1), igA = tid / JC;
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
#pragma unroll
    for (int ii = 0; ii < QPG; ++ii) {
      const float dd = d2[ii].x + d2[ii].y;
      float v;
      if (MODE == PAIR_K1) v = (jA < jc) ? __builtin_amdgcn_exp2f(-0.72134752044448170f * dd) : 0.f;
      else v = (jA < jc) ? __builtin_amdgcn_rsqf(1.0f + dd) : 0.f;
      kv[(igA * QPG + ii) * (JC + 1) + jA] = v;
    }
  }
  wg_sync();
  DUST_TLP(tlp, 9);
  // repulsive term: lane = (query, 8 column groups)
#pragma unroll 4
  for (int jj = 0; jj < JC; ++jj) {
    const float kq = kv[iB * (JC + 1) + jj];
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
  const int il = tile_x * TI + iB;
  if (il < a.n_local) {
    const size_t row = ((size_t)js * a.n_local + il) * DP;
#pragma unroll
    for (int c = 0; c < CPT; c += 4) {
      v4f ob;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int col = cB + c + k;
        const float un = 1.0f / a.inv_s[da == 1 ? 0 : (da == 2 ? (col & 1) : col % da)];
        ob[k] = (((c + k) & 1) ? accB[(c + k) / 2].y : accB[(c + k) / 2].x) * un;
      }
      store16(a.pB + row + cB + c, ob, true);
    }
  }
  // ---- the score rows of this key slice, published by the rollout role of this launch ----
  DUST_TL(tl, 1);
  DUST_TLP(tlp, 10);
  using RLV = RowLane<JC, DP, NT>;
  float vv[RLV::NB];
  const int vlr = tid / RLV::W, vlc = min(tid % RLV::W, D - 1);
  if (score_pub) {
    // handed over as data: every word of the buffer holds SCORE_SENTINEL until its row is written (through) by the rollout
    // role.  A cheap look at the last word of each row until all 64 have landed, then the tile is loaded and EVERY word is
    // checked (a row is several memory transactions; no assumption on their order or granularity); bounded.
    const float *base = score_pub + (size_t)jbeg * D + vlc;
    unsigned int spins = 0;
    for (;;) {
      {
        int any = 0;
#pragma unroll
        for (int u = 0; u < RLV::NB; ++u) {
          vv[u] = __hip_atomic_load(base + (size_t)min(u * RLV::RB + vlr, jc - 1) * D, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          any |= __float_as_uint(vv[u]) == SCORE_SENTINEL;
        }
        if (!__syncthreads_or(any)) break;
      }
      __builtin_amdgcn_s_sleep(4);
      if (++spins > (1u << 22)) {
        if (tid == 0) *timeout_flag = 1u;
        break;
      }
    }
    DUST_TL(tl, 2);
  } else {
    if (gate) {  // persistent tick (persist.hpp): the keys' score rows are counted per 32-particle group
      if (tid < gate->nlines) spin_until(gate->cnt + (size_t)(gate->line0 + tid) * 32, tid == 0 ? gate->target0 : gate->target1, timeout_flag);
    } else if (tid == 0) {
      const unsigned int target = (unsigned int)(jend - jbeg);
      unsigned int spins = 0;
      while (__hip_atomic_load(score_cnt + js * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {  // 32 = CNT_STRIDE (rollout.hpp)
        __builtin_amdgcn_s_sleep(4);
        if (++spins > (1u << 24)) {
          *timeout_flag = 1u;
          break;
        }
      }
    }
    DUST_TL(tl, 2);
    wg_sync();
    // row-lane staging with sc1 loads (the rows were written through by other CUs in this launch)
    const float *base = a.V + (size_t)jbeg * D + vlc;
#pragma unroll
    for (int u = 0; u < RLV::NB; ++u) vv[u] = __hip_atomic_load(base + (size_t)min(u * RLV::RB + vlr, jc - 1) * D, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  DUST_TLP(tlp, 11);
  if (gate) __builtin_amdgcn_s_setprio(3);  // persistent tick: from here on the tile is on the critical path
  rowlane_commit<JC, DP, YS, NT, false>(vv, jc, D, da, a.inv_s, Vs, tid);
  wg_sync();
  v4f accM[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) accM[t] = v4f{0.f, 0.f, 0.f, 0.f};
  {
    const float *kb = kv + (mqh * 16 + (ml & 15)) * (JC + 1) + (ml >> 4);
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
    for (int t = 0; t < TPW; ++t) store16(a.pA + rowm + (mct0 + t) * 16 + 4 * (ml >> 4), accM[t], true);
  }
}

template <int MODE, int CPT>
__global__ __launch_bounds__(PAIR_NT, (CPT <= 4 ? 4 : 2)) void pairwise_kernel(const PairArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  pairwise_body<MODE, CPT>(a, lds, blockIdx.x, blockIdx.y);
}

static inline size_t pairwise_lds_bytes(int mode, int CPT) {
  const int DP = 8 * CPT;
  return sizeof(float) * ((size_t)PAIR_TI * DP + (size_t)((mode == PAIR_PRIOR || mode == PAIR_LOGP) ? 1 : 2) * PAIR_JC * (DP + 4) + (size_t)PAIR_TI * (PAIR_JC + 1) + PAIR_TI);
}

// Combine the JS slice partials of the prior pass for one row (fixed order -> reproducible): returns grad_pri[d] for the
// columns the caller owns and log p(x_i).  Used by rollout_kernel (score) and logp_merge_kernel (forward).
struct PriorMerge {
  int JS, n_local;
  int ldp;  // row stride of pA (D padded to 8*CPT)
  const float *pA, *pM, *pL;
  float inv_s2[4];
  float log_norm;  // -H sum(log sigma_p) - D/2 log(2 pi)
  // full 2 x 2 prior covariance Sigma_p = L L^T (svgd.py:84-89): the pass ran on WHITENED rows z = L^-1 x (per time step; inv_s2 = 1,
  // log_norm carries -H log det L), so its weighted sum g_z = sum_k r_k (z_k - z_i) is the gradient in whitened coordinates and
  // grad_pri = L^-T g_z - what autograd returns through MultivariateNormal.log_prob's triangular solve
  int full;
  float Lp[3];  // l00, l10, l11
};
// loads are issued in batches of 8 with clamped (never predicated) indices so they overlap instead of serialising
template <bool SC1 = false /* the partials were written (write-through) inside this launch */>
__device__ __forceinline__ void prior_merge_row(const PriorMerge &pm, int il, float *m_out, float *l_out) {
  float m = -INFINITY, l = 0.f;
  for (int q0 = 0; q0 < pm.JS; q0 += 16) {  // 32 independent loads in flight: one round trip for JS <= 16
    float mq[16], lq[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const size_t r = (size_t)min(q0 + u, pm.JS - 1) * pm.n_local + il;
      mq[u] = SC1 ? __hip_atomic_load(pm.pM + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : pm.pM[r];
      lq[u] = SC1 ? __hip_atomic_load(pm.pL + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : pm.pL[r];
    }
    float mc = -INFINITY;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      mq[u] = (q0 + u < pm.JS) ? mq[u] : -INFINITY;  // clamped duplicates carry no mass
      mc = fmaxf(mc, mq[u]);
    }
    const float mn = fmaxf(m, mc);
    float lc = 0.f;
    if (mn != -INFINITY) {  // branch-free inside: exp2(-inf) = 0 drops empty slices (bare v_exp_f32, as in the passes themselves)
#pragma unroll
      for (int u = 0; u < 16; ++u) lc = fmaf(lq[u], __builtin_amdgcn_exp2f((mq[u] - mn) * 1.44269504088896340736f), lc);
      l = l * __builtin_amdgcn_exp2f((m - mn) * 1.44269504088896340736f) + lc;
    }
    m = mn;
  }
  *m_out = m;
  *l_out = l;
}
__device__ __forceinline__ float prior_merge_col(const PriorMerge &pm, int il, int D, int d, int da, float m, float l) {
  float acc = 0.f;
  for (int q0 = 0; q0 < pm.JS; q0 += 8) {
    float mq[8], aq[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const size_t r = (size_t)min(q0 + u, pm.JS - 1) * pm.n_local + il;
      mq[u] = pm.pM[r];
      aq[u] = pm.pA[r * pm.ldp + d];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (q0 + u < pm.JS && mq[u] != -INFINITY) acc = fmaf(aq[u], expf(mq[u] - m), acc);
  }
  return (acc / l) * pm.inv_s2[d % da];
}

// score = grad_lik + grad_pri and/or log p, when no rollout kernel follows the prior pass (stage-wise API, forward)
struct PriorFinishArgs {
  PriorMerge pm;
  int D, da, i0, n_local;
  const float *grad_lik;  // [N][D] or nullptr
  float *grad_pri;        // [N][D] or nullptr
  float *score;           // [N][D] or nullptr
  float *logp;            // [N] or nullptr
  const float *logl;      // [N] with lw: the log-likelihoods
  float *lw;              // [N] or nullptr: log-weights logl + logp out (SVMPC.forward, svmpc.py:190)
};
__global__ void prior_finish_kernel(const PriorFinishArgs a) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= a.n_local * a.D) return;
  const int il = idx / a.D, d = idx - il * a.D;
  float m, l;
  prior_merge_row(a.pm, il, &m, &l);
  const size_t o = (size_t)(a.i0 + il) * a.D + d;
  if (a.grad_pri || a.score) {
    float gp = prior_merge_col(a.pm, il, a.D, d, a.da, m, l);
    if (a.pm.full) {  // back-substitution with L^T over the (even, odd) column pair of this time step
      const float gq = prior_merge_col(a.pm, il, a.D, d ^ 1, a.da, m, l);
      const float g0 = (d & 1) ? gq : gp, g1 = (d & 1) ? gp : gq;
      const float w1 = g1 / a.pm.Lp[2];
      const float w0 = (g0 - a.pm.Lp[1] * w1) / a.pm.Lp[0];
      gp = (d & 1) ? w1 : w0;
    }
    if (a.grad_pri) a.grad_pri[o] = gp;
    if (a.score) a.score[o] = a.grad_lik[o] + gp;
  }
  if (a.logp && d == 0) {
    const float lp = (m + logf(l)) + a.pm.log_norm;
    a.logp[a.i0 + il] = lp;
    if (a.lw) a.lw[a.i0 + il] = a.logl[a.i0 + il] + lp;
  }
}

// z = L^-1 x per time step (forward substitution; MultivariateNormal.log_prob's _batch_mahalanobis): rows [n][H][2] -> [n][H][2]
__global__ void whiten_rows_kernel(const float *x, float *z, const int n_pairs, const float l00, const float l10, const float l11) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pairs) return;
  const float2 v = reinterpret_cast<const float2 *>(x)[i];
  const float z0 = v.x / l00;
  const float z1 = (v.y - l10 * z0) / l11;
  reinterpret_cast<float2 *>(z)[i] = make_float2(z0, z1);
}

// ---------------------------------------------------------------------------------------------------------------
// Stein partial combine + optimiser update (svmpc.py:83 phi = grad_k + K score / N ; svmpc.py:87-95 theta.grad = -phi;
// optimizer.step()).  Writes phi, theta in place (row-major) - each element is touched by exactly one lane.
struct UpdateArgs {
  int N, D, i0, n_local, JS;
  int JSA;               // slices of pA when they differ from pB's (pass 2 of pairwise_fused.hpp has its own grid); 0 = JS
  int ldp;               // row stride of pA / pB (D padded to 8*CPT)
  int optimizer, apply;  // apply = 0: only materialise phi (stage-wise SVMPC.phi)
  float lr, beta1, beta2, eps;
  float inv_l2, inv_n;
  uint32_t *ctr;  // device counters {tick, iter, adam_step}; this kernel advances iter after use
  unsigned int *fused_cnt;  // [fused_tiles] hand-off counters of the fused prior+rollout launch: re-armed (zeroed) here
  int fused_tiles;
  const float *pA, *pB;  // [JS][n_local][ldp]
  float *phi;     // [N][D]
  float *theta;   // [N][D] current particles
  float *theta_out;  // [N][D] where the updated particles go (== theta, or the other buffer of the ping-pong)
  float *adam_m, *adam_v;
};

// `sc1`: the partials were published inside the SAME launch (fused.hpp stein_update_kernel) with write-through stores and
// must be read with sc1 loads; across a kernel boundary plain loads do.
template <bool SC1>
__device__ __forceinline__ void update_body(const UpdateArgs &a, const int idx) {
  if (idx < a.fused_tiles) a.fused_cnt[idx * 32] = 0u;  // CNT_STRIDE (rollout.hpp): one counter per 128-byte line
  if (idx >= a.n_local * a.D) return;
  // counters: this kernel READS adam_step (bumped by the rollout kernel of the same iteration) and ADVANCES iter (read
  // only by rollout kernels) - no launch both reads and writes the same counter
  const float adam_t = SC1 ? (float)__hip_atomic_load(a.ctr + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (float)a.ctr[2];
  const int il = idx / a.D, d = idx - il * a.D;
  const size_t o = (size_t)(a.i0 + il) * a.D + d;
  float th = a.apply ? a.theta[o] : 0.f;  // independent of the partials: in flight together with them
  float sa = 0.f, sb = 0.f;
  const int JSA = a.JSA > 0 ? a.JSA : a.JS;
  for (int q0 = 0; q0 < max(a.JS, JSA); q0 += 16) {  // 32 independent loads in flight (one round trip for JS <= 16), fixed summation order
    float va[16], vb[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const size_t pa = ((size_t)min(q0 + u, JSA - 1) * a.n_local + il) * a.ldp + d;
      const size_t pb = ((size_t)min(q0 + u, a.JS - 1) * a.n_local + il) * a.ldp + d;
      if (SC1) {
        va[u] = __hip_atomic_load(a.pA + pa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        vb[u] = __hip_atomic_load(a.pB + pb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        va[u] = a.pA[pa];
        vb[u] = a.pB[pb];
      }
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (q0 + u < JSA) sa += va[u];
      if (q0 + u < a.JS) sb += vb[u];
    }
  }
  // (after the partials have been read: whatever admitted this lane to them also means every rollout has read the counter)
  if (a.apply && idx == a.n_local * a.D - 1) a.ctr[1] += 1u;
  const float phi = sb * a.inv_l2 + sa * a.inv_n;
  a.phi[o] = phi;
  if (!a.apply) return;
  const float g = -phi;
  if (a.optimizer == DUST_OPT_SGD) {
    th = fmaf(-a.lr, g, th);  // torch SGD: p.add_(grad, alpha=-lr), a vectorised fmadd
  } else {  // torch.optim.Adam (no weight decay, no amsgrad)
    float m = a.adam_m[o], v = a.adam_v[o];
    th = adam_step(th, g, m, v, a.lr, a.beta1, a.beta2, a.eps, adam_t);
    a.adam_m[o] = m;
    a.adam_v[o] = v;
  }
  a.theta_out[o] = th;
}

__global__ void update_kernel(const UpdateArgs a) { update_body<false>(a, blockIdx.x * blockDim.x + threadIdx.x); }

// optimiser update from an already materialised phi (K2 branch, which writes phi directly)
__global__ void update_from_phi_kernel(const UpdateArgs a) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < a.fused_tiles) a.fused_cnt[idx * 32] = 0u;
  if (idx >= a.n_local * a.D) return;
  const float adam_t = (float)a.ctr[2];
  if (idx == a.n_local * a.D - 1) a.ctr[1] += 1u;
  const int il = idx / a.D, d = idx - il * a.D;
  const size_t o = (size_t)(a.i0 + il) * a.D + d;
  float th = a.theta[o];
  const float g = -a.phi[o];
  if (a.optimizer == DUST_OPT_SGD) {
    th = fmaf(-a.lr, g, th);
  } else {
    float m = a.adam_m[o], v = a.adam_v[o];
    th = adam_step(th, g, m, v, a.lr, a.beta1, a.beta2, a.eps, adam_t);
    a.adam_m[o] = m;
    a.adam_v[o] = v;
  }
  a.theta[o] = th;
}

// row-major [N][D] -> transposed [D][N] (K2's per-dimension kernels read the transposed copy)
__global__ void transpose_kernel(const float *src, float *dst, int N, int D) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * D) return;
  const int i = idx / D, d = idx - i * D;
  dst[(size_t)d * N + i] = src[idx];
}

// [R][C] -> [C][R] for the host-facing cost / weight layouts
__global__ void transpose2_kernel(const float *src, float *dst, int R, int C) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= R * C) return;
  const int r = idx / C, c = idx - r * C;
  dst[(size_t)c * R + r] = src[idx];
}

}  // namespace dust
