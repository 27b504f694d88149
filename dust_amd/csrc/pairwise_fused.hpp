// pairwise_fused.hpp - the two pairwise passes of one SVGD iteration for LARGE particle sets (N >= 2048: cfg3 / cfg4) when the
// prior means alias theta (every tick after the first: SVMPC.forward -> _update_prior, svmpc.py), in TWO launches instead of
// 2 x (N x N x D) distance passes:
//
//   pass 1  pairwise_fused_kernel: ONE exact-difference distance per (i, j) serves the prior logit (-d2_p/2 + log w_j) AND the Stein
//           kernel value k_ij; pass B forms the difference y_j - x_i once and feeds it to BOTH weighted sums - the prior's
//           softmax-weighted sum (grad_pri) and the Stein repulsion sum_j k'_ij (x_i - x_j), which needs no score.  k_ij is written
//           to HBM ([n_local][ldK] fp32, 1 GB at N = 16384).  Outputs: the prior partials in the format of stein.hpp (pA, pM, pL:
//           the rollout kernel / prior_finish merge them as before), the repulsion partials pB, the Gram matrix K.
//   pass 2  gram_score_kernel: pA = K x score on the matrix cores (v_mfma_f32_16x16x4_f32, fp32 in / fp32 accumulate - the same
//           product stein.hpp's pass B runs), K streamed from HBM: a plain GEMM, HBM- and MFMA-balanced (1 GB / 43 GFLOP).
//           update_kernel then combines pA and pB over the key slices exactly as after the unfused Stein pass.
//
// Arithmetic is unchanged from pairwise_big.hpp: differences first (x - y in fp32, squared and accumulated unscaled in packed
// halves, scaled once), bare v_exp_f32, the chunk-wise online softmax for the prior.  What changes is the work: per (i, j)
// 1 D (distance) + 2 D (difference, two FMAs) packed lane-ops here + D on the matrix cores, against 2 x (1 D + 1 D) + D before.
//
// Register blocking: pass A lane = key (row in registers), query rows wave-uniform through scalar loads (as pairwise_big.hpp);
// pass B lane = 4 queries x CB columns.  CB = 8 at D > 64: per key 2 b128 reads of the key row + 8 b32 reads of the two weights
// feed 72 packed lane-ops - the LDS pipe stays below half of the four SIMDs' demand (at CB = 4 it would saturate).
#pragma once
#include "stein.hpp"

namespace dust {

struct PairFusedArgs {
  PairArgs p;       // the PRIOR's arguments (X = Y = theta, logmix, pA / pM / pL, geometry); inv_s unused
  const float *Xp;  // [N][DPB] zero-padded copy of the particles (pad_rows_kernel)
  int ldp;          // row stride of the partial outputs
  float wP[2];      // 1 / sigma_p^2 for even / odd dimensions
  float wS[2];      // 1 / ell^2 (both)
  float *pB;        // [JS][n_local][ldp] repulsion partials
  float *K;         // [n_local][ldK] Stein kernel values
  int ldK;
};

template <int DPB>
struct FusedGeom {
  static constexpr int CB = DPB <= 64 ? 4 : 8;           // columns per lane in pass B
  static constexpr int LCG = DPB / CB;                   // column groups
  static constexpr int QG = (PAIR_NT / LCG) & ~1;        // query groups of 4 (even: pass A walks query pairs)
  static constexpr int TQ = 4 * (QG > 32 ? 32 : QG);     // queries per tile: 128 / 64 / 96 at DPB = 32 / 64 / 80
  static constexpr int KS = PAIR_JC + 1;
  static constexpr int YS = DPB + 4;
};

template <int DPB>
static inline size_t pairwise_fused_lds_bytes() {
  using G = FusedGeom<DPB>;
  return sizeof(float) * ((size_t)PAIR_JC * G::YS + 2 * (size_t)G::TQ * G::KS + 2 * (size_t)G::TQ);
}

#ifndef DUST_FUSED_WGS
#define DUST_FUSED_WGS 2  // resident workgroups per CU the register budget is set for (tools/fused_race.hip builds it at 1 too)
#endif
template <int MODE /* PAIR_K1 / PAIR_IMQ: the Stein kernel */, int DPB>
__global__ __launch_bounds__(PAIR_NT, DUST_FUSED_WGS) void pairwise_fused_kernel(const PairFusedArgs b) {
  using G = FusedGeom<DPB>;
  constexpr int JC = PAIR_JC, NT = PAIR_NT, TQ = G::TQ, YS = G::YS, KS = G::KS, CB = G::CB, LCG = G::LCG, NV = CB / 4;
  constexpr int QW = TQ / 4;                                  // queries per wave in pass A
  constexpr int QS = TQ / 4;  // pass-B ownership: lane group qg holds queries qg + QS r (r < 4): with the odd row stride KS the
                              // per-key weight reads of a wave then fall into TQ / 4 different banks (rows 4 qg + r collide 3-way)
  constexpr int LQ = NT / TQ >= 8 ? 8 : (NT / TQ >= 4 ? 4 : 2);  // lanes per query in the softmax step
  static_assert(MODE == PAIR_K1 || MODE == PAIR_IMQ, "Stein kernel family");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const PairArgs &a = b.p;
  float *Ys = lds;                 // [JC][YS] keys (raw coordinates, zero padded)
  float *kvP = Ys + JC * YS;       // [TQ][KS] prior logits -> softmax terms
  float *kvS = kvP + TQ * KS;      // [TQ][KS] Stein kernel values
  float *mrow = kvS + TQ * KS;     // [TQ] running max
  float *scl = mrow + TQ;          // [TQ] rescale factor of this chunk
  const int tid = threadIdx.x, D = a.D, N = a.N;
  const int tile = blockIdx.x, js = blockIdx.y;
  const int ib = a.i0 + tile * TQ;  // first query (global index)
  const int jbeg = js * a.slice, jend = min(N, jbeg + a.slice);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), jA = tid & 63;
  const int qg = tid / LCG, cg = tid - qg * LCG, c0 = CB * cg;  // pass-B ownership: queries qg + QS r, columns c0 .. c0 + CB - 1
  const bool pb = qg < TQ / 4;                                  // (DPB = 80: 240 of the 256 lanes)
  const int qgc = pb ? qg : 0;

  v4f xB[4][NV] /* -x_i */, accA[4][NV], accB[4][NV];
  float accL[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int gi = min(ib + qgc + QS * r, N - 1);
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      xB[r][u] = -*reinterpret_cast<const v4f *>(b.Xp + (size_t)gi * DPB + c0 + 4 * u);  // NEGATED: y + (-x) is one v_pk_add_f32 per pair of columns
      asm volatile("" : "+v"(xB[r][u]));  // (opaque: otherwise the negation folds back into 4 scalar v_sub_f32 per difference)
      accA[r][u] = accB[r][u] = v4f{0.f, 0.f, 0.f, 0.f};
    }
    accL[r] = 0.f;
  }
  for (int i = tid; i < TQ; i += NT) mrow[i] = -INFINITY;

  // Key chunks: the keys ARE the particles (prior means aliased to theta), so a chunk is rows j0 .. j0 + 63 of the padded copy -
  // one contiguous 64 * DPB float run, fetched with 16-byte loads (NLD per lane).  The NEXT chunk's loads are issued before pass B
  // and committed to LDS after it: their HBM / L2 latency hides under pass B instead of opening every chunk (2 workgroups per CU
  // cannot hide it by themselves: measured 0.7 of 3.0 ms at cfg4).
  constexpr int NLD = (JC * DPB / 4 + NT - 1) / NT;
  v4f ky[NLD];
  float lm_next;
  auto keys_issue = [&](const int j0) {
    const int jc = min(JC, jend - j0);
    const v4f *src = reinterpret_cast<const v4f *>(b.Xp + (size_t)j0 * DPB);
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int f = tid + NT * u;                       // 16-byte piece of the chunk
      const int row = (f * 4) / DPB;                    // (DPB % 4 == 0: a piece never straddles two rows)
      ky[u] = src[min(row, jc - 1) * (DPB / 4) + (f - row * (DPB / 4))];  // rows past the slice: clamped, zeroed at the commit
    }
    lm_next = a.logmix[j0 + min(jA, jc - 1)];
  };
  auto keys_commit = [&](const int j0) {
    const int jc = min(JC, jend - j0);
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int f = tid + NT * u;
      const int row = (f * 4) / DPB, col = f * 4 - row * DPB;
      if (row < JC) *reinterpret_cast<v4f *>(&Ys[row * YS + col]) = row < jc ? ky[u] : v4f{0.f, 0.f, 0.f, 0.f};
    }
  };
  keys_issue(jbeg);
  keys_commit(jbeg);
  for (int j0 = jbeg; j0 < jend; j0 += JC) {
    const int jc = min(JC, jend - j0);
    const float lm = lm_next;
    wg_sync();  // Ys holds this chunk
    // ---- pass A: lane = key jA (row in registers), wave = QW queries, query rows through the scalar path ----
    {
      v2f y[DPB / 2];
#pragma unroll
      for (int p = 0; p < DPB / 4; ++p) {
        const v4f t = *reinterpret_cast<const v4f *>(&Ys[jA * YS + 4 * p]);
        y[2 * p] = v2f{t.x, t.y};
        y[2 * p + 1] = v2f{t.z, t.w};
      }
      const bool kval = jA < jc;
      for (int qi = 0; qi < QW; qi += 2) {
        const int i = wave * QW + qi;  // wave-uniform
        // uniform addresses -> scalar loads.  The kernel also STORES to global memory inside this loop (the Gram rows), so a plain
        // load of Xp counts as clobberable and would become a per-lane vector load; the padded query copy is never written by
        // this kernel: address it through the constant address space, whose loads are invariant by definition
        typedef const v2f __attribute__((address_space(4))) * cv2;
        const cv2 xa = (cv2)(uintptr_t)(b.Xp + (size_t)min(ib + i, N - 1) * DPB);
        const cv2 xb = (cv2)(uintptr_t)(b.Xp + (size_t)min(ib + i + 1, N - 1) * DPB);
        v2f da2 = {0.f, 0.f}, db2 = {0.f, 0.f};
#pragma unroll
        for (int s0 = 0; s0 < DPB / 2; s0 += 16) {
          constexpr int dummy = 0;
          (void)dummy;
          v2f ra[16], rb[16];
#pragma unroll
          for (int p = 0; p < 16; ++p)
            if (s0 + p < DPB / 2) {
              ra[p] = xa[s0 + p];
              rb[p] = xb[s0 + p];
            }
#pragma unroll
          for (int p = 0; p < 16; ++p)
            if (s0 + p < DPB / 2) {
              const v2f za = ra[p] - y[s0 + p], zb = rb[p] - y[s0 + p];
              da2 = __builtin_elementwise_fma(za, za, da2);
              db2 = __builtin_elementwise_fma(zb, zb, db2);
            }
        }
        // prior logit and Stein kernel value from the same squared differences
        const float pa = da2.x * b.wP[0] + da2.y * b.wP[1], pbq = db2.x * b.wP[0] + db2.y * b.wP[1];
        const float sa = da2.x * b.wS[0] + da2.y * b.wS[1], sb = db2.x * b.wS[0] + db2.y * b.wS[1];
        kvP[i * KS + jA] = kval ? lm - 0.5f * pa : -INFINITY;
        kvP[(i + 1) * KS + jA] = kval ? lm - 0.5f * pbq : -INFINITY;
        float ka, kb;
        if (MODE == PAIR_K1) {
          ka = __builtin_amdgcn_exp2f(-0.72134752044448170f * sa);
          kb = __builtin_amdgcn_exp2f(-0.72134752044448170f * sb);
        } else {
          ka = __builtin_amdgcn_rsqf(1.0f + sa);
          kb = __builtin_amdgcn_rsqf(1.0f + sb);
        }
        ka = kval ? ka : 0.f;
        kb = kval ? kb : 0.f;
        kvS[i * KS + jA] = ka;
        kvS[(i + 1) * KS + jA] = kb;
        // Gram matrix rows for pass 2: one 256-byte run per query and wave
        const int il = tile * TQ + i;
        if (kval && il < a.n_local) b.K[(size_t)il * b.ldK + j0 + jA] = ka;
        if (kval && il + 1 < a.n_local) b.K[(size_t)(il + 1) * b.ldK + j0 + jA] = kb;
      }
    }
    wg_sync();
    {
      // online softmax over key chunks: LQ consecutive lanes per query (DPP max, bare v_exp_f32 as in the 32 x 64 kernel)
      const int q = tid / LQ, l = tid - q * LQ;
      if (q < TQ) {
        float m = -INFINITY;
#pragma unroll
        for (int t = 0; t < JC / LQ; ++t) m = fmaxf(m, kvP[q * KS + l + LQ * t]);
        m = LQ == 8 ? oct_max(m) : (LQ == 4 ? quad_max(m) : pair_max(m));
        const float mo = mrow[q];
        const float mn = fmaxf(mo, m);
#pragma unroll
        for (int t = 0; t < JC / LQ; ++t) {
          const int jj = l + LQ * t;
          const float lg = kvP[q * KS + jj];
          kvP[q * KS + jj] = (mn == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((lg - mn) * 1.44269504088896340736f);
        }
        if (l == 0) {  // the LQ lanes of a query run in lockstep: all have read mrow[q] by now
          mrow[q] = mn;
          scl[q] = (mo == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((mo - mn) * 1.44269504088896340736f);
        }
      }
      wg_sync();
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float sc = scl[qgc + QS * r];
#pragma unroll
        for (int u = 0; u < NV; ++u) accA[r][u] *= sc;
        accL[r] *= sc;
      }
    }
    const bool more = j0 + JC < jend;
    if (more) keys_issue(j0 + JC);  // in flight during pass B
    // ---- pass B: lane = 4 queries x CB columns; the difference y_j - x_i feeds the prior sum and the repulsion sum ----
    if (pb) {
#pragma unroll 2
      for (int jj = 0; jj < JC; ++jj) {
        v4f yv[NV];
#pragma unroll
        for (int u = 0; u < NV; ++u) yv[u] = *reinterpret_cast<const v4f *>(&Ys[jj * YS + c0 + 4 * u]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float wp = kvP[(qg + QS * r) * KS + jj];
          const float ks = kvS[(qg + QS * r) * KS + jj];
          // -k' of stein.hpp's pass B: accB += k' (x_i - y_j) = (-k') (y_j - x_i), the same product bit for bit
          const float nk = (MODE == PAIR_K1) ? ks : (ks * ks) * ks;
#pragma unroll
          for (int u = 0; u < NV; ++u) {
            const v4f diff = yv[u] + xB[r][u];  // y_j - x_i
            accA[r][u] = __builtin_elementwise_fma(v4f{wp, wp, wp, wp}, diff, accA[r][u]);
            accB[r][u] = __builtin_elementwise_fma(v4f{nk, nk, nk, nk}, diff, accB[r][u]);
          }
          accL[r] += wp;
        }
      }
    }
    wg_sync();  // pass B is done with Ys / kvP / kvS
    if (more) keys_commit(j0 + JC);
  }

  // ---- partial outputs (layout of stein.hpp: [js][n_local][ldp], raw coordinates) ----
  if (pb) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int il = tile * TQ + qg + QS * r;
      if (il >= a.n_local) continue;
      const size_t row = ((size_t)js * a.n_local + il) * b.ldp;
#pragma unroll
      for (int u = 0; u < NV; ++u)
        if (c0 + 4 * u < b.ldp) {
          *reinterpret_cast<v4f *>(a.pA + row + c0 + 4 * u) = accA[r][u];
          *reinterpret_cast<v4f *>(b.pB + row + c0 + 4 * u) = accB[r][u];
        }
      if (cg == 0) {
        a.pM[(size_t)js * a.n_local + il] = mrow[qg + QS * r];
        a.pL[(size_t)js * a.n_local + il] = accL[r];
      }
    }
  }
}

// ---- log p(theta) only (SVMPC.forward, svmpc.py:128-140): pass A + the chunk-wise log-sum-exp, nothing else -----------------------
// The same tile geometry and key staging as pairwise_fused_kernel without its pass-B state: ~120 VGPRs, three workgroups per CU.
// Partials: pM (slice max of the logits), pL (sum of exp(logit - max)); prior_finish_kernel merges the slices.
template <int DPB>
static inline size_t pairwise_logp_big_lds_bytes() {
  using G = FusedGeom<DPB>;
  return sizeof(float) * ((size_t)PAIR_JC * G::YS + (size_t)G::TQ * G::KS + 2 * (size_t)G::TQ);
}

template <int DPB>
__global__ __launch_bounds__(PAIR_NT, 3) void pairwise_logp_big_kernel(const PairFusedArgs b) {
  using G = FusedGeom<DPB>;
  constexpr int JC = PAIR_JC, NT = PAIR_NT, TQ = G::TQ, YS = G::YS, KS = G::KS;
  constexpr int QW = TQ / 4;
  constexpr int LQ = NT / TQ >= 8 ? 8 : (NT / TQ >= 4 ? 4 : 2);
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const PairArgs &a = b.p;
  float *Ys = lds;              // [JC][YS]
  float *kvP = Ys + JC * YS;    // [TQ][KS] logits
  float *mrow = kvP + TQ * KS;  // [TQ] running max
  float *lrow = mrow + TQ;      // [TQ] running sum of exp(logit - max)
  const int tid = threadIdx.x, N = a.N;
  const int tile = blockIdx.x, js = blockIdx.y;
  const int ib = a.i0 + tile * TQ;
  const int jbeg = js * a.slice, jend = min(N, jbeg + a.slice);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), jA = tid & 63;
  for (int i = tid; i < TQ; i += NT) {
    mrow[i] = -INFINITY;
    lrow[i] = 0.f;
  }
  constexpr int NLD = (JC * DPB / 4 + NT - 1) / NT;
  v4f ky[NLD];
  float lm_next;
  auto keys_issue = [&](const int j0) {
    const int jc = min(JC, jend - j0);
    const v4f *src = reinterpret_cast<const v4f *>(b.Xp + (size_t)j0 * DPB);
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int f = tid + NT * u;
      const int row = (f * 4) / DPB;
      ky[u] = src[min(row, jc - 1) * (DPB / 4) + (f - row * (DPB / 4))];
    }
    lm_next = a.logmix[j0 + min(jA, jc - 1)];
  };
  auto keys_commit = [&](const int j0) {
    const int jc = min(JC, jend - j0);
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int f = tid + NT * u;
      const int row = (f * 4) / DPB, col = f * 4 - row * DPB;
      if (row < JC) *reinterpret_cast<v4f *>(&Ys[row * YS + col]) = row < jc ? ky[u] : v4f{0.f, 0.f, 0.f, 0.f};
    }
  };
  keys_issue(jbeg);
  keys_commit(jbeg);
  for (int j0 = jbeg; j0 < jend; j0 += JC) {
    const int jc = min(JC, jend - j0);
    const float lm = lm_next;
    wg_sync();  // Ys holds this chunk; the previous chunk's reduction is done with kvP
    const bool more = j0 + JC < jend;
    v2f y[DPB / 2];
#pragma unroll
    for (int p = 0; p < DPB / 4; ++p) {
      const v4f t = *reinterpret_cast<const v4f *>(&Ys[jA * YS + 4 * p]);
      y[2 * p] = v2f{t.x, t.y};
      y[2 * p + 1] = v2f{t.z, t.w};
    }
    wg_sync();  // every lane holds its key row: Ys may be refilled
    if (more) keys_issue(j0 + JC);  // in flight during the distance pass
    const bool kval = jA < jc;
    for (int qi = 0; qi < QW; qi += 2) {
      const int i = wave * QW + qi;
      typedef const v2f __attribute__((address_space(4))) * cv2;  // (scalar loads: see pairwise_fused_kernel)
      const cv2 xa = (cv2)(uintptr_t)(b.Xp + (size_t)min(ib + i, N - 1) * DPB);
      const cv2 xb = (cv2)(uintptr_t)(b.Xp + (size_t)min(ib + i + 1, N - 1) * DPB);
      v2f da2 = {0.f, 0.f}, db2 = {0.f, 0.f};
#pragma unroll
      for (int s0 = 0; s0 < DPB / 2; s0 += 16) {
        v2f ra[16], rb[16];
#pragma unroll
        for (int p = 0; p < 16; ++p)
          if (s0 + p < DPB / 2) {
            ra[p] = xa[s0 + p];
            rb[p] = xb[s0 + p];
          }
#pragma unroll
        for (int p = 0; p < 16; ++p)
          if (s0 + p < DPB / 2) {
            const v2f za = ra[p] - y[s0 + p], zb = rb[p] - y[s0 + p];
            da2 = __builtin_elementwise_fma(za, za, da2);
            db2 = __builtin_elementwise_fma(zb, zb, db2);
          }
      }
      const float pa = da2.x * b.wP[0] + da2.y * b.wP[1], pbq = db2.x * b.wP[0] + db2.y * b.wP[1];
      kvP[i * KS + jA] = kval ? lm - 0.5f * pa : -INFINITY;
      kvP[(i + 1) * KS + jA] = kval ? lm - 0.5f * pbq : -INFINITY;
    }
    if (more) keys_commit(j0 + JC);
    wg_sync();
    {
      const int q = tid / LQ, l = tid - q * LQ;
      if (q < TQ) {
        float m = -INFINITY;
#pragma unroll
        for (int t = 0; t < JC / LQ; ++t) m = fmaxf(m, kvP[q * KS + l + LQ * t]);
        m = LQ == 8 ? oct_max(m) : (LQ == 4 ? quad_max(m) : pair_max(m));
        const float mo = mrow[q];
        const float mn = fmaxf(mo, m);
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < JC / LQ; ++t) {
          const float lg = kvP[q * KS + l + LQ * t];
          sum += (mn == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((lg - mn) * 1.44269504088896340736f);
        }
        // sum over the LQ lanes of the query (consecutive lanes of one wave)
        for (int o = 1; o < LQ; o <<= 1) sum += __shfl_xor(sum, o);
        if (l == 0) {
          const float sc = (mo == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((mo - mn) * 1.44269504088896340736f);
          mrow[q] = mn;
          lrow[q] = lrow[q] * sc + sum;
        }
      }
    }
  }
  wg_sync();
  for (int i = tid; i < TQ; i += NT) {
    const int il = tile * TQ + i;
    if (il < a.n_local) {
      a.pM[(size_t)js * a.n_local + il] = mrow[i];
      a.pL[(size_t)js * a.n_local + il] = lrow[i];
    }
  }
}

// ---- pass 2: pA[js][i][:] = sum_{j in slice js} K[i][j] score[j][:] -----------------------------------------------------------
struct GramScoreArgs {
  int N, D, i0, n_local, JS, slice, ldp, ldK;
  const float *K;  // [n_local][ldK]
  const float *V;  // [N][D] score
  float *pA;       // [JS][n_local][ldp]
};

template <int DPB>
static inline size_t gram_score_lds_bytes() {
  return sizeof(float) * ((size_t)PAIR_JC * (DPB + 4) + 64 * (size_t)(PAIR_JC + 4));
}

// Tile: 64 queries x DPB columns per workgroup (wave w owns queries 16 w .. 16 w + 15 and all DPB / 16 column tiles), keys in
// chunks of 64.  D'[col][query] += V^T[col][key] K^T[key][query]: the A operand is a score column block, the B operand the Gram
// rows - both read from LDS as stein.hpp's pass B reads them.  The K tile arrives with 16-byte loads along the key index.
template <int DPB>
__global__ __launch_bounds__(PAIR_NT, 4) void gram_score_kernel(const GramScoreArgs a) {
  constexpr int JC = PAIR_JC, NT = PAIR_NT, YS = DPB + 4, KS2 = JC + 4, NCT = DPB / 16;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *Vs = lds;            // [JC][YS] score rows of the chunk
  float *Kt = Vs + JC * YS;   // [64][KS2] Gram rows of the tile (query-major)
  const int tid = threadIdx.x, D = a.D, N = a.N;
  const int tile = blockIdx.x, js = blockIdx.y;
  const int il0 = tile * 64;
  const int jbeg = js * a.slice, jend = min(N, jbeg + a.slice);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), jA = tid & 63;
  v4f acc[NCT];
#pragma unroll
  for (int t = 0; t < NCT; ++t) acc[t] = v4f{0.f, 0.f, 0.f, 0.f};
  // K tile loads: lane = (query row kr + 16 u, 4 keys at kc): 4 b128 loads per lane and chunk
  const int kr = tid >> 4, kc = 4 * (tid & 15);
  for (int j0 = jbeg; j0 < jend; j0 += JC) {
    const int jc = min(JC, jend - j0);
    float vv[RowLane<JC, DPB, NT>::NB];
    rowlane_issue<JC, DPB, NT>(a.V, j0, jc, D, vv);
    v4f kt[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int il = min(il0 + kr + 16 * u, a.n_local - 1);
      kt[u] = *reinterpret_cast<const v4f *>(a.K + (size_t)il * a.ldK + j0 + kc);  // (ldK is a multiple of 64: in bounds; the tail is masked)
    }
    wg_sync();  // the previous chunk's products are done with Vs / Kt
    rowlane_commit<JC, DPB, YS, NT, false>(vv, jc, D, 1, nullptr, Vs);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      v4f t = kt[u];
      t.x = kc + 0 < jc ? t.x : 0.f;
      t.y = kc + 1 < jc ? t.y : 0.f;
      t.z = kc + 2 < jc ? t.z : 0.f;
      t.w = kc + 3 < jc ? t.w : 0.f;
      *reinterpret_cast<v4f *>(&Kt[(kr + 16 * u) * KS2 + kc]) = t;
    }
    wg_sync();
#pragma unroll 4
    for (int k4 = 0; k4 < JC / 4; ++k4) {
      const float bq = Kt[(wave * 16 + (jA & 15)) * KS2 + 4 * k4 + (jA >> 4)];
      float as[NCT];
#pragma unroll
      for (int t = 0; t < NCT; ++t) as[t] = Vs[(4 * k4 + (jA >> 4)) * YS + 16 * t + (jA & 15)];
#pragma unroll
      for (int t = 0; t < NCT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(as[t], bq, acc[t], 0, 0, 0);
    }
  }
  // rows from the accumulators: query = l % 16 of the wave's tile, columns 16 t + 4 (l / 16) ..
  const int il = il0 + wave * 16 + (jA & 15);
  if (il < a.n_local) {
    const size_t row = ((size_t)js * a.n_local + il) * a.ldp;
#pragma unroll
    for (int t = 0; t < NCT; ++t)
      if (16 * t + 4 * (jA >> 4) < a.ldp) *reinterpret_cast<v4f *>(a.pA + row + 16 * t + 4 * (jA >> 4)) = acc[t];
  }
}

}  // namespace dust
