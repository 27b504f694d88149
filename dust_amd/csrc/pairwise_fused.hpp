// pairwise_fused.hpp - geometry shared by the large-set pairwise passes (N >= 2048 with the prior means aliasing theta: every tick after
// the first, SVMPC.forward -> _update_prior, svmpc.py), and the exact-difference log-p pass.
//
// The two passes of one SVGD iteration themselves - ONE exact-difference distance per (i, j) serving the prior logit and the Stein
// kernel value, then kernel x score on the matrix cores - live in pairwise_packed.hpp (round 6: they walk run lists of near keys; the
// chunk-by-chunk kernels that used to be here visited every (query tile, key chunk) unit).
//
// Register blocking of pass 1: pass A lane = key (row in registers), query rows wave-uniform through scalar loads (as
// pairwise_big.hpp); pass B lane = 4 queries x CB columns.  CB = 8 at D > 32: per key 2 b128 reads of the key row + 4 b64 reads of the
// (softmax term, kernel value) pairs feed 48 packed lane-ops - the LDS pipe stays well below the four SIMDs' demand (at CB = 4 the two
// balance: the 64-wide tile ran 1.75x slower per flop than the 80-wide one until it took CB = 8 with 28 of its 32 query groups).
#pragma once
#include "stein.hpp"

namespace dust {

struct PairFusedArgs {  // the log-p pass below
  PairArgs p;       // the PRIOR's arguments (X = Y = theta, logmix, pM / pL, geometry); inv_s unused
  const float *Xp;  // [N][DPB] zero-padded copy of the particles (pad_rows_kernel)
  int ldp;          // row stride of the partial outputs
  float wP[2];      // 1 / sigma_p^2 for even / odd dimensions
};

template <int DPB>
struct FusedGeom {
  static constexpr int CB = DPB <= 32 ? 4 : 8;           // columns per lane in pass B
  static constexpr int LCG = DPB / CB;                   // column groups
  static constexpr int QG = (PAIR_NT / LCG) & ~1;        // query groups of 4 (even: pass A walks query pairs)
  static constexpr int QGU = DPB == 64 ? 28 : (QG > 32 ? 32 : QG);  // groups in use (DPB = 64: 32 would need 84 KB of LDS - one workgroup per CU)
  static constexpr int TQ = 4 * QGU;                     // queries per tile: 128 / 112 / 96 at DPB = 32 / 64 / 80
  static constexpr int KS = PAIR_JC + 1;
  static constexpr int YS = DPB + 4;
};

template <int DPB>
static inline size_t pairwise_fused_lds_bytes() {
  using G = FusedGeom<DPB>;
  return sizeof(float) * ((size_t)PAIR_JC * G::YS + 2 * (size_t)G::TQ * G::KS + 3 * (size_t)G::TQ + 8);
}

#ifndef DUST_FUSED_WGS
#define DUST_FUSED_WGS 2  // resident workgroups per CU the register budget of pass 1 is set for (pairwise_packed.hpp)
#endif

// ---- log p(theta) only (SVMPC.forward, svmpc.py:128-140): pass A + the chunk-wise log-sum-exp, nothing else -----------------------
// The same tile geometry and key staging as pairwise_packed_kernel without its pass-B state: ~120 VGPRs, three workgroups per CU.
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
  if (jbeg >= jend) {  // (a slice behind the last key: neutral partials)
    wg_sync();
    for (int i = tid; i < TQ; i += NT) {
      const int il = tile * TQ + i;
      if (il < a.n_local) {
        a.pM[(size_t)js * a.n_local + il] = -INFINITY;
        a.pL[(size_t)js * a.n_local + il] = 0.f;
      }
    }
    return;
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
      typedef const v2f __attribute__((address_space(4))) * cv2;  // (scalar loads: see pairwise_packed_kernel)
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

}  // namespace dust
