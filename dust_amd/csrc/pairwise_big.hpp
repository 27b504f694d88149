// pairwise_big.hpp - the pairwise (N x N x D) passes for LARGE key sets (N >= 2048: cfg3 / cfg4, and every rank of the
// weak-scaled multi-GPU runs).  Same mathematics, partial-output format and combine kernels as stein.hpp's 32 x 64 tile
// kernel; different register blocking.
//
// Why a second kernel: the 32 x 64 kernel reads every operand of every FMA from LDS (pass A: 9 b128 reads per 64 lane-ops,
// pass B: 3 reads per 12) and is LDS-bandwidth bound at ~13 % of the fp32 peak once the key loop is long (this kernel:
// 28 % at N = 16384, D = 30).  Here
//   pass A  lane = key j with its row y_j in REGISTERS; the query row x_i is WAVE-UNIFORM, so it is fetched with scalar
//           loads (s_load_dwordx16 from a zero-padded copy of the queries) and used as SGPR operands of v_pk_add_f32:
//           no LDS read in the inner loop.  Squared differences are accumulated unscaled, even and odd dimensions in the
//           two halves of a packed register, and scaled once (1/s^2 per control dimension) - still exact differences first;
//   pass B  lane = 4 queries x 4 columns: per key 4 Gram values (b32) + one b128 of the key row (+ one of the score row)
//           feed 48 lane-ops, which balances the LDS pipe against the four SIMDs.
// Tile: TQ = 4096 / DPB queries x 64 keys per chunk (DPB = padded D: 32 / 64 -> TQ = 128 / 64), 256 lanes.  Used for
// unsharded N >= 2048 with D <= 64 (dust_amd.hip pair_is_big: measured 1.3-1.6x the 32 x 64 kernel there; at D = 80 it
// loses on the prior pass, and a rank of a sharded run has too few 128-query tiles to fill the chip).
#pragma once
#include "stein.hpp"

namespace dust {

struct PairBigArgs {
  PairArgs p;       // X is ignored: the queries come from Xp
  const float *Xp;  // [N][DPB] zero-padded copy of the query particles (pad_rows_kernel)
  int ldp;          // row stride of the partial outputs
  float w[2];       // 1/s^2 for even / odd dimensions (prior: per control dimension; Stein: 1/l^2)
};

// [N][D] -> [N][DPB], zero padded
__global__ void pad_rows_kernel(const float *src, float *dst, int N, int D, int DPB) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * DPB) return;
  const int r = idx / DPB, c = idx - r * DPB;
  dst[idx] = c < D ? src[(size_t)r * D + c] : 0.f;
}

static inline size_t pairwise_big_lds_bytes(int mode, int DPB) {
  const int TQ = 4096 / DPB;
  return sizeof(float) * ((size_t)(mode == PAIR_PRIOR ? 1 : 2) * PAIR_JC * (DPB + 4) + (size_t)TQ * (PAIR_JC + 1) + 2 * (size_t)TQ);
}

template <int MODE, int DPB>
__global__ __launch_bounds__(PAIR_NT, (DPB <= 64 ? 2 : 1)) void pairwise_big_kernel(const PairBigArgs b) {
  constexpr int JC = PAIR_JC, NT = PAIR_NT, TQ = 4096 / DPB, YS = DPB + 4, LC = DPB / 4, KS = JC + 1;
  constexpr int QW = TQ / 4;    // queries per wave in pass A
  constexpr int LQ = NT / TQ;   // lanes per query in the softmax step (2 / 4 / 8)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const PairArgs &a = b.p;
  float *Ys = lds;                                        // [JC][YS] keys (raw coordinates, zero padded)
  float *Vs = Ys + JC * YS;                               // [JC][YS] score rows (Stein)
  float *kv = Vs + (MODE == PAIR_PRIOR ? 0 : JC * YS);    // [TQ][KS] Gram values / softmax terms
  float *mrow = kv + TQ * KS;                             // [TQ] running max (prior)
  float *scl = mrow + TQ;                                 // [TQ] rescale factor of this chunk (prior)
  const int tid = threadIdx.x, D = a.D, N = a.N;
  const int tile = blockIdx.x, js = blockIdx.y;
  const int ib = a.i0 + tile * TQ;  // first query (global index)
  const int jbeg = js * a.slice, jend = min(N, jbeg + a.slice);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), jA = tid & 63;
  const int qg = tid / LC, cg = tid - qg * LC, c0 = 4 * cg;  // pass-B ownership: queries 4 qg .. 4 qg + 3, columns c0 .. c0 + 3

  constexpr int NCT = DPB / 16, NQW = (TQ / 16) / 4;  // MFMA tiles: NCT column tiles x NQW query tiles per wave (4 in all)
  v4f xB[4], accA[4], accB[4], accM[NCT * NQW];
#pragma unroll
  for (int t = 0; t < NCT * NQW; ++t) accM[t] = v4f{0.f, 0.f, 0.f, 0.f};
  float accL[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int gi = min(ib + 4 * qg + r, N - 1);
    xB[r] = *reinterpret_cast<const v4f *>(b.Xp + (size_t)gi * DPB + c0);
    accA[r] = accB[r] = v4f{0.f, 0.f, 0.f, 0.f};
    accL[r] = 0.f;
  }
  if (MODE == PAIR_PRIOR)
    for (int i = tid; i < TQ; i += NT) mrow[i] = -INFINITY;

  for (int j0 = jbeg; j0 < jend; j0 += JC) {
    const int jc = min(JC, jend - j0);
    // ---- key chunk -> LDS (rows j0 .. j0 + jc - 1 are contiguous in HBM) ----
    float vy[RowLane<JC, DPB, NT>::NB], vv[RowLane<JC, DPB, NT>::NB];
    rowlane_issue<JC, DPB, NT>(a.Y, j0, jc, D, vy);
    if (MODE != PAIR_PRIOR) rowlane_issue<JC, DPB, NT>(a.V, j0, jc, D, vv);
    const float lm = (MODE == PAIR_PRIOR) ? a.logmix[j0 + min(jA, jc - 1)] : 0.f;
    wg_sync();  // the previous chunk's pass B is done with Ys / Vs / kv
    rowlane_commit<JC, DPB, YS, NT, false>(vy, jc, D, a.da, a.inv_s, Ys);
    if (MODE != PAIR_PRIOR) rowlane_commit<JC, DPB, YS, NT, false>(vv, jc, D, a.da, a.inv_s, Vs);
    wg_sync();
    // ---- pass A: lane = key jA (row in registers), wave = QW queries, query rows through the scalar path ----
    {
      v2f y[DPB / 2];
#pragma unroll
      for (int p = 0; p < DPB / 4; ++p) {
        const v4f t = *reinterpret_cast<const v4f *>(&Ys[jA * YS + 4 * p]);
        y[2 * p] = v2f{t.x, t.y};
        y[2 * p + 1] = v2f{t.z, t.w};
      }
      // two queries per step, 32 dimensions per segment: the four s_load_dwordx16 of a step are issued together (64
      // SGPRs), so the scalar-cache latency is paid once per 32 packed operations instead of once per 8
      auto gram = [&](const v2f d2) {
        const float dd = d2.x * b.w[0] + d2.y * b.w[1];
        if (MODE == PAIR_PRIOR) return (jA < jc) ? lm - 0.5f * dd : -INFINITY;
        if (MODE == PAIR_K1) return (jA < jc) ? __builtin_amdgcn_exp2f(-0.72134752044448170f * dd) : 0.f;
        return (jA < jc) ? __builtin_amdgcn_rsqf(1.0f + dd) : 0.f;
      };
      for (int qi = 0; qi < QW; qi += 2) {
        const int i = wave * QW + qi;  // wave-uniform
        // rows past the shard / the set are computed and dropped
        const v2f *xa = reinterpret_cast<const v2f *>(b.Xp + (size_t)min(ib + i, N - 1) * DPB);      // uniform addresses:
        const v2f *xb = reinterpret_cast<const v2f *>(b.Xp + (size_t)min(ib + i + 1, N - 1) * DPB);  // scalar loads
        v2f da2 = {0.f, 0.f}, db2 = {0.f, 0.f};
#pragma unroll
        for (int seg = 0; seg < DPB / 32; ++seg) {
          v2f ra[16], rb[16];
#pragma unroll
          for (int p = 0; p < 16; ++p) {
            ra[p] = xa[seg * 16 + p];
            rb[p] = xb[seg * 16 + p];
          }
#pragma unroll
          for (int p = 0; p < 16; ++p) {
            const v2f za = ra[p] - y[seg * 16 + p], zb = rb[p] - y[seg * 16 + p];
            da2 = __builtin_elementwise_fma(za, za, da2);
            db2 = __builtin_elementwise_fma(zb, zb, db2);
          }
        }
        kv[i * KS + jA] = gram(da2);
        kv[(i + 1) * KS + jA] = gram(db2);
      }
    }
    wg_sync();
    if (MODE == PAIR_PRIOR) {
      // online softmax over key chunks: LQ consecutive lanes per query (DPP max, bare v_exp_f32 as in the 32 x 64 kernel)
      const int q = tid / LQ, l = tid - q * LQ;
      float m = -INFINITY;
#pragma unroll
      for (int t = 0; t < JC / LQ; ++t) m = fmaxf(m, kv[q * KS + l + LQ * t]);
      m = LQ == 8 ? oct_max(m) : (LQ == 4 ? quad_max(m) : pair_max(m));
      const float mo = mrow[q];
      const float mn = fmaxf(mo, m);
#pragma unroll
      for (int t = 0; t < JC / LQ; ++t) {
        const int jj = l + LQ * t;
        const float lg = kv[q * KS + jj];
        kv[q * KS + jj] = (mn == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((lg - mn) * 1.44269504088896340736f);
      }
      if (l == 0) {  // the LQ lanes of a query run in lockstep: all have read mrow[q] by now
        mrow[q] = mn;
        scl[q] = (mo == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((mo - mn) * 1.44269504088896340736f);
      }
      wg_sync();
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float sc = scl[4 * qg + r];
        accA[r] *= sc;
        accL[r] *= sc;
      }
    }
    // ---- pass B: lane = 4 queries x 4 columns; Stein modes: the Gram x score product on the matrix cores ----
    if (MODE != PAIR_PRIOR) {
      // 16 tiles of 16 x 16 per chunk (D'[col][query], transposed as in stein.hpp): wave w owns NQW query tiles x NCT column
      // tiles; v_mfma_f32_16x16x4_f32 runs beside the VALU repulsion loop below
#pragma unroll
      for (int k4 = 0; k4 < JC / 4; ++k4) {
        float bq[NQW], as[NCT];
#pragma unroll
        for (int u = 0; u < NQW; ++u) bq[u] = kv[((wave * NQW + u) * 16 + (jA & 15)) * KS + 4 * k4 + (jA >> 4)];
#pragma unroll
        for (int t = 0; t < NCT; ++t) as[t] = Vs[(4 * k4 + (jA >> 4)) * YS + 16 * t + (jA & 15)];
#pragma unroll
        for (int u = 0; u < NQW; ++u)
#pragma unroll
          for (int t = 0; t < NCT; ++t) accM[u * NCT + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(as[t], bq[u], accM[u * NCT + t], 0, 0, 0);
      }
    }
#pragma unroll 4
    for (int jj = 0; jj < JC; ++jj) {
      const v4f yv = *reinterpret_cast<const v4f *>(&Ys[jj * YS + c0]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float k = kv[(4 * qg + r) * KS + jj];
        if (MODE == PAIR_PRIOR) {
          accA[r] = __builtin_elementwise_fma(v4f{k, k, k, k}, yv - xB[r], accA[r]);
          accL[r] += k;
        } else {
          const float kp = (MODE == PAIR_K1) ? -k : -(k * k) * k;
          accB[r] = __builtin_elementwise_fma(v4f{kp, kp, kp, kp}, xB[r] - yv, accB[r]);
        }
      }
    }
  }

  // ---- partial outputs (same layout as stein.hpp: [js][n_local][ldp], raw coordinates) ----
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int il = tile * TQ + 4 * qg + r;
    if (il >= a.n_local) continue;
    if (c0 < b.ldp) {
      const size_t row = ((size_t)js * a.n_local + il) * b.ldp;
      if (MODE == PAIR_PRIOR) *reinterpret_cast<v4f *>(a.pA + row + c0) = accA[r];
      else *reinterpret_cast<v4f *>(a.pB + row + c0) = accB[r];
    }
    if (MODE == PAIR_PRIOR && cg == 0) {
      a.pM[(size_t)js * a.n_local + il] = mrow[4 * qg + r];
      a.pL[(size_t)js * a.n_local + il] = accL[r];
    }
  }
  if (MODE != PAIR_PRIOR) {  // Gram x score rows from the MFMA accumulators: query = l % 16 of the tile, columns 16 t + 4 (l / 16) ..
#pragma unroll
    for (int u = 0; u < NQW; ++u) {
      const int il = tile * TQ + (wave * NQW + u) * 16 + (jA & 15);
      if (il >= a.n_local) continue;
      const size_t row = ((size_t)js * a.n_local + il) * b.ldp;
#pragma unroll
      for (int t = 0; t < NCT; ++t)
        if (16 * t + 4 * (jA >> 4) < b.ldp) *reinterpret_cast<v4f *>(a.pA + row + 16 * t + 4 * (jA >> 4)) = accM[u * NCT + t];
    }
  }
}

}  // namespace dust
