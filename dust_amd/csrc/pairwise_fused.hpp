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
// pass B lane = 4 queries x CB columns.  CB = 8 at D > 32: per key 2 b128 reads of the key row + 4 b64 reads of the (softmax
// term, kernel value) pairs feed 48 packed lane-ops - the LDS pipe stays well below the four SIMDs' demand (at CB = 4 the two
// balance: the 64-wide tile ran 1.75x slower per flop than the 80-wide one until it took CB = 8 with 28 of its 32 query groups).
// Work split: equal contiguous runs of (query tile, key chunk) units per workgroup (fused_balance below), not a (tile, slice) grid.
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
  float *K;         // [tiles * TQ][ldK] Stein kernel values (tiles * TQ >= n_local: whole query tiles)
  int ldK;
  int tiles, chunks;  // query tiles of TQ rows, key chunks of 64: the launch covers tiles * chunks units (see fused_balance)
  unsigned char *nz;  // [chunks][ldnz] 1 = the 64 Stein kernel values of (key chunk, query row) are not all exactly 0 (see below); ldnz >= rows, multiple of 64
  int ldnz;
  const unsigned char *far;  // [tiles][chunks] 1 = the unit contributes exactly nothing and is not visited (pairwise_far.hpp), or nullptr
  const unsigned int *qmask;  // [tiles][chunks][8] words 0-3: bit q = query q of the tile has a near key in the chunk; words 4-5: bit k = key k of
                              // the chunk has a near query in the tile (pairwise_far.hpp), or nullptr: all
  const float *m0;           // [N] where each query's running max starts (pairwise_far.hpp: a lower bound of its final max), or nullptr: -inf
};

// Exact zeros.  K1's lengthscale is fixed at ln 2 (svmpc.py:78), so k_ij = exp(-d2 / 0.96) UNDERFLOWS to exactly 0 in fp32 once
// d2 > ~84 - in H d_a = 80 dimensions at sigma = 5 that is every pair except near-duplicates (tools/kernel_sparsity.py: 8 615 of
// 16.8 M sampled pairs at cfg4 after 100 ticks; a clustered Pendulum set has 33 % non-zeros).  A zero kernel value contributes
// exactly nothing to the repulsion sum and to K x score, so (query row, 64-key chunk) blocks whose kernel values are all zero are
// flagged by pass A (one ballot per row and chunk) and skipped: pass B drops the repulsion FMAs of a chunk that is zero for the
// whole tile, and gram_score_kernel skips the (64 rows x 64 keys) blocks of its tile that are zero (the Gram rows themselves are
// always stored: a block with one non-zero row is read whole).  This pays only for WELL separated sets - a few thousand scattered
// near-duplicates among 16 384 particles already touch most blocks.
// Results are bit-identical to the dense evaluation for finite inputs (0 x finite = 0; DUST_DENSE=1 evaluates everything, and the
// tests compare the two bitwise); the time is data dependent: a dense (clustered) set runs as before.

// Work split of pairwise_fused_kernel.  A unit is (query tile, 64-key chunk), units ordered tile-major; workgroup s of W takes the
// contiguous run [s T / W, (s + 1) T / W) of the T = tiles * chunks units: every workgroup does the same work to within one
// chunk and the launch is ONE resident round (W = 2 per CU).  A (tile, key slice) grid has to round its workgroup count up to
// whole rounds instead - 171 tiles x 20 slices = 6.7 rounds of 512 at N = 16384: 4 % idle, and 20 partial rows per particle
// for the merge kernels against 4 here.  A run that crosses a tile boundary flushes its partial sums and starts the next tile;
// the runs that cover tile t are numbered 0 .. m(t) - 1 in order (the partial-slice index), JS = max m(t); the run holding a
// tile's last chunk also writes the neutral rows of the unused slices.  Static, so the summation order is fixed.
static inline long fused_first_wg(long unit, long W, long T) { return ((unit + 1) * W - 1) / T; }  // the workgroup whose run holds `unit`
static inline void fused_balance(int tiles, int chunks, int slots, int *W_out, int *JS_out) {
  const long T = (long)tiles * chunks, W = T < slots ? T : slots;
  int js = 1;
  for (int t = 0; t < tiles; ++t) {
    const long m = fused_first_wg((long)t * chunks + chunks - 1, W, T) - fused_first_wg((long)t * chunks, W, T) + 1;
    js = m > js ? (int)m : js;
  }
  *W_out = (int)W;
  *JS_out = js;
}

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
#define DUST_FUSED_WGS 2  // resident workgroups per CU the register budget is set for (tools/fused_race.hip builds it at 1 too)
#endif
template <int MODE /* PAIR_K1 / PAIR_IMQ: the Stein kernel */, int DPB, bool STREAM_K = true>
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
  v2f *kv = reinterpret_cast<v2f *>(Ys + JC * YS);  // [TQ][KS] (prior logit -> softmax term, Stein kernel value): pass B fetches the
  float *kvf = reinterpret_cast<float *>(kv);       //           two weights of a (query, key) with one 8-byte read
  float *mrow = kvf + 2 * TQ * KS;  // [TQ] running max
  float *scl = mrow + TQ;           // [TQ] rescale factor of this chunk
  float *lrow = scl + TQ;           // [TQ] running sum of the softmax terms (relative to mrow)
  unsigned int *wany = reinterpret_cast<unsigned int *>(lrow + TQ);  // [4] per wave: some Stein kernel value of this chunk is non-zero
  unsigned int *pany = wany + 4;                                     // [4] per wave: some softmax term of this chunk is non-zero
  const int tid = threadIdx.x, D = a.D, N = a.N;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), jA = tid & 63;
  const int qg = tid / LCG, cg = tid - qg * LCG, c0 = CB * cg;  // pass-B ownership: queries qg + QS r, columns c0 .. c0 + CB - 1
  const bool pb = qg < TQ / 4;                                  // (DPB = 80: 240 of the 256 lanes)
  const int qgc = pb ? qg : 0;

  // this workgroup's run of units (fused_balance), one tile segment at a time
  const long T = (long)b.tiles * b.chunks;
  int unit = (int)((long)blockIdx.x * T / gridDim.x);
  const int unit_end = (int)(((long)blockIdx.x + 1) * T / gridDim.x);
  while (unit < unit_end) {
  const int tile = unit / b.chunks, ch0 = unit - tile * b.chunks, ch1 = min(b.chunks, ch0 + (unit_end - unit));
  const int js = (int)blockIdx.x - (int)((((long)tile * b.chunks + 1) * gridDim.x - 1) / T);  // ordinal of this run within the tile
  const int ib = a.i0 + tile * TQ;  // first query (global index)
  const int jbeg = ch0 * JC, jend = min(N, ch1 * JC);
  unit += ch1 - ch0;

  v4f xB[4][NV] /* -x_i */, accA[4][NV], accB[4][NV];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int gi = min(ib + qgc + QS * r, N - 1);
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      xB[r][u] = -*reinterpret_cast<const v4f *>(b.Xp + (size_t)gi * DPB + c0 + 4 * u);  // NEGATED: y + (-x) is one v_pk_add_f32 per pair of columns
      asm volatile("" : "+v"(xB[r][u]));  // (opaque: otherwise the negation folds back into 4 scalar v_sub_f32 per difference)
      accA[r][u] = accB[r][u] = v4f{0.f, 0.f, 0.f, 0.f};
    }
  }
  // The running max starts at a lower bound of its final value known before the pass (pairwise_far.hpp): a max known from the start
  // is what lets the pre-pass prove that a unit's softmax terms are exact zeros.
  for (int i = tid; i < TQ; i += NT) {
    mrow[i] = b.m0 ? b.m0[min(ib + i, N - 1)] : -INFINITY;
    lrow[i] = 0.f;
  }
  // live chunks of this run: 64 flags per ballot (wave-uniform; every wave reads the same bytes)
  int fgb = ch0;
  unsigned long long fmask = 0ull;
  auto far_group = [&](const int base) {
    const int cidx = base + jA;
    const bool lv = cidx < ch1 && (b.far == nullptr || b.far[(size_t)tile * b.chunks + cidx] == 0);
    fgb = base;
    fmask = __ballot(lv);
  };
  auto next_live = [&](int from) {  // first live chunk >= from, or ch1
    while (from < ch1) {
      if (from >= fgb + 64) far_group(from);
      const unsigned long long m = fmask >> (from - fgb);
      if (m) return from + (int)__builtin_ctzll(m);
      from = fgb + 64;
    }
    return ch1;
  };
  far_group(ch0);

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
  int ci = next_live(ch0);
  if (ci < ch1) {
    keys_issue(ci * JC);
    keys_commit(ci * JC);
  }
  while (ci < ch1) {
    const int j0 = ci * JC;
    const int jc = min(JC, jend - j0);
    const float lm = lm_next;
    unsigned long long keymask = ~0ull;  // keys of this chunk that have a term at all (wave-uniform; set in pass A)
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
      // the unit's masks (pairwise_far.hpp): near queries, near keys.  A key without a near query in the tile gets NO term - logit
      // -inf, kernel value 0 - whatever query it is paired with (exact zeros at the exact threshold, below 2^-43 at the default one)
      typedef const unsigned int __attribute__((address_space(4))) * cu32;  // (written by an earlier launch: scalar loads)
      const bool masks = MODE == PAIR_K1 && b.qmask != nullptr;  // (the pre-pass speaks for K1 only: IMQ's kernel values never vanish)
      const cu32 qm = (cu32)(uintptr_t)(masks ? b.qmask + ((size_t)tile * b.chunks + ci) * 8 : nullptr);
      keymask = masks ? ((unsigned long long)qm[4] | ((unsigned long long)qm[5] << 32)) : ~0ull;
      const bool kval = jA < jc && ((keymask >> jA) & 1ull);
      bool wave_any = (MODE != PAIR_K1) || b.nz == nullptr;
      // the wave's queries with a NEAR key in this chunk (pairwise_far.hpp: bit q of the unit's mask; without the pre-pass: all).
      // The others have no term here at all: logit -inf, kernel value 0, Gram row not stored (flag 0: gram_score_kernel masks it).
      static_assert(QW <= 32, "query mask of a wave");
      constexpr unsigned int QALL = QW == 32 ? 0xffffffffu : ((1u << QW) - 1u);
      unsigned int near = QALL;
      if (masks) {
        const unsigned long long lo = (unsigned long long)qm[0] | ((unsigned long long)qm[1] << 32);
        const unsigned long long hi = (unsigned long long)qm[2] | ((unsigned long long)qm[3] << 32);
        const int sft = wave * QW;
        const unsigned long long sel = sft < 64 ? ((lo >> sft) | (sft ? hi << (64 - sft) : 0ull)) : (hi >> (sft - 64));
        near &= (unsigned int)sel;
      }
      for (unsigned int fq = QALL & ~near; fq; fq &= fq - 1u) {
        const int i = wave * QW + (int)__builtin_ctz(fq);
        kv[i * KS + jA] = v2f{-INFINITY, 0.f};
        if (b.nz && jA == 0) b.nz[(size_t)(j0 >> 6) * b.ldnz + tile * TQ + i] = 0;
      }
      while (near) {  // two near queries per trip (an odd one out runs twice: the same values stored twice)
        const int qa = (int)__builtin_ctz(near);
        near &= near - 1u;
        const int qb = near ? (int)__builtin_ctz(near) : qa;
        near &= near - 1u;
        const int i = wave * QW + qa, i2 = wave * QW + qb;  // wave-uniform
        // uniform addresses -> scalar loads.  The kernel also STORES to global memory inside this loop (the Gram rows), so a plain
        // load of Xp counts as clobberable and would become a per-lane vector load; the padded query copy is never written by
        // this kernel: address it through the constant address space, whose loads are invariant by definition.  (Scalar loads
        // return out of order, so each wait drains all of them: 3 waits per query pair.  Rotating two register sets so that a
        // batch lands under the previous one's FMAs changed nothing - the second wave of the SIMD already covers the latency.)
        typedef const v2f __attribute__((address_space(4))) * cv2;
        const cv2 xa = (cv2)(uintptr_t)(b.Xp + (size_t)min(ib + i, N - 1) * DPB);
        const cv2 xb = (cv2)(uintptr_t)(b.Xp + (size_t)min(ib + i2, N - 1) * DPB);
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
        kv[i * KS + jA] = v2f{kval ? lm - 0.5f * pa : -INFINITY, ka};
        kv[i2 * KS + jA] = v2f{kval ? lm - 0.5f * pbq : -INFINITY, kb};
        // Gram matrix rows for pass 2: one 256-byte run per query and wave, streamed past the caches when pass 2 will read them
        // from HBM anyway (a 1 GB matrix at N = 16384: -7 % on this kernel).  Unconditional - no exec-mask branches inside the
        // distance loop (-6 %): K holds gridDim.x * TQ rows of ldK >= 64 ceil(N / 64) floats, so the rows behind n_local and the
        // columns behind N exist (the latter receive 0)
        const int il = tile * TQ + i, il2 = tile * TQ + i2;
        float *ka_p = &b.K[(size_t)il * b.ldK + j0 + jA], *kb_p = &b.K[(size_t)il2 * b.ldK + j0 + jA];
        // exact zeros (see PairFusedArgs): a row of 64 zero kernel values is flagged
        bool anya = true, anyb = true;
        if (MODE == PAIR_K1 && b.nz) {
          anya = __ballot(ka != 0.f) != 0ull;
          anyb = __ballot(kb != 0.f) != 0ull;
          wave_any = wave_any || anya || anyb;
        }
        if (b.nz && jA == 0) {
          unsigned char *fz = b.nz + (size_t)(j0 >> 6) * b.ldnz;
          fz[il] = anya ? 1 : 0;
          fz[il2] = anyb ? 1 : 0;
        }
        if (STREAM_K) {
          __builtin_nontemporal_store(ka, ka_p);
          __builtin_nontemporal_store(kb, kb_p);
        } else {
          *ka_p = ka;
          *kb_p = kb;
        }
      }
      if (jA == 0) wany[wave] = wave_any ? 1u : 0u;
    }
    wg_sync();
    {
      // online softmax over key chunks: LQ consecutive lanes per query (DPP max, bare v_exp_f32 as in the 32 x 64 kernel)
      const int q = tid / LQ, l = tid - q * LQ;
      bool psome = false;
      if (q < TQ) {
        float m = -INFINITY;
#pragma unroll
        for (int t = 0; t < JC / LQ; ++t) m = fmaxf(m, kvf[2 * (q * KS + l + LQ * t)]);
        m = LQ == 8 ? oct_max(m) : (LQ == 4 ? quad_max(m) : pair_max(m));
        const float mo = mrow[q];
        const float mn = fmaxf(mo, m);
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < JC / LQ; ++t) {
          const int jj = l + LQ * t;
          const float lg = kvf[2 * (q * KS + jj)];
          const float e = (mn == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((lg - mn) * 1.44269504088896340736f);
          kvf[2 * (q * KS + jj)] = e;  // (a 4-byte store: the kernel value beside it stays)
          sum += e;
        }
        sum = LQ == 8 ? oct_sum(sum) : (LQ == 4 ? quad_sum(sum) : pair_sum(sum));  // the chunk's mass of this query
        psome = sum != 0.f || b.nz == nullptr;  // (DUST_DENSE: everything is evaluated)
        if (l == 0) {  // the LQ lanes of a query run in lockstep: all have read mrow[q] by now
          const float sc = (mo == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((mo - mn) * 1.44269504088896340736f);
          mrow[q] = mn;
          scl[q] = sc;
          lrow[q] = lrow[q] * sc + sum;
        }
      }
      {
        const bool wsome = __ballot(psome) != 0ull;
        if (jA == 0) pany[wave] = wsome ? 1u : 0u;
      }
      wg_sync();
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float sc = scl[qgc + QS * r];
#pragma unroll
        for (int u = 0; u < NV; ++u) accA[r][u] *= sc;
      }
    }
    const int cn = next_live(ci + 1);
    const bool more = cn < ch1;
    if (more) keys_issue(cn * JC);  // in flight during pass B
    // ---- pass B: lane = 4 queries x CB columns; the difference y_j - x_i feeds the prior sum and the repulsion sum ----
    // (a chunk whose Stein kernel values are zero for the whole tile runs without the repulsion FMAs: they would add exact zeros)
    // (and a chunk whose softmax terms are zero for the whole tile - after the first tick the mixture weights are one-hot: every
    //  chunk but the heavy particle's - runs without the prior FMAs)
    const bool tile_any = (wany[0] | wany[1] | wany[2] | wany[3]) != 0u;
    const bool tile_pany = (pany[0] | pany[1] | pany[2] | pany[3]) != 0u;
    auto pass_b = [&](auto with_p, auto with_k) {
      constexpr bool WP = decltype(with_p)::value, WK = decltype(with_k)::value;
      auto one_key = [&](const int jj) {
        v4f yv[NV];
#pragma unroll
        for (int u = 0; u < NV; ++u) yv[u] = *reinterpret_cast<const v4f *>(&Ys[jj * YS + c0 + 4 * u]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const v2f wk = kv[(qg + QS * r) * KS + jj];
          const float wp = wk.x, ks = wk.y;
          // -k' of stein.hpp's pass B: accB += k' (x_i - y_j) = (-k') (y_j - x_i), the same product bit for bit
          const float nk = (MODE == PAIR_K1) ? ks : (ks * ks) * ks;
#pragma unroll
          for (int u = 0; u < NV; ++u) {
            const v4f diff = yv[u] + xB[r][u];  // y_j - x_i
            if (WP) accA[r][u] = __builtin_elementwise_fma(v4f{wp, wp, wp, wp}, diff, accA[r][u]);
            if (WK) accB[r][u] = __builtin_elementwise_fma(v4f{nk, nk, nk, nk}, diff, accB[r][u]);
          }
        }
      };
      // (the keys without a term carry exact zero weights: left out.  In a set with near-duplicates scattered through it nearly every
      //  unit has a near pair, but a key has a near query among the tile's 96 only now and then - 12 % of the keys at cfg4 after 140 ticks)
      for (unsigned long long km = keymask; km; km &= km - 1ull) one_key((int)__builtin_ctzll(km));
    };
    if (pb) {
      if (tile_pany && tile_any) pass_b(std::true_type{}, std::true_type{});
      else if (tile_pany) pass_b(std::true_type{}, std::false_type{});
      else if (tile_any) pass_b(std::false_type{}, std::true_type{});
    }
    wg_sync();  // pass B is done with Ys / kv
    if (more) keys_commit(cn * JC);
    ci = cn;
  }

  // ---- partial outputs (layout of stein.hpp: [js][n_local][ldp], raw coordinates) ----
  {
    // (lane coordinates re-derived from an opaque copy of tid: otherwise the output offsets, invariant across the segments, are
    //  hoisted out of the segment loop and held - spilled - across the chunk loop)
    int tid_e = tid;
    asm volatile("" : "+v"(tid_e));
    const int qg_e = tid_e / LCG, cg_e = tid_e - qg_e * LCG, c0_e = CB * cg_e;
    if (qg_e < TQ / 4) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int il = tile * TQ + qg_e + QS * r;
        if (il >= a.n_local) continue;
        const size_t row = ((size_t)js * a.n_local + il) * b.ldp;
#pragma unroll
        for (int u = 0; u < NV; ++u)
          if (c0_e + 4 * u < b.ldp) {
            *reinterpret_cast<v4f *>(a.pA + row + c0_e + 4 * u) = accA[r][u];
            *reinterpret_cast<v4f *>(b.pB + row + c0_e + 4 * u) = accB[r][u];
          }
        if (cg_e == 0) {
          a.pM[(size_t)js * a.n_local + il] = mrow[qg_e + QS * r];
          a.pL[(size_t)js * a.n_local + il] = lrow[qg_e + QS * r];
        }
      }
    }
  }
  if (ch1 == b.chunks) {  // the tile is complete: neutral rows for the slices it does not use (-inf / 0 mass, zero sums)
    const int rows = min(TQ, a.n_local - tile * TQ), l4 = b.ldp / 4;
    for (int k = js + 1; k < a.JS; ++k) {
      const size_t r0 = (size_t)k * a.n_local + (size_t)tile * TQ;
      for (int idx = tid; idx < rows * l4; idx += NT) {
        *reinterpret_cast<v4f *>(a.pA + r0 * b.ldp + 4 * (size_t)idx) = v4f{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<v4f *>(b.pB + r0 * b.ldp + 4 * (size_t)idx) = v4f{0.f, 0.f, 0.f, 0.f};
      }
      for (int idx = tid; idx < rows; idx += NT) {
        a.pM[r0 + idx] = -INFINITY;
        a.pL[r0 + idx] = 0.f;
      }
    }
  }
  wg_sync();  // the next segment re-initialises mrow / lrow and refills Ys
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
  const unsigned char *nz;  // [chunks][ldnz] non-zero flags of (key chunk, query row) written by pass 1, or nullptr: dense
  int ldnz;
};

template <int DPB>
static inline size_t gram_score_lds_bytes() {
  return sizeof(float) * ((size_t)PAIR_JC * (DPB + 4) + 64 * (size_t)(PAIR_JC + 4) + 64);  // + 2 048-chunk bitmap
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
  auto products = [&]() {
#pragma unroll 4
    for (int k4 = 0; k4 < JC / 4; ++k4) {
      const float bq = Kt[(wave * 16 + (jA & 15)) * KS2 + 4 * k4 + (jA >> 4)];
      float as[NCT];
#pragma unroll
      for (int t = 0; t < NCT; ++t) as[t] = Vs[(4 * k4 + (jA >> 4)) * YS + 16 * t + (jA & 15)];
#pragma unroll
      for (int t = 0; t < NCT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(as[t], bq, acc[t], 0, 0, 0);
    }
  };
  unsigned int *nzm = reinterpret_cast<unsigned int *>(Kt + 64 * KS2);  // [64] bitmap: chunks of this slice with a non-zero block
  const int ch0 = jbeg >> 6, nch = (jend - jbeg + JC - 1) >> 6;
  const bool sparse = a.nz != nullptr && nch <= 2048;
  if (sparse) {
    if (tid < 64) nzm[tid] = 0u;
    wg_sync();
    for (int ci = tid >> 4; ci < nch; ci += NT / 16) {  // 16 lanes x 4 flag bytes = the tile's 64 rows of one chunk
      const uint32_t w = *reinterpret_cast<const uint32_t *>(a.nz + (size_t)(ch0 + ci) * a.ldnz + il0 + 4 * (tid & 15));
      if (w) atomicOr(&nzm[ci >> 5], 1u << (ci & 31));
    }
    wg_sync();
  }
  auto next_chunk = [&](int ci) {  // first chunk >= ci with a non-zero block (nch: none); uniform
    if (!sparse) return ci;
    while (ci < nch) {
      const uint32_t w = nzm[ci >> 5] >> (ci & 31);
      if (w) return ci + __builtin_ctz(w);
      ci = (ci | 31) + 1;
    }
    return nch;
  };
  // (sparse: a row whose flag is 0 holds zeros - or, where pairwise_far.hpp kept pass 1 away from the unit, nothing at all: masked)
  unsigned int kfl = 0xfu;
  auto k_issue = [&](const int j0, v4f (&kt)[4]) {
    kfl = 0u;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int il = min(il0 + kr + 16 * u, a.n_local - 1);
      kt[u] = *reinterpret_cast<const v4f *>(a.K + (size_t)il * a.ldK + j0 + kc);  // (ldK is a multiple of 64: in bounds; the tail is masked)
      kfl |= (a.nz == nullptr || a.nz[(size_t)(j0 >> 6) * a.ldnz + il] != 0) ? 1u << u : 0u;
    }
  };
  auto k_commit = [&](const int jc, const v4f (&kt)[4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      v4f t = (kfl >> u) & 1u ? kt[u] : v4f{0.f, 0.f, 0.f, 0.f};
      t.x = kc + 0 < jc ? t.x : 0.f;
      t.y = kc + 1 < jc ? t.y : 0.f;
      t.z = kc + 2 < jc ? t.z : 0.f;
      t.w = kc + 3 < jc ? t.w : 0.f;
      *reinterpret_cast<v4f *>(&Kt[(kr + 16 * u) * KS2 + kc]) = t;
    }
  };
  {
    // The score chunk is one contiguous 64 x D float run from a 16-byte aligned offset (j0 is a multiple of 64): NLV b128 loads per
    // lane (5 at D = 80; the row-per-lane-group staging of stein.hpp needs 32 dword registers).  Few enough registers to hold the
    // NEXT chunk's K tile and score pieces across the products: their HBM / L2 latency runs under the MFMAs instead of opening
    // every chunk (the products alone bound this kernel: 443 of 456 us with the K loads removed, tools/gram_probe.hip).
    // D % 4 == 0: a piece never straddles two rows and is committed with one b128 write; otherwise element by element.
    constexpr int NLV = (JC * DPB / 4 + NT - 1) / NT;
    const bool vec4 = (D & 3) == 0;
    const uint32_t magicD = (uint32_t)((1ull << 32) / (uint64_t)D) + 1u;
    for (int e = tid; e < JC * (DPB - D); e += NT) {  // columns D .. DPB - 1: never staged, read as zeros
      const int r = e / (DPB - D);
      Vs[r * YS + D + (e - r * (DPB - D))] = 0.f;
    }
    v4f kt[4], vq[NLV];
    auto v_issue = [&](const int j0) {
      const int nval = min(JC, jend - j0) * D;  // floats of the chunk
      const float *src = a.V + (size_t)j0 * D;
#pragma unroll
      for (int u = 0; u < NLV; ++u) {
        const int e = 4 * (tid + NT * u);
        if (e + 3 < nval) vq[u] = *reinterpret_cast<const v4f *>(src + e);
        else {  // the piece that holds the end of the chunk (D % 4 != 0), and the ones behind it
          vq[u].x = e + 0 < nval ? src[e + 0] : 0.f;
          vq[u].y = e + 1 < nval ? src[e + 1] : 0.f;
          vq[u].z = e + 2 < nval ? src[e + 2] : 0.f;
          vq[u].w = 0.f;
        }
      }
    };
    int ci = next_chunk(0);  // chunks whose block is zero for all 64 rows are skipped: they would add exact zeros
    if (ci < nch) {
      v_issue(jbeg + ci * JC);
      k_issue(jbeg + ci * JC, kt);
    }
    while (ci < nch) {
      const int j0 = jbeg + ci * JC;
      const int jc = min(JC, jend - j0);
      wg_sync();  // the previous chunk's products are done with Vs / Kt
#pragma unroll
      for (int u = 0; u < NLV; ++u) {
        const int e = 4 * (tid + NT * u);
        if (vec4) {
          const int row = (int)__umulhi((uint32_t)e, magicD), col = e - row * D;
          if (row < JC) *reinterpret_cast<v4f *>(&Vs[row * YS + col]) = vq[u];  // (pieces behind the slice were loaded as zeros)
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int row = (int)__umulhi((uint32_t)(e + q), magicD), col = e + q - row * D;
            if (row < JC) Vs[row * YS + col] = vq[u][q];
          }
        }
      }
      k_commit(jc, kt);
      wg_sync();
      ci = next_chunk(ci + 1);
      if (ci < nch) {  // in flight during the products
        v_issue(jbeg + ci * JC);
        k_issue(jbeg + ci * JC, kt);
      }
      products();
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
