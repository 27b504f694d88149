// pairwise_packed.hpp - the large-set pairwise passes of one SVGD iteration over RUN LISTS (round 6; svmpc.py:38-41 prior gradient,
// 76-83 Stein kernel).  pairwise_far.hpp decides, per (query tile, 64-key chunk) unit, which queries and which keys have a pair whose
// terms can matter.  On a set that has been moved around for a while (cfg4 after ~30 ticks at lr = 100) near-duplicates are scattered
// through the whole index range: nearly every unit has SOME near pair (0.1 % far units), but only ~8 of a unit's 64 keys and ~8 of its
// 96 queries take part.  Visiting every unit to find it nearly empty cost 9.5 us per unit and workgroup (staging 64 key rows, filling
// 6 144 weight slots with "no term", 6 144 exponentials, four barriers) - 0.81 of the 2.2 ms tick.  Here the pre-pass' key masks are
// turned into one list of near keys per query tile (far_pack_kernel), cut into PACKED units of 64 keys, and both passes walk the lists:
//
//   far_pack_kernel        per tile: near-key list kidx (ascending), unit offsets, the near-query mask of every unit (the union over
//                          the chunks its keys come from), the slice boundaries of the two passes below
//   pairwise_packed_kernel pass 1 (pairwise_fused.hpp's arithmetic, unchanged): one exact-difference distance per (near query, packed
//                          key) serves the prior logit and the Stein kernel value; prior partials, repulsion partials, and the
//                          kernel values of the unit as a [TQ][64] block
//   gram_packed_kernel     pass 2: pA = K x score over the packed units on the matrix cores (v_mfma_f32_16x16x4_f32), score rows
//                          gathered through the same lists
//
// Two list forms.  MERGED (the default threshold): the near keys of consecutive chunks share units - 8x fewer units on the aged cfg4
// set, every key row staged is a key that matters.  Sums over a tile's keys are then grouped differently from the chunk-by-chunk
// evaluation (chunk-wise online softmax, 4-key MFMA steps): results agree to rounding, not bit for bit - as the default threshold
// already does (pairwise_far.hpp: terms below 2^-43 left out).  PLAIN (DUST_FAR_T >= 224, DUST_FAR=0, DUST_DENSE=1, IMQ): one unit per
// live chunk holding ALL its keys at their own lane positions, slices cut at fixed chunk positions - the arithmetic and its order are
// those of visiting every chunk, minus the chunks whose terms are exact zeros: the three modes stay bit-identical to one another.
#pragma once
#include "pairwise_fused.hpp"
#include "pairwise_logp_mfma.hpp"

namespace dust {

struct PackArgs {
  int N, tiles, chunks;
  int merge;                  // 1: MERGED lists, 0: PLAIN (see above)
  int JS, JSG;                // slices of pass 1 / pass 2 per tile
  const unsigned char *far;   // [tiles][chunks] 1 = the unit contributes nothing, or nullptr: every chunk is live
  const unsigned int *qmask;  // [tiles][chunks][8] pairwise_far.hpp's masks (words 0-3 queries, 4-5 keys), or nullptr: all near
  int *kidx;                  // [tiles][ldi] key indices, ascending
  int ldi;                    // chunks * 64
  int *uoff;                  // [tiles][chunks + 1] first list slot of unit u; uoff[U] = list length
  unsigned int *uq;           // [tiles][chunks][4] near-query mask of unit u (bit q = query q of the tile)
  int *soff, *goff;           // [tiles][JS + 1] / [tiles][JSG + 1] unit ranges of the slices
  unsigned int *total;        // [2] {units, tiles x chunks} of this launch (atomicAdd; zeroed by pairwise_far.hpp's row kernel), or nullptr
};

static inline size_t far_pack_lds_bytes(int chunks) { return sizeof(int) * ((size_t)3 * chunks + 2 + 4 * (size_t)chunks + 2 * (size_t)chunks + 64); }

// one workgroup (1024 lanes: the list loop walks a chunk per wave - 16 trips at N = 16 384 instead of 64 with four waves, 13 -> ~8 us) per
// query tile; chunks <= 1024 (N <= 65536)
__global__ __launch_bounds__(1024) void far_pack_kernel(const PackArgs a) {
  constexpr int NT = 1024, NW = NT / 64;
  extern __shared__ int pk_lds[];
  int *cnt = pk_lds;                       // [chunks] keys of the chunk that enter the list
  int *koff = cnt + a.chunks;              // [chunks + 1] exclusive prefix of cnt
  int *uid = koff + a.chunks + 1;          // [chunks + 1] PLAIN: exclusive prefix of (cnt > 0) = the unit a live chunk becomes
  unsigned int *uql = reinterpret_cast<unsigned int *>(uid + a.chunks + 1);  // [chunks][4] MERGED: unit masks under construction
  unsigned int *kml = uql + 4 * a.chunks;  // [chunks][2] the listed keys of each chunk (every lane fetches its chunk's words at once:
                                           //  read chunk by chunk in the list loop below, 64 dependent global loads per wave were 20 us)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tile = blockIdx.x;
  const unsigned char *far = a.far ? a.far + (size_t)tile * a.chunks : nullptr;
  const unsigned int *qm = a.qmask ? a.qmask + (size_t)tile * a.chunks * 8 : nullptr;
  const bool use_keys = a.merge && qm != nullptr;
  for (int c = tid; c < a.chunks; c += NT) {
    const int nk = min(64, a.N - 64 * c);
    const unsigned long long valid = nk >= 64 ? ~0ull : ((1ull << nk) - 1ull);
    unsigned long long km = valid;
    if (use_keys) km = ((unsigned long long)qm[8 * c + 4] | ((unsigned long long)qm[8 * c + 5] << 32)) & valid;
    if (far && far[c]) km = 0ull;
    kml[2 * c] = (unsigned int)km;
    kml[2 * c + 1] = (unsigned int)(km >> 32);
    cnt[c] = (int)__builtin_popcountll(km);
    uql[4 * c + 0] = uql[4 * c + 1] = uql[4 * c + 2] = uql[4 * c + 3] = 0u;
  }
  wg_sync();
  if (wave == 0) {  // exclusive prefixes: each lane a contiguous segment, the segment sums through a wave scan
    const int seg = (a.chunks + 63) / 64, c0 = lane * seg, c1 = min(a.chunks, c0 + seg);
    int s = 0, f = 0;
    for (int c = c0; c < c1; ++c) {
      s += cnt[c];
      f += cnt[c] > 0 ? 1 : 0;
    }
    int ps = s, pf = f;
    for (int o = 1; o < 64; o <<= 1) {
      const int ts = __shfl_up(ps, o), tf = __shfl_up(pf, o);
      if (lane >= o) {
        ps += ts;
        pf += tf;
      }
    }
    int es = ps - s, ef = pf - f;  // exclusive
    for (int c = c0; c < c1; ++c) {
      koff[c] = es;
      uid[c] = ef;
      es += cnt[c];
      ef += cnt[c] > 0 ? 1 : 0;
    }
    if (lane == 63) {
      koff[a.chunks] = ps;
      uid[a.chunks] = pf;
    }
  }
  wg_sync();
  const int L = koff[a.chunks];
  const int U = a.merge ? (L + 63) / 64 : uid[a.chunks];
  int *kidx = a.kidx + (size_t)tile * a.ldi;
  int *uoff = a.uoff + (size_t)tile * (a.chunks + 1);
  // the list: one wave per chunk, lane = key of the chunk (its rank among the chunk's listed keys is its slot)
  for (int c = wave; c < a.chunks; c += NW) {
    const unsigned long long km = (unsigned long long)kml[2 * c] | ((unsigned long long)kml[2 * c + 1] << 32);
    if ((km >> lane) & 1ull) kidx[koff[c] + (int)__builtin_popcountll(km & ((1ull << lane) - 1ull))] = 64 * c + lane;
  }
  // units and their query masks
  if (a.merge) {
    for (int u = tid; u <= U; u += NT) uoff[u] = min(64 * u, L);
    for (int c = tid; c < a.chunks; c += NT) {
      if (cnt[c] == 0) continue;
      const int ua = koff[c] >> 6, ub = (koff[c] + cnt[c] - 1) >> 6;
      for (int u = ua; u <= ub; ++u)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const unsigned int m = qm ? qm[8 * c + w] : 0xffffffffu;
          if (m) atomicOr(&uql[4 * u + w], m);
        }
    }
    wg_sync();
    unsigned int *uq = a.uq + (size_t)tile * a.chunks * 4;
    for (int e = tid; e < 4 * U; e += NT) uq[e] = uql[e];
  } else {
    unsigned int *uq = a.uq + (size_t)tile * a.chunks * 4;
    for (int c = tid; c < a.chunks; c += NT) {
      if (cnt[c] == 0) continue;
      const int u = uid[c];
      uoff[u] = koff[c];
#pragma unroll
      for (int w = 0; w < 4; ++w) uq[4 * u + w] = qm ? qm[8 * c + w] : 0xffffffffu;
    }
    if (tid == 0) uoff[U] = L;
  }
  // slices.  MERGED: equal shares of the tile's units.  PLAIN: cut at fixed CHUNK positions, so that the partial sums of a slice are
  // those of the same chunks whichever of them are live (bit-identity across DUST_FAR_T=224 / DUST_FAR=0 / DUST_DENSE=1).
  for (int k = tid; k <= a.JS; k += NT) a.soff[(size_t)tile * (a.JS + 1) + k] = a.merge ? (int)((long)k * U / a.JS) : uid[(int)((long)k * a.chunks / a.JS)];
  for (int k = tid; k <= a.JSG; k += NT) a.goff[(size_t)tile * (a.JSG + 1) + k] = a.merge ? (int)((long)k * U / a.JSG) : uid[(int)((long)k * a.chunks / a.JSG)];
  if (tid == 0 && a.total) {
    atomicAdd(a.total + 0, (unsigned int)U);
    atomicAdd(a.total + 1, (unsigned int)a.chunks);
  }
}

struct PairPackedArgs {
  PairArgs p;       // the PRIOR's arguments (X = Y = theta, logmix, pA / pM / pL, JS, i0, n_local); inv_s unused
  const float *Xp;  // [N][DPB] zero-padded copy of the particles
  int ldp;          // row stride of the partial outputs
  float wP[2];      // 1 / sigma_p^2 for even / odd dimensions
  float wS[2];      // 1 / ell^2 (both)
  float *pB;        // [JS][n_local][ldp] repulsion partials
  float *Kp;        // [tiles][umax][TQ][64] Stein kernel values of the units (rows of queries outside a unit's mask are not written)
  int tiles, umax;  // umax = chunks: units per tile the buffers are laid out for
  const int *kidx;
  int ldi;
  const int *uoff;
  const unsigned int *uq;  // nullptr: every query of every unit is near
  const int *soff;
  unsigned char *nzu;  // [tiles][umax] 1 = some Stein kernel value of the unit is not exactly 0 (pass 2 skips the others), or nullptr
  const float *m0;     // [N] where each query's running max starts (pairwise_far.hpp), or nullptr: -inf
  const int *qperm;    // [n_local] the rank's queries in TILE ORDER (position -> local row; query_order_kernel below), or nullptr: identity
  int *lead;           // [n_local] by local row: the smallest key index with a kernel value above PACK_LEAD_K (atomicMin), or nullptr
};

// Tile order.  A tile's run list is the union of its 96 queries' near keys.  Near-duplicates are scattered over the index range, so
// 96 consecutive particles are near 96 different groups of keys: 1 380 listed keys per tile on the aged cfg4 set, 38 % of a unit's
// queries near one of its keys.  Tiles of queries that share their near keys need far shorter lists (host analysis of the same set,
// tools/far_granularity.py: 355 keys per tile, 3.6x fewer units, 77 % of a unit's queries near).  So the queries are walked in the
// order of their LEADER - the smallest key index with a non-negligible kernel value, which pass 1 notes on the way (one ballot per
// query and unit) - sorted once per tick behind pass 2, for the NEXT tick's tiles: the particles move little between ticks and the
// order only groups the work - every row's result is the same whatever tile it rides in, up to the regrouping of its key units.
#define PACK_LEAD_K 1.0e-27f  // k = exp(-d2 / 0.96) above this: d2 < 60 (the far pre-pass' default threshold)

template <int DPB>
static inline size_t pairwise_packed_lds_bytes() {
  return pairwise_fused_lds_bytes<DPB>();
}

template <int MODE /* PAIR_K1 / PAIR_IMQ: the Stein kernel */, int DPB, bool STREAM_K>
__global__ __launch_bounds__(PAIR_NT, DUST_FUSED_WGS) void pairwise_packed_kernel(const PairPackedArgs b) {
  using G = FusedGeom<DPB>;
  constexpr int JC = PAIR_JC, NT = PAIR_NT, TQ = G::TQ, YS = G::YS, KS = G::KS, CB = G::CB, LCG = G::LCG, NV = CB / 4;
  constexpr int QW = TQ / 4;  // queries per wave in pass A
  constexpr int QS = TQ / 4;  // pass-B ownership: lane group qg holds queries qg + QS r (pairwise_fused.hpp)
  constexpr int LQ = NT / TQ >= 8 ? 8 : (NT / TQ >= 4 ? 4 : 2);  // lanes per query in the softmax step
  static_assert(MODE == PAIR_K1 || MODE == PAIR_IMQ, "Stein kernel family");
  static_assert(QW <= 32, "query mask of a wave");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const PairArgs &a = b.p;
  float *Ys = lds;                                  // [JC][YS] the unit's keys (raw coordinates, zero padded)
  v2f *kv = reinterpret_cast<v2f *>(Ys + JC * YS);  // [TQ][KS] (prior logit -> softmax term, Stein kernel value)
  float *kvf = reinterpret_cast<float *>(kv);
  float *mrow = kvf + 2 * TQ * KS;  // [TQ] running max
  float *scl = mrow + TQ;           // [TQ] rescale factor of this unit
  float *lrow = scl + TQ;           // [TQ] running sum of the softmax terms (relative to mrow)
  unsigned int *wany = reinterpret_cast<unsigned int *>(lrow + TQ);  // [4] per wave: some Stein kernel value of this unit is non-zero
  unsigned int *pany = wany + 4;                                     // [4] per wave: some softmax term of this unit is non-zero
  const int tid = threadIdx.x, N = a.N;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), jA = tid & 63;
  const int qg = tid / LCG, cg = tid - qg * LCG, c0 = CB * cg;
  const bool pb = qg < TQ / 4;
  const int qgc = pb ? qg : 0;
  const int tile = blockIdx.x, js = blockIdx.y;
  typedef const int __attribute__((address_space(4))) * ci32;  // (written by an earlier launch: scalar loads)
  // local row of the tile's query q (positions behind the rank's last row repeat it: computed, never stored)
  const int nl1 = a.n_local - 1;
  const int *qpv = b.qperm;
  const ci32 qps = (ci32)(uintptr_t)b.qperm;
  auto row_v = [&](const int q) { const int pos = min(tile * TQ + q, nl1); return qpv ? qpv[pos] : pos; };   // per lane
  auto row_s = [&](const int q) { const int pos = min(tile * TQ + q, nl1); return qps ? qps[pos] : pos; };   // wave-uniform
  typedef const unsigned int __attribute__((address_space(4))) * cu32;
  const ci32 soff = (ci32)(uintptr_t)(b.soff + (size_t)tile * (a.JS + 1));
  const ci32 uoff = (ci32)(uintptr_t)(b.uoff + (size_t)tile * (b.umax + 1));
  const int *kidx = b.kidx + (size_t)tile * b.ldi;
  const int u0 = soff[js], u1 = soff[js + 1];

  v4f xB[4][NV] /* -x_i */, accA[4][NV], accB[4][NV];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int gi = a.i0 + row_v(qgc + QS * r);
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      xB[r][u] = -*reinterpret_cast<const v4f *>(b.Xp + (size_t)gi * DPB + c0 + 4 * u);
      asm volatile("" : "+v"(xB[r][u]));
      accA[r][u] = accB[r][u] = v4f{0.f, 0.f, 0.f, 0.f};
    }
  }
  for (int i = tid; i < TQ; i += NT) {
    mrow[i] = b.m0 ? b.m0[a.i0 + row_v(i)] : -INFINITY;
    lrow[i] = 0.f;
  }

  // Key staging: the unit's rows are fetched THROUGH the list (16-byte pieces, NLD per lane).  The next unit's loads are issued
  // before pass B and committed to LDS after it.
  constexpr int NLD = (JC * DPB / 4 + NT - 1) / NT;
  v4f ky[NLD];
  float lm_next = 0.f;
  auto keys_issue = [&](const int u) {
    const int k0 = uoff[u], jc = uoff[u + 1] - k0;
    int key[NLD];
#pragma unroll
    for (int w = 0; w < NLD; ++w) {
      const int row = ((tid + NT * w) * 4) / DPB;
      key[w] = kidx[k0 + min(row, jc - 1)];  // rows past the unit: clamped, zeroed at the commit
    }
    const int kl = kidx[k0 + min(jA, jc - 1)];
#pragma unroll
    for (int w = 0; w < NLD; ++w) {
      const int f = tid + NT * w, row = (f * 4) / DPB;
      ky[w] = *reinterpret_cast<const v4f *>(b.Xp + (size_t)key[w] * DPB + (f * 4 - row * DPB));
    }
    lm_next = a.logmix[kl];
  };
  auto keys_commit = [&](const int u) {
    const int jc = uoff[u + 1] - uoff[u];
#pragma unroll
    for (int w = 0; w < NLD; ++w) {
      const int f = tid + NT * w;
      const int row = (f * 4) / DPB, col = f * 4 - row * DPB;
      if (row < JC) *reinterpret_cast<v4f *>(&Ys[row * YS + col]) = row < jc ? ky[w] : v4f{0.f, 0.f, 0.f, 0.f};
    }
  };
  if (u0 < u1) {
    keys_issue(u0);
    keys_commit(u0);
  }
  for (int u = u0; u < u1; ++u) {
    const int jc = uoff[u + 1] - uoff[u];
    const float lm = lm_next;
    wg_sync();  // Ys holds this unit
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
      bool wave_any = (MODE != PAIR_K1) || b.nzu == nullptr;
      constexpr unsigned int QALL = QW == 32 ? 0xffffffffu : ((1u << QW) - 1u);
      unsigned int near = QALL;
      if (b.uq) {
        const cu32 qm = (cu32)(uintptr_t)(b.uq + ((size_t)tile * b.umax + u) * 4);
        const unsigned long long lo = (unsigned long long)qm[0] | ((unsigned long long)qm[1] << 32);
        const unsigned long long hi = (unsigned long long)qm[2] | ((unsigned long long)qm[3] << 32);
        const int sft = wave * QW;
        const unsigned long long sel = sft < 64 ? ((lo >> sft) | (sft ? hi << (64 - sft) : 0ull)) : (hi >> (sft - 64));
        near &= (unsigned int)sel;
      }
      // queries without a near key in this unit have no term here: logit -inf, kernel value 0, kernel row not stored (pass 2 masks it)
      for (unsigned int fq = QALL & ~near; fq; fq &= fq - 1u) {
        const int i = wave * QW + (int)__builtin_ctz(fq);
        kv[i * KS + jA] = v2f{-INFINITY, 0.f};
      }
      float *kblk = b.Kp + ((size_t)tile * b.umax + u) * TQ * 64;
      while (near) {  // two near queries per trip (an odd one out runs twice: the same values stored twice)
        const int qa = (int)__builtin_ctz(near);
        near &= near - 1u;
        const int qb = near ? (int)__builtin_ctz(near) : qa;
        near &= near - 1u;
        const int i = wave * QW + qa, i2 = wave * QW + qb;  // wave-uniform
        typedef const v2f __attribute__((address_space(4))) * cv2;  // (scalar loads: see pairwise_fused_kernel)
        const int ra_ = row_s(i), rb_ = row_s(i2);
        const cv2 xa = (cv2)(uintptr_t)(b.Xp + (size_t)(a.i0 + ra_) * DPB);
        const cv2 xb = (cv2)(uintptr_t)(b.Xp + (size_t)(a.i0 + rb_) * DPB);
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
        if (MODE == PAIR_K1 && b.nzu) wave_any = wave_any || __ballot(ka != 0.f || kb != 0.f) != 0ull;
        if (MODE == PAIR_K1 && b.lead) {  // the queries' leaders (tile order of the next tick): the unit's first key that is close
          const unsigned long long la = __ballot(ka > PACK_LEAD_K), lb = __ballot(kb > PACK_LEAD_K);
          const int k0u = uoff[u];  // (the key of list slot k0u + lane: looked up by the one lane that reports it)
          if (la && jA == 0) atomicMin(b.lead + ra_, kidx[k0u + (int)__builtin_ctzll(la)]);
          if (lb && i2 != i && jA == 0) atomicMin(b.lead + rb_, kidx[k0u + (int)__builtin_ctzll(lb)]);
        }
        float *ka_p = kblk + (size_t)i * 64 + jA, *kb_p = kblk + (size_t)i2 * 64 + jA;
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
      // online softmax over the units: LQ consecutive lanes per query (DPP max, bare v_exp_f32)
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
          kvf[2 * (q * KS + jj)] = e;
          sum += e;
        }
        sum = LQ == 8 ? oct_sum(sum) : (LQ == 4 ? quad_sum(sum) : pair_sum(sum));
        psome = sum != 0.f || b.nzu == nullptr;  // (DUST_DENSE: everything is evaluated)
        if (l == 0) {
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
        for (int w = 0; w < NV; ++w) accA[r][w] *= sc;
      }
    }
    const bool more = u + 1 < u1;
    if (more) keys_issue(u + 1);  // in flight during pass B
    // ---- pass B: lane = 4 queries x CB columns; the difference y_j - x_i feeds the prior sum and the repulsion sum ----
    const bool tile_any = (wany[0] | wany[1] | wany[2] | wany[3]) != 0u;
    const bool tile_pany = (pany[0] | pany[1] | pany[2] | pany[3]) != 0u;
    if (b.nzu && tid == 0) b.nzu[(size_t)tile * b.umax + u] = tile_any ? 1 : 0;
    auto pass_b = [&](auto with_p, auto with_k) {
      constexpr bool WP = decltype(with_p)::value, WK = decltype(with_k)::value;
      // (skipping the keys whose weights are zero for all of a wave's queries was built and measured: 345 against 326 us at cfg4 - the
      //  near keys of a tile cluster, nearly every listed key carries a weight for some query of every wave)
      for (int jj = 0; jj < jc; ++jj) {
        v4f yv[NV];
#pragma unroll
        for (int w = 0; w < NV; ++w) yv[w] = *reinterpret_cast<const v4f *>(&Ys[jj * YS + c0 + 4 * w]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const v2f wk = kv[(qg + QS * r) * KS + jj];
          const float wp = wk.x, ks = wk.y;
          const float nk = (MODE == PAIR_K1) ? ks : (ks * ks) * ks;
#pragma unroll
          for (int w = 0; w < NV; ++w) {
            const v4f diff = yv[w] + xB[r][w];  // y_j - x_i
            if (WP) accA[r][w] = __builtin_elementwise_fma(v4f{wp, wp, wp, wp}, diff, accA[r][w]);
            if (WK) accB[r][w] = __builtin_elementwise_fma(v4f{nk, nk, nk, nk}, diff, accB[r][w]);
          }
        }
      }
    };
    if (pb) {
      if (tile_pany && tile_any) pass_b(std::true_type{}, std::true_type{});
      else if (tile_pany) pass_b(std::true_type{}, std::false_type{});
      else if (tile_any) pass_b(std::false_type{}, std::true_type{});
    }
    wg_sync();  // pass B is done with Ys / kv
    if (more) keys_commit(u + 1);
  }

  // ---- partial outputs (layout of stein.hpp: [js][n_local][ldp], raw coordinates) ----
  wg_sync();
  {
    int tid_e = tid;
    asm volatile("" : "+v"(tid_e));
    const int qg_e = tid_e / LCG, cg_e = tid_e - qg_e * LCG, c0_e = CB * cg_e;
    if (qg_e < TQ / 4) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (tile * TQ + qg_e + QS * r >= a.n_local) continue;
        const int il = row_v(qg_e + QS * r);
        const size_t row = ((size_t)js * a.n_local + il) * b.ldp;
#pragma unroll
        for (int w = 0; w < NV; ++w)
          if (c0_e + 4 * w < b.ldp) {
            *reinterpret_cast<v4f *>(a.pA + row + c0_e + 4 * w) = accA[r][w];
            *reinterpret_cast<v4f *>(b.pB + row + c0_e + 4 * w) = accB[r][w];
          }
        if (cg_e == 0) {
          a.pM[(size_t)js * a.n_local + il] = mrow[qg_e + QS * r];
          a.pL[(size_t)js * a.n_local + il] = lrow[qg_e + QS * r];
        }
      }
    }
  }
}

// ---- pass 2: pA[js][i][:] = sum over the units of slice js of K_unit[i][:] x score[keys of the unit][:] ---------------------------
struct GramPackedArgs {
  int N, D, n_local, JS, ldp, tiles, umax;
  const float *Kp;
  const float *V;  // [N][D] score
  float *pA;       // [JS][n_local][ldp]
  const int *kidx;
  int ldi;
  const int *uoff;
  const int *qperm;          // pass 1's tile order (position -> local row), or nullptr: identity
  const unsigned int *uq;    // rows outside a unit's mask were not written by pass 1: read as zeros (nullptr: all written)
  const int *goff;
  const unsigned char *nzu;  // units whose kernel values are all exactly 0 are skipped (nullptr: dense)
};

template <int TQ>
struct GramGeom {
  static constexpr int RT = TQ / 16;  // 16-query MFMA row tiles: one per wave
  static constexpr int NT = 64 * RT;
};

template <int DPG, int TQ>
static inline size_t gram_packed_lds_bytes() {
  return sizeof(float) * ((size_t)PAIR_JC * (DPG + 4) + (size_t)TQ * (PAIR_JC + 4));
}

// Workgroup = one query tile of pass 1 (TQ rows: 6 - 8 waves) x one slice of its units; wave w owns row tile w and all DPG / 16
// column tiles: per 4-key step one read of the kernel block and DPG / 16 of the score rows feed DPG / 16 MFMAs.
// D'[col][query] += V^T[col][key] K^T[key][query] as in gram_score_kernel (A operand: a score column block, B operand: kernel rows).
template <int DPG, int TQ>
__global__ __launch_bounds__(GramGeom<TQ>::NT, 2) void gram_packed_kernel(const GramPackedArgs a) {
  using GG = GramGeom<TQ>;
  constexpr int JC = PAIR_JC, NT = GG::NT, YS = DPG + 4, KS2 = JC + 4, NCT = DPG / 16;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *Vs = lds;           // [JC][YS] score rows of the unit's keys
  float *Kt = Vs + JC * YS;  // [TQ][KS2] kernel rows of the unit (query-major)
  const int tid = threadIdx.x, D = a.D;
  const int tile = blockIdx.x, js = blockIdx.y;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), jA = tid & 63, r16 = jA & 15, g = jA >> 4;
  typedef const int __attribute__((address_space(4))) * ci32;
  typedef const unsigned int __attribute__((address_space(4))) * cu32;
  const ci32 goff = (ci32)(uintptr_t)(a.goff + (size_t)tile * (a.JS + 1));
  const ci32 uoff = (ci32)(uintptr_t)(a.uoff + (size_t)tile * (a.umax + 1));
  const int *kidx = a.kidx + (size_t)tile * a.ldi;
  const int u0 = goff[js], u1 = goff[js + 1];
  const int rt0 = wave;
  v4f acc0[NCT];
#pragma unroll
  for (int t = 0; t < NCT; ++t) acc0[t] = v4f{0.f, 0.f, 0.f, 0.f};
  const bool vec4 = (D & 3) == 0;
  const int d4 = D >> 2;
  constexpr int NLV = (JC * (DPG / 4) + NT - 1) / NT;  // score pieces per lane (16 bytes each) when D % 4 == 0
  constexpr int NLK = (TQ * (JC / 4) + NT - 1) / NT;   // kernel-block pieces per lane
  for (int e = tid; e < JC * (DPG - D); e += NT) {  // columns D .. DPG - 1: never staged, read as zeros
    const int r = e / (DPG - D);
    Vs[r * YS + D + (e - r * (DPG - D))] = 0.f;
  }
  auto next_unit = [&](int u) {  // first unit >= u with a non-zero kernel block (u1: none); uniform
    if (!a.nzu) return u;
    while (u < u1 && a.nzu[(size_t)tile * a.umax + u] == 0) ++u;
    return u;
  };
  v4f vq[NLV], kt[NLK];
  auto issue = [&](const int u) {
    const int k0 = uoff[u], jc = uoff[u + 1] - k0;
    if (vec4) {
#pragma unroll
      for (int w = 0; w < NLV; ++w) {
        const int f = tid + NT * w, row = f / d4, col4 = f - row * d4;
        vq[w] = v4f{0.f, 0.f, 0.f, 0.f};
        if (row < jc) vq[w] = *reinterpret_cast<const v4f *>(a.V + (size_t)kidx[k0 + row] * D + 4 * col4);
      }
    }
    const float *kblk = a.Kp + ((size_t)tile * a.umax + u) * TQ * 64;
    unsigned int qm[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    if (a.uq) {
      const cu32 m = (cu32)(uintptr_t)(a.uq + ((size_t)tile * a.umax + u) * 4);
#pragma unroll
      for (int w = 0; w < 4; ++w) qm[w] = m[w];
    }
#pragma unroll
    for (int w = 0; w < NLK; ++w) {
      const int f = tid + NT * w, q = f >> 4, kc = 4 * (f & 15);
      kt[w] = v4f{0.f, 0.f, 0.f, 0.f};
      if (q < TQ && ((qm[q >> 5] >> (q & 31)) & 1u)) kt[w] = *reinterpret_cast<const v4f *>(kblk + (size_t)q * 64 + kc);
    }
  };
  auto commit = [&](const int u) {
    const int k0 = uoff[u], jc = uoff[u + 1] - k0;
    if (vec4) {
#pragma unroll
      for (int w = 0; w < NLV; ++w) {
        const int f = tid + NT * w, row = f / d4, col4 = f - row * d4;
        if (row < JC) *reinterpret_cast<v4f *>(&Vs[row * YS + 4 * col4]) = vq[w];
      }
    } else {  // D % 4 != 0: element by element, straight from memory
      for (int e = tid; e < JC * D; e += NT) {
        const int row = e / D, col = e - row * D;
        Vs[row * YS + col] = row < jc ? a.V[(size_t)kidx[k0 + row] * D + col] : 0.f;
      }
    }
#pragma unroll
    for (int w = 0; w < NLK; ++w) {
      const int f = tid + NT * w, q = f >> 4, kc = 4 * (f & 15);
      if (q < TQ) {
        v4f t = kt[w];  // (lanes past the unit's last key were stored as zeros by pass 1)
        *reinterpret_cast<v4f *>(&Kt[q * KS2 + kc]) = t;
      }
    }
    return jc;
  };
  int u = next_unit(u0);
  if (u < u1) issue(u);
  while (u < u1) {
    wg_sync();  // the previous unit's products are done with Vs / Kt
    const int jc = commit(u);
    wg_sync();
    u = next_unit(u + 1);
    if (u < u1) issue(u);  // in flight during the products
    (void)jc;  // (slots past the unit's last key hold zeros in both operands: whole 16 steps, a constant trip count)
#pragma unroll 4
    for (int k4 = 0; k4 < JC / 4; ++k4) {
      const float b0 = Kt[(rt0 * 16 + r16) * KS2 + 4 * k4 + g];
      float as[NCT];
#pragma unroll
      for (int t = 0; t < NCT; ++t) as[t] = Vs[(4 * k4 + g) * YS + 16 * t + r16];
#pragma unroll
      for (int t = 0; t < NCT; ++t) acc0[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(as[t], b0, acc0[t], 0, 0, 0);
    }
  }
  // rows from the accumulators: query = l % 16 of the row tile, columns 16 t + 4 (l / 16) ..
  {
    const int pos = tile * TQ + rt0 * 16 + r16;
    if (pos < a.n_local) {
      const int il = a.qperm ? a.qperm[pos] : pos;
      const size_t row = ((size_t)js * a.n_local + il) * a.ldp;
#pragma unroll
      for (int t = 0; t < NCT; ++t)
        if (16 * t + 4 * g < a.ldp) *reinterpret_cast<v4f *>(a.pA + row + 16 * t + 4 * g) = acc0[t];
    }
  }
}

// ---- forward's log p over run lists (round 6): pairwise_logp_mfma.hpp's product-form log-sum-exp, walking the near keys of a 64-query
// group instead of every key.  The dense pass is 372 us at cfg4 (0.79 of the fp32 MFMA peak) for sums whose terms are all but 0.1 %
// below 2^-43 of their leader; its pre-pass (pairwise_far.hpp on the UPDATED particles, 64-query groups, with key masks) + the lists
// cost less than half of that once the set is large.  Workgroup = one group: 4 waves x 16 queries (one MFMA column tile per wave, the
// query rows in registers as B operands), the group's listed keys gathered through LDS in units of 64 (double buffered); 4 MFMAs per
// 16-byte LDS read.  The query's own term is put in exactly (pairwise_logp_mfma.hpp), found by key index.
struct LogpPackedArgs {
  LogpMfmaArgs a;      // Z / hq / hj rows, pM / pL, JS = slices per group
  const int *kidx;     // [groups][ldi]
  int ldi, umax;
  const int *uoff;     // [groups][umax + 1]
  const unsigned int *uq;  // [groups][umax][4] (words 0-1: the group's 64 queries)
  const int *soff;     // [groups][JS + 1]
  const int *qperm;    // tile order (position -> local row), or nullptr
};

template <int DPB>
static inline size_t pairwise_logp_packed_lds_bytes() {
  return sizeof(float) * (2 * (size_t)64 * (DPB + 4) + 4 * 64);
}

template <int DPB>
__global__ __launch_bounds__(256, 2) void pairwise_logp_packed_kernel(const LogpPackedArgs b) {
  constexpr int JC = 64, NT = 256, YS = DPB + 4, NP = DPB / 16, R4 = DPB / 4;
  static_assert(DPB % 16 == 0, "whole 16-column pieces");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const LogpMfmaArgs &a = b.a;
  float *Ys = lds;                // [2][JC][YS] key rows
  float *hs = lds + 2 * JC * YS;  // [2][JC] key constants (slots behind the unit's last key: -inf)
  int *ks = reinterpret_cast<int *>(hs + 2 * JC);  // [2][JC] key indices
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, r16 = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = blockIdx.x, js = blockIdx.y;
  typedef const int __attribute__((address_space(4))) * ci32;
  typedef const unsigned int __attribute__((address_space(4))) * cu32;
  const ci32 soff = (ci32)(uintptr_t)(b.soff + (size_t)grp * (a.JS + 1));
  const ci32 uoff = (ci32)(uintptr_t)(b.uoff + (size_t)grp * (b.umax + 1));
  const int *kidx = b.kidx + (size_t)grp * b.ldi;
  const int u0 = soff[js], u1 = soff[js + 1];
  const int pos = grp * 64 + 16 * wave + r16;  // the lane's query: position in tile order
  const int posc = min(pos, a.n_local - 1);
  const int row = b.qperm ? b.qperm[posc] : posc, gq = a.i0 + row;
  v4f bq[NP];
#pragma unroll
  for (int s = 0; s < NP; ++s) bq[s] = *reinterpret_cast<const v4f *>(a.Z + (size_t)gq * DPB + 16 * s + 4 * g);
  float m = -3.0e38f, sm = 0.f;
  const float hq2 = 2.0f * a.hq[gq];
  v4f ky[NP];
  float hjn = 0.f;
  int kin = 0;
  auto keys_issue = [&](const int u) {
    const int k0 = uoff[u], jc = uoff[u + 1] - k0;
    int key[NP];
#pragma unroll
    for (int w = 0; w < NP; ++w) key[w] = kidx[k0 + min((tid + NT * w) / R4, jc - 1)];
    const int kl = kidx[k0 + min(tid & 63, jc - 1)];
#pragma unroll
    for (int w = 0; w < NP; ++w) {
      const int f = tid + NT * w, rr = f / R4, c4 = f - rr * R4;
      ky[w] = *reinterpret_cast<const v4f *>(a.Z + (size_t)key[w] * DPB + 4 * c4);
    }
    if (tid < JC) {
      hjn = tid < jc ? a.hj[kl] : -INFINITY;
      kin = tid < jc ? kl : -1;
    }
  };
  auto keys_commit = [&](const int buf) {
#pragma unroll
    for (int w = 0; w < NP; ++w) {
      const int f = tid + NT * w, rr = f / R4, c4 = f - rr * R4;
      *reinterpret_cast<v4f *>(&Ys[(buf * JC + rr) * YS + 4 * c4]) = ky[w];
    }
    if (tid < JC) {
      hs[buf * JC + tid] = hjn;
      ks[buf * JC + tid] = kin;
    }
  };
  if (u0 < u1) {
    keys_issue(u0);
    keys_commit(0);
  }
  wg_sync();
  int buf = 0;
  for (int u = u0; u < u1; ++u, buf ^= 1) {
    // does one of the wave's 16 queries have a near key in this unit?
    bool own = true;
    if (b.uq) {
      const cu32 qm = (cu32)(uintptr_t)(b.uq + ((size_t)grp * b.umax + u) * 4);
      const unsigned int w32 = qm[wave >> 1];
      own = ((w32 >> (16 * (wave & 1))) & 0xffffu) != 0u;
    }
    const bool more = u + 1 < u1;
    if (more) keys_issue(u + 1);  // in flight during the products
    const float *Yb = Ys + (size_t)buf * JC * YS;
#pragma unroll 1
    for (int kt = 0; kt < (own ? JC / 16 : 0); ++kt) {
      v4f acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < NP; ++s) {
        const v4f av = *reinterpret_cast<const v4f *>(&Yb[(16 * kt + r16) * YS + 16 * s + 4 * g]);
#pragma unroll
        for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c], bq[s][c], acc, 0, 0, 0);
      }
      // the lane holds (keys 16 kt + 4 g + r, query r16): online log-sum-exp in base 2
      const v4f h = *reinterpret_cast<const v4f *>(&hs[buf * JC + 16 * kt + 4 * g]);
      const int4 kid = *reinterpret_cast<const int4 *>(&ks[buf * JC + 16 * kt + 4 * g]);
      v4f x = acc + h;
      x.x = kid.x == gq ? h.x - hq2 : x.x;  // the query's OWN term, exact (distance 0: logit = log w_i)
      x.y = kid.y == gq ? h.y - hq2 : x.y;
      x.z = kid.z == gq ? h.z - hq2 : x.z;
      x.w = kid.w == gq ? h.w - hq2 : x.w;
      const float mx = fmaxf(fmaxf(x.x, x.y), fmaxf(x.z, x.w));
      const float mn = fmaxf(m, mx);
      const float e = (__builtin_amdgcn_exp2f(x.x - mn) + __builtin_amdgcn_exp2f(x.y - mn)) +
                      (__builtin_amdgcn_exp2f(x.z - mn) + __builtin_amdgcn_exp2f(x.w - mn));
      sm = fmaf(sm, __builtin_amdgcn_exp2f(m - mn), e);
      m = mn;
    }
    if (more) keys_commit(buf ^ 1);
    wg_sync();
  }
  // merge the 4 lane groups of a query (lanes r16, r16 + 16, + 32, + 48), then one lane per query writes the slice partial
#pragma unroll
  for (int o = 16; o < 64; o <<= 1) {
    const float mo = __shfl_xor(m, o), so = __shfl_xor(sm, o);
    const float mn = fmaxf(m, mo);
    sm = sm * __builtin_amdgcn_exp2f(m - mn) + so * __builtin_amdgcn_exp2f(mo - mn);
    m = mn;
  }
  if (g == 0 && pos < a.n_local) {
    a.pM[(size_t)js * a.n_local + row] = sm > 0.f ? (m + a.hq[gq]) * 0.69314718055994531f : -INFINITY;
    a.pL[(size_t)js * a.n_local + row] = sm;
  }
}

// ---- the tile order of the NEXT tick: the rank's rows sorted by (leader, row) - a stable two-pass radix sort (7 + 7 bits, keys below
// 16 384: validate() caps N) of at most 16 384 rows in ONE workgroup; reads the leaders pass 1 noted, re-arms them, writes qperm ------
static inline size_t query_order_lds_bytes(int n) { return (size_t)3 * ((n + 1023) / 1024) * 1024 * sizeof(unsigned short) + (16 * 128 + 128 + 16) * sizeof(int); }

__global__ __launch_bounds__(1024) void query_order_kernel(int *lead, int *qperm, const int n_local, const int i0, const int N) {
  extern __shared__ __attribute__((aligned(16))) unsigned short qo_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int npad = ((n_local + 1023) / 1024) * 1024, seg = npad / 16;  // rows per wave: a multiple of 64
  unsigned short *key = qo_lds;       // [npad] by row
  unsigned short *ia = key + npad;    // [npad] rows in the current order
  unsigned short *ib = ia + npad;     // [npad] ... and the next one
  int *cnt = reinterpret_cast<int *>(ib + npad);  // [16][128] per wave and digit: count, then running offset
  int *tot = cnt + 16 * 128;                      // [128] per digit
  for (int r = tid; r < npad; r += 1024) {
    int k = 0x3fff;  // (padding rows sort behind everything)
    if (r < n_local) {
      const int l = lead[r];
      k = (l >= 0 && l < N) ? l : min(i0 + r, N - 1);  // (no close key seen - a row pass 1 masked out everywhere: its own index)
      lead[r] = 0x7fffffff;                              // re-armed for the next pass 1
    }
    key[r] = (unsigned short)min(k, 0x3fff);
    ia[r] = (unsigned short)r;
  }
  wg_sync();
  const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;  // lanes below this one
  for (int pass = 0; pass < 2; ++pass) {
    const int shift = 7 * pass;
    const unsigned short *src = pass ? ib : ia;
    unsigned short *dst = pass ? ia : ib;
    for (int e = tid; e < 16 * 128; e += 1024) cnt[e] = 0;
    wg_sync();
    auto peers_of = [&](const int d) {  // lanes of the wave holding the same digit
      unsigned long long m = ~0ull;
#pragma unroll
      for (int bit = 0; bit < 7; ++bit) {
        const unsigned long long bb = __ballot((d >> bit) & 1);
        m &= ((d >> bit) & 1) ? bb : ~bb;
      }
      return m;
    };
    for (int c = 0; c < seg; c += 64) {  // the wave's segment, 64 rows at a time, in order
      const int d = (key[src[wave * seg + c + lane]] >> shift) & 127;
      const unsigned long long m = peers_of(d);
      if ((m & lt) == 0ull) cnt[wave * 128 + d] += (int)__builtin_popcountll(m);  // (one lane per digit: no atomic)
    }
    wg_sync();
    if (tid < 128) {  // per digit: the waves' counts -> offsets inside the digit, and the digit's total
      int run = 0;
      for (int w = 0; w < 16; ++w) {
        const int v = cnt[w * 128 + tid];
        cnt[w * 128 + tid] = run;
        run += v;
      }
      tot[tid] = run;
    }
    wg_sync();
    if (wave == 0) {  // exclusive scan of the 128 totals
      int a0 = tot[2 * lane], a1 = tot[2 * lane + 1];
      int s2 = a0 + a1, ps = s2;
      for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(ps, o);
        if (lane >= o) ps += t;
      }
      const int ex = ps - s2;
      tot[2 * lane] = ex;
      tot[2 * lane + 1] = ex + a0;
    }
    wg_sync();
    for (int c = 0; c < seg; c += 64) {
      const unsigned short row = src[wave * seg + c + lane];
      const int d = (key[row] >> shift) & 127;
      const unsigned long long m = peers_of(d);
      const int base = tot[d] + cnt[wave * 128 + d];  // (read by all peers before the first of them moves it on)
      dst[base + (int)__builtin_popcountll(m & lt)] = row;
      if ((m & lt) == 0ull) cnt[wave * 128 + d] += (int)__builtin_popcountll(m);
    }
    wg_sync();
  }
  for (int p = tid; p < n_local; p += 1024) qperm[p] = (int)ia[p];  // (padding rows carry the largest key and the largest rows: they sit behind)
}

}  // namespace dust
