// pairwise_far.hpp - which (query tile, key chunk) units of pairwise_packed_kernel contribute EXACTLY nothing, decided on the
// matrix cores before the exact-difference pass runs (svmpc.py:38-41 prior gradient, 76-83 Stein kernel; K1 only).
//
// In H d_a = 80 dimensions a spread-out particle set has no near neighbours: the K1 value exp(-d2_S / 2) and the prior's softmax
// term exp(log w_j - d2_P / 2 - max) underflow to exact fp32 zeros for all but a handful of pairs, and a unit whose 96 x 64 pairs
// are all such zeros adds nothing to any output of the fused pass - yet the exact-difference distance of every pair is what that
// pass spends its time on (2.5 of 3.8 ms at cfg4).  This pre-pass bounds the distances from below with a binary16 GEMM:
//     G_ij = sum_k g_k (x_ik - x_jk)^2 ,   g_k = min(1 / ell^2, 1 / sigma_p,k^2)   (<= both metrics' distances)
//          = n_i + n_j - 2 z_i . z_j ,     z = sqrt(g) (x - x_0) rounded to binary16 for the product, n = |z|^2 from the fp32 rows
// with the product's error bounded by 2^-10 |z_i| |z_j| <= 2^-11 (n_i + n_j) (two roundings of relative size 2^-11; fp32
// accumulation adds < 1e-5 of that).  A unit is FAR when every pair has
//     G_ij - EB (n_i + n_j) > T + 2 max(0, log w_j - min_tile m0_i) ,        EB = 1.3e-3 ,
// where the fused pass STARTS each query's running max at m0_i (below).  Two thresholds:
//   T = 224 (DUST_FAR_T=224): 0.7213 d2_S > 161 (the K1 exponent in base 2: v_exp_f32 returns 0 below -150 with or without
//     denormals) and every softmax exponent of the unit is below -161 as well: the terms are exact zeros, the running max and sum
//     do not move, the rescale factor is exactly 1 - skipping the unit is BIT-IDENTICAL to evaluating it (the slack between 150
//     and 161 covers the fused pass' own fp32 rounding of d2, < 1e-5 relative);
//   T = 60 (the default): every skipped kernel value and softmax term is below e^-30 = 2^-43 of the leading term of the sum it
//     belongs to (k_ii = 1; the term of the key that attains m0_i - a real key, visited) - at most N = 2^14 .. 2^17 of them, so
//     each sum changes by less than 2^-26 of its leading term: under half an ulp of it.  In 80 dimensions at sigma_p = 1 the
//     pairs of a spread-out set sit at d2 ~ 320 +- 50: beyond 60, mostly short of 224 (cfg4, bench.py).
// Rows the bound cannot speak for (non-finite coordinates or weights, |z| beyond binary16) get n = -inf and are never far.
// m0_i is a lower bound of query i's final max logit, known before the pass: the largest of its OWN logit log w_i (the keys are
// the particles: distance 0) and its exact logits against one candidate key per 256 keys - the group's heaviest particle
// (far_lb_kernel: N x N / 256 exact distances).  After the first tick the mixture weights are
// softmax(-alpha cost) - close to one-hot - and a light query's max logit is its logit against a heavy FAR particle, hundreds above
// its own: without the candidates no unit of such a query could be proven negligible.  Any reference <= the true max + O(ulp)
// serves the softmax equally well, so every mode of the fused pass starts from m0 (tests: DUST_FAR_T=224 against DUST_FAR=0 and
// DUST_DENSE=1 bitwise; the default against them at 1e-6 of the largest element).
// Cost: 2 D flops per pair on v_mfma_f32_16x16x32_f16 instead of 3 D packed fp32 lane-ops - and for a clustered set it buys nothing
// (every unit stays, the pre-pass is ~3 % on top).  Time is data dependent, results are not.
#pragma once
#include "pairwise_fused.hpp"
#include "pairwise_logp_mfma.hpp"

namespace dust {

typedef _Float16 v4h __attribute__((ext_vector_type(4)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

#define DUST_FAR_T_EXACT 224.0f  // every skipped term is an exact fp32 zero
#define DUST_FAR_T_DEFAULT 60.0f  // every skipped term is below e^-30 = 2^-43 of its sum's leading term
#define DUST_FAR_EB 1.3e-3f

struct FarArgs {
  int N, D, i0, n_local, tiles, chunks, cps;  // tiles of TQ query rows (far_flags_kernel's template argument); cps: key chunks per slice (blockIdx.y)
  int q_rows;           // tiles * TQ
  float lscale;         // log weights enter as lscale * logmix (1: natural-log logits; log2 e: the base-2 logits of pairwise_logp_mfma.hpp)
  const float *X;       // [N][D] particles
  const float *logmix;  // [N] log mixture weights
  float sg[2];          // sqrt(g) for even / odd dimensions
  _Float16 *Z;          // [N][far_zh(DPB)] scaled, centred, zero-padded rows in binary16 (far_prep_kernel)
  float *nrm;           // [N] |z|^2 (1 - EB), or -inf: the row is never far
  float *lms;           // [N] log w (finite or -inf)
  const float *Xp;      // [N][DPB] zero-padded fp32 rows (the fused pass' own copy)
  float wP[2];          // the prior metric, 1 / sigma_p^2 for even / odd dimensions
  float *m0;            // [N] lower bound of the query's max prior logit (far_lb_kernel): the fused pass starts its running max there
  unsigned char *far;   // [tiles][chunks] 1 = the unit contributes exactly nothing
  unsigned char *nz;    // the Gram-block flags of pairwise_fused.hpp: zeroed here for far units (the fused pass never visits them)
  int ldnz;
  float T;              // the threshold on G (DUST_FAR_T_DEFAULT; development switch DUST_FAR_T)
  unsigned int *qmask;  // [tiles][chunks][8] per unit (MASKS instances): words 0-3: bit q = query q of the tile has a NEAR key in the chunk (the
                        // fused pass computes exact distances for those queries only); words 4-5: bit k = key k of the chunk has a near
                        // query in the tile (the others get no term at all: their weights are forced to exact zeros, pass B skips them)
  const int *qperm;     // [n_local] tile order of the QUERIES (position -> local row; pairwise_packed.hpp), or nullptr: identity
  unsigned int *count;  // [3] (8-byte aligned) {far units, all units} of this launch (zeroed by the row kernel) and the number of
                        // counted launches so far (never zeroed: it numbers the reports), or nullptr
  unsigned int *host_count;  // [3] pinned host words the LAST workgroup of far_flags_kernel copies {far, all, launch number} to (no
                             // copy node, no stream synchronisation: the host knows how many counted launches it has issued and
                             // reads the report of exactly the last one - dust_amd.hip logp_far_decide)
};

// binary16 row length: whole K = 32 steps of v_mfma_f32_16x16x32_f16 (gfx950's full-rate shape: the 16x16x16 one runs at a quarter of it)
static constexpr __host__ __device__ int far_zh(int dpb) { return ((dpb + 31) / 32) * 32; }

// one wave per row (far_prep_kernel; the log-p pass runs it inside its own row kernel, logp_prep_far_kernel below)
template <int DPB>
__device__ __forceinline__ void far_prep_row(const FarArgs &a, const int row, const int lane) {
  float acc = 0.f;
  bool ok = true;
  float v[(DPB + 63) / 64];
#pragma unroll
  for (int u = 0; u < (DPB + 63) / 64; ++u) {
    const int c = lane + 64 * u;
    v[u] = 0.f;
    if (c < a.D) v[u] = (a.X[(size_t)row * a.D + c] - a.X[c]) * a.sg[c & 1];
    ok = ok && fabsf(v[u]) <= 60000.0f;  // (false for NaN)
    acc = fmaf(v[u], v[u], acc);
  }
  acc = wave_sum(acc);
  const float lm = a.logmix[row];
  ok = __ballot(!ok) == 0ull && (lm < INFINITY) && acc < INFINITY;  // (lm NaN: false)
  constexpr int ZH = far_zh(DPB);
#pragma unroll
  for (int u = 0; u < (ZH + 63) / 64; ++u) {
    const int c = lane + 64 * u;
    const float val = u < (DPB + 63) / 64 ? v[u < (DPB + 63) / 64 ? u : 0] : 0.f;
    if (c < ZH) a.Z[(size_t)row * ZH + c] = ok && c < a.D ? (_Float16)val : (_Float16)0.f;
  }
  if (lane == 0) {
    a.nrm[row] = ok ? acc * (1.0f - DUST_FAR_EB) : -INFINITY;
    a.lms[row] = ok ? lm * a.lscale : 0.f;
    if (row == 0 && a.count) a.count[0] = a.count[1] = 0u;  // (far_flags_kernel, the next launch of the stream, counts into them)
  }
}
template <int DPB>
__global__ __launch_bounds__(256) void far_prep_kernel(const FarArgs a) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row < a.N) far_prep_row<DPB>(a, row, lane);
}

// m0 (see the file comment): lane = query (row in registers), the 16 waves of the workgroup share the candidates (wave w takes
// w, w + 16, ... through the scalar path: the serial loop per wave is what bounds this launch - 57 us with 4 waves, whatever the
// set size); the logit is formed as the fused pass forms it (differences first, even / odd dimensions apart, log w - pa / 2)
#define DUST_FAR_LB_WAVES 16
template <int DPB>
__global__ __launch_bounds__(64 * DUST_FAR_LB_WAVES) void far_lb_kernel(const FarArgs a) {
  constexpr int NW = DUST_FAR_LB_WAVES;
  __shared__ float red[NW][64];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q_end = min(a.N, a.i0 + a.q_rows);
  const int qi = a.i0 + blockIdx.x * 64 + lane, qc = min(qi, a.N - 1);
  v2f x[DPB / 2];
#pragma unroll
  for (int p = 0; p < DPB / 4; ++p) {
    const v4f t = *reinterpret_cast<const v4f *>(a.Xp + (size_t)qc * DPB + 4 * p);
    x[2 * p] = v2f{t.x, t.y};
    x[2 * p + 1] = v2f{t.z, t.w};
  }
  float best = -INFINITY;
  const int groups = (a.chunks + 3) / 4;
  for (int ci = wave; ci < groups; ci += NW) {
    // the candidate of key group ci (4 chunks = 256 keys): its heaviest particle (first of equals; NaN weights count as -inf) - the
    // heaviest particle of the whole set is always one of them, and with softmax(-alpha cost) weights it is the one that matters.
    // Every workgroup finds the candidates again - N log-weights out of L2 per workgroup, 1 % of its row traffic - instead of a
    // launch of its own.  (One candidate per CHUNK - 16 per wave at N = 16384 - made this launch 29 us of serial latency.)
    int c;
    {
      float mx = -INFINITY;
      int arg = ci * 256;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = ci * 256 + 64 * u + lane;
        float v = j < a.N ? a.logmix[j] : -INFINITY;
        v = v == v ? v : -INFINITY;
        const float m1 = wave_max(v);
        const unsigned long long hit = __ballot(v == m1);
        if (m1 > mx) {  // wave-uniform
          mx = m1;
          arg = ci * 256 + 64 * u + (hit ? (int)__builtin_ctzll(hit) : 0);
        }
      }
      c = min(arg, a.N - 1);
    }
    typedef const v2f __attribute__((address_space(4))) * cv2;
    const cv2 y = (cv2)(uintptr_t)(a.Xp + (size_t)c * DPB);
    v2f d2 = {0.f, 0.f};
#pragma unroll
    for (int p = 0; p < DPB / 2; ++p) {
      const v2f z = x[p] - y[p];
      d2 = __builtin_elementwise_fma(z, z, d2);
    }
    const float pa = d2.x * a.wP[0] + d2.y * a.wP[1];
    const float lg = a.logmix[c] * a.lscale - 0.5f * pa;
    best = lg > best ? lg : best;  // (NaN: not taken)
  }
  red[wave][lane] = best;
  wg_sync();
  if (wave == 0 && qi < q_end) {
    float m = a.logmix[qi] * a.lscale;  // the query's own logit
    m = m == m ? m : -INFINITY;
#pragma unroll
    for (int w = 0; w < NW; ++w) m = fmaxf(m, red[w][lane]);
    a.m0[qi] = m;
  }
}

// the log-p pass' row kernel and the far pre-pass' in one launch (one wave per row)
template <int DPB>
__global__ __launch_bounds__(256) void logp_prep_far_kernel(const LogpMfmaArgs b, const FarArgs a) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= b.N) return;
  float acc = 0.f;
#pragma unroll
  for (int c = lane; c < DPB; c += 64) {
    float v = 0.f;
    if (c < b.D) v = (b.X[(size_t)row * b.D + c] - b.X[c]) * b.sw[c & 1];
    b.Z[(size_t)row * DPB + c] = v;
    acc = fmaf(v, v, acc);
  }
  acc = wave_sum(acc);
  if (lane == 0) {
    b.hq[row] = -0.5f * acc;
    b.hj[row] = b.logmix[row] * 1.44269504088896340736f - 0.5f * acc;
  }
  far_prep_row<DPB>(a, row, lane);
}

template <int DPB>
static inline size_t far_flags_lds_bytes() {
  return 2 * ((size_t)64 * (far_zh(DPB) + 8) * sizeof(_Float16) + 2 * 64 * sizeof(float));
}

// Workgroup = 4 waves = 4 query tiles (TQ rows each - the consumer's tile: pairwise_packed_kernel's, or the 64 queries of a wave of
// pairwise_logp_mfma_kernel - held in registers as B operands), the key
// chunks of its slice streamed through LDS (double buffered, one barrier per chunk); a wave runs TQ / 16 MFMAs per 16-byte LDS read.
template <int DPB, int TQ, bool MASKS>
__global__ __launch_bounds__(256, 3) void far_flags_kernel(const FarArgs a) {
  constexpr int JC = 64, NT = 256, QT = TQ / 16, ZH = far_zh(DPB), NP = ZH / 32, ZS = ZH + 8, R8 = ZH / 8;
  constexpr int NLD = (JC * R8 + NT - 1) / NT;
  static_assert(TQ % 16 == 0, "whole MFMA tiles");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr size_t BUF = (size_t)JC * ZS * sizeof(_Float16) + 2 * JC * sizeof(float);
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, r16 = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = blockIdx.x * 4 + wave, N = a.N;
  const bool live = tile < a.tiles;  // (a wave behind the last tile still stages keys)
  const int tq = live ? tile : a.tiles - 1;
  const int c_beg = blockIdx.y * a.cps, c_end = min(a.chunks, c_beg + a.cps);
  unsigned int n_far = 0u, n_all = 0u;  // wave-uniform: this wave's units
  // One 64-bit atomic per WORKGROUP (far units in the low word, all units in the high one - per-wave 32-bit atomics on two words
  // cost 55 us of a 60 us launch, per-unit ones 1.4 ms); the workgroup that completes the total hands the counts to the host.
  auto finish = [&]() {
    if (!a.count || !a.host_count) return;
    __shared__ unsigned int wcnt[4][2];
    if (lane == 0) {
      wcnt[wave][0] = n_far;
      wcnt[wave][1] = n_all;
    }
    wg_sync();
    if (tid == 0) {
      const unsigned long long mine = (unsigned long long)(wcnt[0][0] + wcnt[1][0] + wcnt[2][0] + wcnt[3][0]) |
                                      ((unsigned long long)(wcnt[0][1] + wcnt[1][1] + wcnt[2][1] + wcnt[3][1]) << 32);
      if (mine) {
        const unsigned long long tot = atomicAdd(reinterpret_cast<unsigned long long *>(a.count), mine) + mine;
        if ((unsigned int)(tot >> 32) == (unsigned int)a.tiles * (unsigned int)a.chunks) {
          const unsigned int seq = a.count[2] + 1u;  // (one workgroup per launch gets here, launches of a stream are ordered)
          a.count[2] = seq;
          __hip_atomic_store(a.host_count, (unsigned int)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(a.host_count + 1, (unsigned int)(tot >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(a.host_count + 2, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
  };
  if (c_beg >= c_end) {
    finish();
    return;
  }

  v8h bq[QT][NP];
  float hqh[QT];  // -n_q (1 - EB) / 2: where the query's accumulators start (+inf: the row is never far)
  float lmq = INFINITY;
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    // (rows behind the rank's last: clamped, as pass 1 clamps them; pass 1's tile order when it has one)
    const int qi = a.qperm ? a.i0 + a.qperm[min(tq * TQ + 16 * t + r16, a.n_local - 1)] : min(a.i0 + tq * TQ + 16 * t + r16, N - 1);
#pragma unroll
    for (int s = 0; s < NP; ++s) bq[t][s] = *reinterpret_cast<const v8h *>(a.Z + (size_t)qi * ZH + 32 * s + 8 * g);
    hqh[t] = -0.5f * a.nrm[qi];
    lmq = fminf(lmq, a.m0[qi]);
  }
  lmq = wave_min(lmq);

  v8h ky[NLD];
  float kn_next = 0.f, kl_next = 0.f;
  auto keys_issue = [&](const int ch) {
    const int j0 = ch * JC;
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int f = tid + NT * u, row = f / R8, c8 = f - row * R8;
      ky[u] = *reinterpret_cast<const v8h *>(a.Z + (size_t)min(j0 + row, N - 1) * ZH + 8 * c8);
    }
    if (tid < JC) {
      const bool kval = j0 + tid < N;  // (keys behind the set do not exist for the fused pass: far by definition)
      kn_next = kval ? a.nrm[j0 + tid] - a.T : 3.0e38f;
      kl_next = kval ? a.lms[j0 + tid] : -INFINITY;
    }
  };
  auto keys_commit = [&](const int buf) {
    unsigned char *base = lds_raw + buf * BUF;
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int f = tid + NT * u, row = f / R8, c8 = f - row * R8;
      if (row < JC) *reinterpret_cast<v8h *>(base + ((size_t)row * ZS + 8 * c8) * sizeof(_Float16)) = ky[u];
    }
    float *kf = reinterpret_cast<float *>(base + (size_t)JC * ZS * sizeof(_Float16));
    if (tid < JC) {
      kf[tid] = kn_next;
      kf[JC + tid] = kl_next;
    }
  };
  keys_issue(c_beg);
  keys_commit(0);
  wg_sync();
  int buf = 0;
  for (int ch = c_beg; ch < c_end; ++ch, buf ^= 1) {
    const bool more = ch + 1 < c_end;
    if (more) keys_issue(ch + 1);  // in flight during the products
    const unsigned char *base = lds_raw + buf * BUF;
    const _Float16 *Zs = reinterpret_cast<const _Float16 *>(base);
    const float *kf = reinterpret_cast<const float *>(base + (size_t)JC * ZS * sizeof(_Float16));
    // Round 6: the margins are never formed.  margin = key constant - 2 acc > 0  <=>  acc < key constant / 2: ONE compare per accumulator
    // component, its lane mask straight into scalar registers - and the per-query / per-key minima of round 5 (2 packed FMAs, 6-14
    // three-operand minima and 16 DPP steps per 16 x 16 tile, on the vector pipe the MFMAs share) become ORs on the scalar pipe.
    // A NaN accumulator compares "not below": near, as before.
    unsigned long long qacc[QT];  // lane (g, r16): query r16 of sub-tile t has a near key among the chunk's keys 16 kt + 4 g + r
#pragma unroll
    for (int t = 0; t < QT; ++t) qacc[t] = 0ull;
    unsigned long long kmask = 0ull;  // (MASKS) keys of the chunk with a near query in the tile; wave-uniform
#pragma unroll 1
    for (int kt = 0; kt < JC / 16; ++kt) {
      // (the accumulators start at -n_q / 2: acc = z_q . z_k - n_q / 2, and the pair is far when acc < (n_k - T - 2 dl) / 2)
      v4f acc[QT];
#pragma unroll
      for (int t = 0; t < QT; ++t) acc[t] = v4f{hqh[t], hqh[t], hqh[t], hqh[t]};
#pragma unroll
      for (int s = 0; s < NP; ++s) {
        const v8h av = *reinterpret_cast<const v8h *>(&Zs[(16 * kt + r16) * ZS + 32 * s + 8 * g]);
#pragma unroll
        for (int t = 0; t < QT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bq[t][s], acc[t], 0, 0, 0);
      }
      // the lane holds (keys 16 kt + 4 g + r, query r16 of sub-tile t)
      const v4f kn = *reinterpret_cast<const v4f *>(&kf[16 * kt + 4 * g]);
      const v4f kl = *reinterpret_cast<const v4f *>(&kf[JC + 16 * kt + 4 * g]);
      float thr[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float dl = kl[r] == -INFINITY ? 0.f : fmaxf(kl[r] - lmq, 0.f);  // (a key of zero weight has no prior term at all)
        thr[r] = 0.5f * fmaf(-2.0f, dl, kn[r]);
      }
      unsigned long long kr[4] = {0ull, 0ull, 0ull, 0ull};  // component r: lanes whose key 16 kt + 4 g + r is near SOME sub-tile's query r16
#pragma unroll
      for (int t = 0; t < QT; ++t) {
        const unsigned long long b0 = __ballot(!(acc[t][0] < thr[0])), b1 = __ballot(!(acc[t][1] < thr[1]));
        const unsigned long long b2 = __ballot(!(acc[t][2] < thr[2])), b3 = __ballot(!(acc[t][3] < thr[3]));
        qacc[t] |= (b0 | b1) | (b2 | b3);
        if (MASKS) {
          kr[0] |= b0;
          kr[1] |= b1;
          kr[2] |= b2;
          kr[3] |= b3;
        }
      }
      if (MASKS) {
        unsigned long long k16 = 0ull;  // bit 4 g + r: some lane of row g (its 16 queries) flagged key 16 kt + 4 g + r
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int gg = 0; gg < 4; ++gg) k16 |= ((kr[r] >> (16 * gg)) & 0xffffull) ? (1ull << (4 * gg + r)) : 0ull;
        kmask |= k16 << (16 * kt);
      }
    }
    bool any_near = false;
#pragma unroll
    for (int t = 0; t < QT; ++t) any_near = any_near || qacc[t] != 0ull;
    const bool is_far = !any_near;  // wave-uniform
    if (MASKS) {
      // per query: near when one of its four lanes (r16, r16 + 16, + 32, + 48) saw a near key
      unsigned int qm[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int t = 0; t < QT; ++t) {
        const unsigned long long q = qacc[t];
        const unsigned int b16 = (unsigned int)((q | (q >> 16) | (q >> 32) | (q >> 48)) & 0xffffull);  // queries 16 t + r16
        qm[t >> 1] |= b16 << (16 * (t & 1));
      }
      if (live && lane == 0) {
        unsigned int *dst = a.qmask + ((size_t)tile * a.chunks + ch) * 8;
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[k] = qm[k];
        dst[4] = (unsigned int)kmask;
        dst[5] = (unsigned int)(kmask >> 32);
      }
    }
    if (live) {
      n_far += is_far ? 1u : 0u;  // (one atomic per wave at the end: 65 536 units adding to one word took 1.4 ms)
      n_all += 1u;
      if (lane == 0) a.far[(size_t)tile * a.chunks + ch] = is_far ? 1 : 0;
      if (is_far && a.nz)
        for (int i = lane; i < TQ; i += 64) a.nz[(size_t)ch * a.ldnz + tile * TQ + i] = 0;
    }
    if (more) keys_commit(buf ^ 1);
    wg_sync();
  }
  finish();
}

}  // namespace dust
