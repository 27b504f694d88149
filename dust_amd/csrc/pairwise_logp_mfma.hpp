// pairwise_logp_mfma.hpp - log p(theta_i) under the particle-centred prior GMM for LARGE aliased sets (SVMPC.forward ->
// get_weights, svmpc.py:128-140: `self.prior.log_prob(theta)`, prior = get_gmm(theta, weights, sigma_p^2 I) svgd.py:84-89) on the
// matrix cores.  Only the log-sum-exp of the logits is needed here - no gradient - so the squared distance may be taken in its
// product form:
//     logit_ij = log w_j - |z_i - z_j|^2 / 2 = (log w_j - |z_j|^2 / 2) + z_i . z_j - |z_i|^2 / 2 ,   z = (theta - theta_0) / sigma_p
// z_i . z_j is a GEMM (v_mfma_f32_16x16x4_f32, fp32 in / fp32 accumulate): 2 D flops per pair on the MFMA pipe instead of the 4 D
// packed lane-flops of the exact-difference pass (pairwise_logp_big_kernel: 904 us at cfg4, N = 16384, D = 80).  The rows are
// centred on particle 0 so that |z|^2 is the spread of the cloud, not its offset: the cancellation error of the product form is
// ~ 4 eps |z|^2 on a logit (2e-5 at |z|^2 = 80), far below the 1e-5 RELATIVE tolerance on log p (|log p| >= log N).  The passes that
// need gradients (pairwise_fused.hpp) keep exact differences.
//
// Everything is kept in base-2 units: z is pre-scaled by sqrt(log2 e), so the accumulator plus the key constant feeds v_exp_f32
// directly.  Tile: a workgroup = 4 waves x 64 queries (4 MFMA column tiles per wave, the query rows live in registers as B
// operands for the whole kernel), keys streamed through LDS in chunks of 64 (double buffered, one barrier per chunk); a wave runs
// 16 MFMAs per 16-byte LDS read.  Each lane keeps an online (max, sum) for its query over the 4 keys per tile it sees.
#pragma once
#include "stein.hpp"

namespace dust {

struct LogpMfmaArgs {
  int N, D, i0, n_local, JS, slice;
  const float *X;       // [N][D] particles (= prior means)
  const float *logmix;  // [N] log mixture weights
  float sw[2];          // sqrt(log2 e) / sigma_p for even / odd dimensions
  float *Z;             // [N][DPB] scaled, centred, zero-padded rows (logp_prep_kernel)
  float *hq;            // [N] -|z'|^2 / 2
  float *hj;            // [N] log2(w_j) - |z'_j|^2 / 2
  float *pM, *pL;       // [JS][n_local] slice partials (natural-log max, sum of exp relative to it): prior_finish_kernel merges
  const unsigned char *far;  // [groups][chunks] 1 = every term of (64-query group, key chunk) is negligible (pairwise_far.hpp), or nullptr
  int groups, chunks;
};

// one wave per row: scale, centre, pad; squared norm
template <int DPB>
__global__ __launch_bounds__(256) void logp_prep_kernel(const LogpMfmaArgs a) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= a.N) return;
  float acc = 0.f;
#pragma unroll
  for (int c = lane; c < DPB; c += 64) {
    float v = 0.f;
    if (c < a.D) v = (a.X[(size_t)row * a.D + c] - a.X[c]) * a.sw[c & 1];
    a.Z[(size_t)row * DPB + c] = v;
    acc = fmaf(v, v, acc);
  }
  acc = wave_sum(acc);
  if (lane == 0) {
    a.hq[row] = -0.5f * acc;
    a.hj[row] = a.logmix[row] * 1.44269504088896340736f - 0.5f * acc;
  }
}

template <int DPB>
static inline size_t pairwise_logp_mfma_lds_bytes() {
  return sizeof(float) * (2 * (size_t)64 * (DPB + 4) + 2 * 64);
}

template <int DPB>
__global__ __launch_bounds__(256, 2) void pairwise_logp_mfma_kernel(const LogpMfmaArgs a) {
  constexpr int JC = 64, NT = 256, YS = DPB + 4, NP = DPB / 16, QT = 4, R4 = DPB / 4;
  static_assert(DPB % 16 == 0, "whole 16-column pieces");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *Ys = lds;                // [2][JC][YS] key rows
  float *hs = lds + 2 * JC * YS;  // [2][JC] key constants (masked keys: -inf)
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, r16 = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = blockIdx.x, js = blockIdx.y, N = a.N;
  const int q0 = tile * (4 * 16 * QT) + wave * (16 * QT);  // first local query of this wave
  const int jbeg = js * a.slice, jend = min(N, jbeg + a.slice);
  if (jbeg >= jend) {  // (a slice behind the last key: neutral partials)
    for (int i = tid; i < 4 * 16 * QT; i += NT) {
      const int il = tile * (4 * 16 * QT) + i;
      if (il < a.n_local) {
        a.pM[(size_t)js * a.n_local + il] = -INFINITY;
        a.pL[(size_t)js * a.n_local + il] = 0.f;
      }
    }
    return;
  }
  // B operands: the wave's 64 query rows, lane (query r16 of tile t, dims 16 s + 4 g .. + 3)
  v4f bq[QT][NP];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const int qi = min(a.i0 + q0 + 16 * t + r16, N - 1);
#pragma unroll
    for (int s = 0; s < NP; ++s) bq[t][s] = *reinterpret_cast<const v4f *>(a.Z + (size_t)qi * DPB + 16 * s + 4 * g);
  }
  float m[QT], sm[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    m[t] = -3.0e38f;  // (finite: a lane whose first keys are all masked must not form inf - inf)
    sm[t] = 0.f;
  }
  // The query's OWN term - the dominant one of a spread-out set - is exact in the exact-difference pass (distance 0: logit = log w_i),
  // and here it would carry the product form's whole cancellation error, 4 eps |z_i|^2: the one key step of a wave that
  // holds its own queries (a wave-uniform test) puts hj_i + |z_i|^2 / 2 + |z_i|^2 / 2 = hj_i - 2 hq_i in the accumulator's place.
  const int qg0 = a.i0 + q0;  // the wave's first query (global index)
  float hq2[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) hq2[t] = 2.0f * a.hq[min(qg0 + 16 * t + r16, N - 1)];
  // key chunk staging: 64 rows x DPB floats = NP 16-byte pieces per lane
  v4f ky[NP];
  float hjn = 0.f;
  auto keys_issue = [&](const int j0) {
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const int f = tid + NT * u, row = f / R4, c4 = f - row * R4;
      ky[u] = *reinterpret_cast<const v4f *>(a.Z + (size_t)min(j0 + row, jend - 1) * DPB + 4 * c4);
    }
    if (tid < JC) hjn = j0 + tid < jend ? a.hj[j0 + tid] : -INFINITY;
  };
  auto keys_commit = [&](const int buf) {
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const int f = tid + NT * u, row = f / R4, c4 = f - row * R4;
      *reinterpret_cast<v4f *>(&Ys[(buf * JC + row) * YS + 4 * c4]) = ky[u];
    }
    if (tid < JC) hs[buf * JC + tid] = hjn;
  };
  // live chunks of the slice: 64 flags per ballot - wmask: some wave of the workgroup needs the chunk, omask: this wave does
  // (a.slice is a multiple of 64: chunk c = keys 64 c ..)
  const int c_beg = jbeg >> 6, c_end = (jend + JC - 1) >> 6;
  int fgb = c_beg;
  unsigned long long wmask = 0ull, omask = 0ull;
  auto far_group = [&](const int base) {
    const int cidx = base + lane;
    bool any_live = cidx < c_end, own_live = any_live;
    if (a.far && cidx < c_end) {
      any_live = own_live = false;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const int grp = tile * 4 + w;
        const bool lv = grp < a.groups && a.far[(size_t)grp * a.chunks + cidx] == 0;
        any_live = any_live || lv;
        own_live = w == wave ? lv : own_live;
      }
    }
    fgb = base;
    wmask = __ballot(any_live);
    omask = __ballot(own_live);
  };
  auto next_live = [&](int from) {  // first chunk >= from some wave needs, or c_end
    while (from < c_end) {
      if (from >= fgb + 64) far_group(from);
      const unsigned long long mk = wmask >> (from - fgb);
      if (mk) return from + (int)__builtin_ctzll(mk);
      from = fgb + 64;
    }
    return c_end;
  };
  far_group(c_beg);
  int ci = next_live(c_beg);
  if (ci < c_end) {
    keys_issue(ci * JC);
    keys_commit(0);
  }
  wg_sync();
  int buf = 0;
  while (ci < c_end) {
    const int j0 = ci * JC;
    const bool own = (omask >> (ci - fgb)) & 1ull;
    const int cn = next_live(ci + 1);
    const bool more = cn < c_end;
    if (more) keys_issue(cn * JC);  // in flight during the products
    const float *Yb = Ys + (size_t)buf * JC * YS;
#pragma unroll 1
    for (int kt = 0; kt < (own ? JC / 16 : 0); ++kt) {
      v4f acc[QT];
#pragma unroll
      for (int t = 0; t < QT; ++t) acc[t] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < NP; ++s) {
        const v4f av = *reinterpret_cast<const v4f *>(&Yb[(16 * kt + r16) * YS + 16 * s + 4 * g]);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int t = 0; t < QT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c], bq[t][s][c], acc[t], 0, 0, 0);
      }
      // the lane holds (keys 16 kt + 4 g + r, query r16 of tile t): online log-sum-exp in base 2
      const v4f h = *reinterpret_cast<const v4f *>(&hs[buf * JC + 16 * kt + 4 * g]);
      const int kb = j0 + 16 * kt;
      const bool diag = kb < qg0 + 16 * QT && kb + 16 > qg0;  // (wave-uniform: 4 of a wave's key steps at most)
#pragma unroll
      for (int t = 0; t < QT; ++t) {
        v4f x = acc[t] + h;
        if (diag) {
          const int d = (qg0 + 16 * t + r16) - (kb + 4 * g);  // the lane's query is key r = d of its four
#pragma unroll
          for (int r = 0; r < 4; ++r) x[r] = d == r ? h[r] - hq2[t] : x[r];
        }
        const float mx = fmaxf(fmaxf(x.x, x.y), fmaxf(x.z, x.w));
        const float mn = fmaxf(m[t], mx);
        const float e = (__builtin_amdgcn_exp2f(x.x - mn) + __builtin_amdgcn_exp2f(x.y - mn)) +
                        (__builtin_amdgcn_exp2f(x.z - mn) + __builtin_amdgcn_exp2f(x.w - mn));
        sm[t] = fmaf(sm[t], __builtin_amdgcn_exp2f(m[t] - mn), e);
        m[t] = mn;
      }
    }
    if (more) keys_commit(buf ^ 1);
    wg_sync();
    ci = cn;
    buf ^= 1;
  }
  // merge the 4 lane groups of a query (lanes r16, r16 + 16, + 32, + 48), then one lane per query writes the slice partial
#pragma unroll
  for (int t = 0; t < QT; ++t) {
#pragma unroll
    for (int o = 16; o < 64; o <<= 1) {
      const float mo = __shfl_xor(m[t], o), so = __shfl_xor(sm[t], o);
      const float mn = fmaxf(m[t], mo);
      sm[t] = sm[t] * __builtin_amdgcn_exp2f(m[t] - mn) + so * __builtin_amdgcn_exp2f(mo - mn);
      m[t] = mn;
    }
    const int il = q0 + 16 * t + r16;
    if (g == 0 && il < a.n_local) {
      a.pM[(size_t)js * a.n_local + il] = (m[t] + a.hq[a.i0 + il]) * 0.69314718055994531f;
      a.pL[(size_t)js * a.n_local + il] = sm[t];
    }
  }
}

}  // namespace dust
