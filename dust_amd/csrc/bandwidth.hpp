// bandwidth.hpp - kernel branch K2: iid_mp(RBF) "message passing" kernel with per-dimension median bandwidth.
//
// Replaces (reference file:line): SVMPC.phi kernel branch svmpc.py:64-74 -> iid_mp.eval composite_kernels.py:33-64
// (a Python double loop over H x d_a, each building N x N matrices) -> RBF.compute_bandwidth / RBF.eval
// base_kernels.py:53-108 (h = clamp(bw_scale * median(pairwise sq. dists) / log(N+1), 1e-5); torch.median = lower middle
// of all N^2 entries, the N zero diagonal entries included).
//
// Median without materialising N^2 values (one workgroup per scalar dimension c):
//   1. sort the N coordinates x_.c in LDS (bitonic);
//   2. bisect on the BIT PATTERN of the answer v (non-negative floats order like their bits): count(v) =
//      #{(i,j): (x_i - x_j)^2 <= v} = N + 2 sum_i #{j > i : (x_j - x_i)^2 <= v}; with x sorted the inner set is a prefix,
//      found by a binary search per lane.  The smallest v with count(v) >= rank+1 IS the rank-th order statistic.
//   Exact order statistic, as SURVEY.md section 7 requires (no histogram approximation).
// K2_SHARED (indep_controls=False: one kernel per timestep over the d_a controls) has no 1-D order, so it bisects with
// an O(N^2) count per step; no demo uses that mode.
#pragma once
#include "common.hpp"
#include "handoff.hpp"

namespace dust {

struct K2Args {
  int N, H, da, D;
  int shared;  // 1: one kernel per timestep (indep_controls=False)
  int i0, n_local;
  float bw_scale;
  float min_bw;         // RBF(minimum_bw=) base_kernels.py:44, 83-89: the clamp of every bandwidth (1e-5 by default)
  float fixed_h;        // > 0: RBF(bandwidth >= 0) base_kernels.py:66-67 - every kernel uses this h (host-evaluated), no median pass
  const float *theta;   // [N][D]
  const float *thetaT;  // [D][N]
  const float *score;   // [N][D]
  float *h;             // [G] bandwidths out (G = D, or H when shared)
  float log_n1;         // (float)log((double)N + 1.0), formed on the host (k2_bandwidth256)
  const float *h_prev;  // [G] the bandwidths of the previous call (warm start of the sorted kernels); may be h itself
  float *phi;           // [N][D]
  // optimiser step folded into k2_phi_kernel (apply != 0): what update_from_phi_kernel does in a launch of its own otherwise.  The phi
  // kernel reads the particles through the TRANSPOSED copy only, so the row-major theta may be updated in place while other
  // workgroups are still summing.
  int apply, optimizer;
  float lr, beta1, beta2, eps;
  float *theta_rw;      // [N][D] row-major particles (updated in place)
  float *thetaT_out;    // [D][N] or nullptr: the updated particles, transposed, for the NEXT iteration's bandwidth and phi kernels (a second
                        // buffer: this launch's other workgroups still read the current transposed copy) - no transpose launch between iterations
  // k2_phi2_kernel without a transposed copy (round 6, unsharded ping-pong of the particle buffers): x_rows != nullptr - the coordinates
  // are read from these ROW-MAJOR particles (strided, like the scores) and the updated particles go to theta_out (the OTHER buffer:
  // other workgroups still read x_rows); theta_out == nullptr: theta_rw is updated in place as described above
  const float *x_rows;
  float *theta_out;
  float *adam_m, *adam_v;
  uint32_t *ctr;        // {tick, iter, adam_step}
  unsigned int *fused_cnt;
  int fused_tiles;
};

__device__ __forceinline__ unsigned long long block_sum_u64(unsigned long long v, unsigned long long *scratch) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  wg_sync();
  if (lane == 0) scratch[wid] = v;
  wg_sync();
  unsigned long long r = 0;
  for (int w = 0; w < nw; ++w) r += scratch[w];
  return r;
}

// (several candidate thresholds per bisection round were measured at N = 1024 and lost: 1 -> 43 us, 2 -> 50, 3 -> 54, 7 -> 77; again in round 4
//  behind the warm start: 282 / 291 / 297 us per cfg2 tick)

// The bandwidth of scalar dimension c, computed by ONE 1024-lane workgroup (every lane returns it): `v` is lane tid's coordinate
// x_{tid,c} (INFINITY past N), xs = 2 npow2 floats of LDS.
__device__ __forceinline__ float k2_sorted_bandwidth(const K2Args &a, const int npow2, const int c, float *const xs, float v) {
  __shared__ unsigned redc[16 * 8];  // [3][8] rotating count slots, [24] the gathered answer, [25] the bandwidth
  const int tid = threadIdx.x, N = a.N;
  // bitonic sort, one element per lane (npow2 <= 1024 = blockDim): partners inside a wave (j < 64) are exchanged with a lane
  // shuffle - no barrier - and only the 10 stages with j >= 64 go through LDS (55 barrier-separated LDS passes before: ~30 of 76 us)
  // (round 4: the partner at distance 1 / 2 / 8 comes through one DPP move, at distance 4 through two - 34 of the 45 in-wave stages - and
  //  the LDS stages alternate between two buffers, so one barrier per stage suffices: a wave that passes the barrier of stage s has seen
  //  every wave finish its reads of stage s - 1, whose buffer stage s + 1 writes.  8.5 -> 5 us of the kernel's 20.)
  {
    float *const xs2 = xs + npow2;  // (the host allocates 2 npow2 floats)
    int pp = 0;
    auto cmpx = [&](const int k, const int j, const float other) {
      const bool up = (tid & k) == 0, lower = (tid & j) == 0;
      v = (lower == up) ? fminf(v, other) : fmaxf(v, other);
    };
    for (int k = 2; k <= npow2; k <<= 1) {
      for (int j = k >> 1; j >= 64; j >>= 1) {  // LDS stages
        float *const buf = pp ? xs2 : xs;
        pp ^= 1;
        if (tid < npow2) buf[tid] = v;
        wg_sync();
        cmpx(k, j, tid < npow2 ? buf[tid ^ j] : v);
      }
      // in-wave stages: j is a compile-time constant in each statement (scalar branches on k only)
      if (k > 32) cmpx(k, 32, __shfl_xor(v, 32, 64));
      if (k > 16) cmpx(k, 16, __shfl_xor(v, 16, 64));
      if (k > 8) cmpx(k, 8, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128 /* row_ror:8 */, 0xf, 0xf, false)));
      if (k > 4) {
        int o = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x104 /* row_shl:4 */, 0xf, 0x5, false);  // banks 0 / 2: the lane 4 up
        o = __builtin_amdgcn_update_dpp(o, __builtin_bit_cast(int, v), 0x114 /* row_shr:4 */, 0xf, 0xa, false);      // banks 1 / 3: the lane 4 down
        cmpx(k, 4, __builtin_bit_cast(float, o));
      }
      if (k > 2) cmpx(k, 2, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E /* quad_perm [2,3,0,1] */, 0xf, 0xf, true)));
      cmpx(k, 1, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, true)));
    }
    wg_sync();  // (the last LDS stage's reads of xs, if it was the buffer in use, are done)
    if (tid < npow2) xs[tid] = v;
    wg_sync();
  }
  const unsigned long long want = ((unsigned long long)N * N - 1ull) / 2ull + 1ull;  // rank (lower middle) + 1
  const float span = xs[N - 1] - xs[0];
  unsigned lo = 0u, hi = __float_as_uint(span * span);
  // Bisection on the bit pattern, one threshold per round.  b_v(i) = largest j >= i with
  // (x_j - x_i)^2 <= v is monotone in v, so every lane keeps a bracket [bl, br] of b for the current [lo, hi] and only
  // searches inside it: the per-lane searches shrink from log2 N steps to 1-2 as the bisection narrows.
  // (blockDim >= N is required: one particle per lane - the host launches 1024 lanes and N <= 1024 takes this kernel.)
  const bool has = tid < N;
  const float xi = has ? xs[tid] : 0.f;
  int bl = tid, br = N - 1;
  int round = 0;
  // pair counts (j > i) at the ends of the bracket: Clo = #{d2 <= float(lo - 1)} (0 below the smallest float), Chi = #{d2 <= float(hi)}; the
  // answer is the (Wc - Clo)-th smallest of the Chi - Clo pair distances in between (Wc: the rank in pairs - count(v) = N + 2 C(v))
  const unsigned Wc = (unsigned)((want - (unsigned long long)N + 1ull) / 2ull);
  unsigned Clo = 0u, Chi = (unsigned)(((unsigned long long)N * (N - 1)) / 2ull);
  // Warm start (round 3): between two SVGD iterations the particles move by lr * phi, so the median moves by a fraction of a
  // percent.  a.h[c] still holds the previous bandwidth: two probes at v_prev (1 -+ 2^-7) - if they bracket the rank, the bisection
  // starts from 2^17 bit patterns instead of 2^31 (17 rounds instead of 31; the median moves 0.2-0.8 % per iteration at cfg2: +-2^-9 misses too often, +-2^-5 costs two more rounds); if
  // not (first call, a jump), nothing is lost but the two probes.  The answer is the same exact order statistic either way.
  {
    const float hp = a.h_prev[c];
    const float vp = hp > a.min_bw && a.bw_scale > 0.f ?  /* (a clamped bandwidth says nothing about the median) */ (hp / a.bw_scale) * (float)log((double)N + 1.0) : 0.f;
    if (vp > 0.f && vp < span * span) {  // (wave-uniform: one value per workgroup)
      const unsigned plo = __float_as_uint(vp * (1.0f - 0.0078125f)), phi_ = __float_as_uint(vp * (1.0f + 0.0078125f));
      int bb[2];
      unsigned cc[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const float v = __uint_as_float(q ? phi_ : plo);
        int l = q ? bb[0] : tid, r = N - 1;
        if (has)
          while (l < r) {
            const int m = (l + r + 1) >> 1;
            const float dlt = xs[m] - xi;
            if (dlt * dlt <= v) l = m;
            else r = m - 1;
          }
        bb[q] = l;
        cc[q] = has ? (unsigned)(l - tid) : 0u;
      }
      // block sums of the two counts: wave sums by DPP (per-lane counts < 2^10: exact in fp32), one LDS atomic per wave and count
      __shared__ unsigned red2[2];
      if (tid < 2) red2[tid] = 0u;
      wg_sync();
      const unsigned w0 = (unsigned)wave_sum((float)cc[0]), w1 = (unsigned)wave_sum((float)cc[1]);
      if ((tid & 63) == 0) {
        atomicAdd(&red2[0], w0);
        atomicAdd(&red2[1], w1);
      }
      wg_sync();
      const unsigned long long c_lo = red2[0], c_hi = red2[1];
      const bool lo_below = 2ull * c_lo + (unsigned long long)N < want, hi_reaches = !(2ull * c_hi + (unsigned long long)N < want);
      if (plo < phi_ && phi_ <= hi) {
        // a probe that misses still halves the problem: it becomes one end of the bracket, and up to two wider probes on the open side
        // (v_prev (1 -+ 2^-4), (1 -+ 2^-2): the first iteration of a tick follows the roll, which moves the median by a few per cent)
        // close it - a round each instead of the ~ 12 that a bracket ending at 0 or at the span costs
        if (lo_below) {
          lo = plo + 1u;
          bl = bb[0];
          Clo = (unsigned)c_lo;
        }
        if (hi_reaches) {
          hi = phi_;
          br = bb[1];
          Chi = (unsigned)c_hi;
        } else {  // the answer lies above both probes
          lo = phi_ + 1u;
          bl = bb[1];
          Clo = (unsigned)c_hi;
        }
        if (!lo_below) {  // ... or below both
          hi = plo;
          br = bb[0];
          Chi = (unsigned)c_lo;
        }
        const bool open_hi = !hi_reaches, open_lo = !lo_below;
        if (open_hi || open_lo) {
#pragma unroll 1
          for (int e = 4; e >= 2; e -= 2) {
            const float wv = open_hi ? vp * (1.0f + 1.0f / (float)(1 << e)) : vp * (1.0f - 1.0f / (float)(1 << e));
            const unsigned pv = __float_as_uint(wv);
            if (!(pv >= lo && pv < hi)) break;  // (wave-uniform)
            int l = bl, r = br;
            if (has)
              while (l < r) {
                const int m = (l + r + 1) >> 1;
                const float dlt = xs[m] - xi;
                if (dlt * dlt <= wv) l = m;
                else r = m - 1;
              }
            if (tid == 0) red2[0] = 0u;
            wg_sync();
            const unsigned ws = (unsigned)wave_sum(has ? (float)(l - tid) : 0.f);
            if ((tid & 63) == 0) atomicAdd(&red2[0], ws);
            wg_sync();
            const unsigned cw = red2[0];
            wg_sync();  // (red2 is re-armed by the next probe)
            if (cw >= Wc) {
              hi = pv;
              br = l;
              Chi = cw;
              if (open_hi) break;  // closed
            } else {
              lo = pv + 1u;
              bl = l;
              Clo = cw;
              if (open_lo) break;  // closed
            }
          }
        }
      }
    }
  }
  if (tid < 24) redc[tid] = 0u;
  __shared__ float cand[64];
  __shared__ unsigned ncand;
  if (tid == 0) ncand = 0u;
  wg_sync();
  // Narrowing (round 4).  Every probe keeps the invariant C(lo - 1) < Wc <= C(hi), so ANY threshold inside the bracket is a legal probe:
  // while the pair count is locally linear in the threshold (it is, to ~ 1 / sqrt(K) over the K candidates of a warm-started bracket) the
  // interpolated threshold leaves a few dozen candidates after one or two rounds instead of halving them 17 times; a probe that does
  // not at least halve the candidates is followed by a plain bit-pattern midpoint (termination as before).  With <= 64 candidates
  // left they are gathered and sorted by one wave: the (Wc - Clo)-th smallest IS the order statistic - no more rounds.
  bool interp = true;
  while (lo < hi) {
    const unsigned K = Chi - Clo;
    if (K <= 64u) break;
    unsigned mid = lo + ((hi - lo) >> 1);
    if (interp) {
      const float vlo = lo ? __uint_as_float(lo - 1u) : 0.f, vhi = __uint_as_float(hi);
      const float fr = ((float)(Wc - Clo) - 0.5f) / (float)K;
      const unsigned mi = __float_as_uint(fmaf(vhi - vlo, fr, vlo));
      mid = min(max(mi, lo), hi - 1u);
    }
    const float v = __uint_as_float(mid);
    int l = bl, r = br;
    if (has)
      while (l < r) {
        const int m = (l + r + 1) >> 1;
        const float dlt = xs[m] - xi;
        if (dlt * dlt <= v) l = m;
        else r = m - 1;
      }
    unsigned cnt = has ? (unsigned)(l - tid) : 0u;
    {
      const int lane = tid & 63;
      unsigned *slot = redc + (round % 3) * 8, *other = redc + ((round + 1) % 3) * 8;
      cnt = (unsigned)wave_sum((float)cnt);  // (per-lane counts < 2^10: exact in fp32; DPP, not 6 LDS-crossbar shuffles)
      if (lane == 0) atomicAdd(&slot[0], cnt);
      if (tid == 64) other[0] = 0u;
      wg_sync();
      cnt = slot[0];
      ++round;
    }
    if (cnt >= Wc) {  // the answer is <= mid
      interp = (cnt - Clo) * 2u <= K;
      hi = mid;
      Chi = cnt;
      br = l;
    } else {
      interp = (Chi - cnt) * 2u <= K;
      lo = mid + 1u;
      Clo = cnt;
      bl = l;
    }
  }
  if (lo < hi) {  // <= 64 candidates: lane i's are the pairs (i, j), bl < j <= br
    if (has)
      for (int j = bl + 1; j <= br; ++j) {
        const float dlt = xs[j] - xi;
        cand[atomicAdd(&ncand, 1u)] = dlt * dlt;
      }
    wg_sync();
    if (tid < 64) {
      float x = tid < (int)ncand ? cand[tid] : INFINITY;
      for (int k = 2; k <= 64; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
          const float o = __shfl_xor(x, j, 64);
          const bool up = (tid & k) == 0, lower = (tid & j) == 0;
          x = (lower == up) ? fminf(x, o) : fmaxf(x, o);
        }
      const float ans = __shfl(x, (int)(Wc - Clo) - 1, 64);
      if (tid == 0) redc[24] = __float_as_uint(ans);
    }
    wg_sync();
    lo = redc[24];
  }
  if (tid == 0) {
    float h = __uint_as_float(lo);
    h = h / (float)log((double)N + 1.0);  // base_kernels.py:77
    h = a.bw_scale * h;
    redc[25] = __float_as_uint(fmaxf(h, a.min_bw));
  }
  wg_sync();
  return __uint_as_float(redc[25]);
}

// one workgroup per independent scalar dimension c
__global__ __launch_bounds__(1024) void k2_bandwidth_sorted_kernel(const K2Args a, int npow2) {
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [2 npow2]
  const int tid = threadIdx.x, N = a.N, c = blockIdx.x;
  const float h = k2_sorted_bandwidth(a, npow2, c, xs, tid < N ? a.thetaT[(size_t)c * N + tid] : INFINITY);
  if (tid == 0) a.h[c] = h;
}

// The same order statistic by a 256-LANE workgroup, four sorted positions per lane (round 6).  It exists so that the bandwidths can ride
// in the prior + rollout launch (fused.hpp: fused_prior_rollout_kernel's workgroups are 256 lanes): they read the particles only, and
// as a launch of their own behind the rollouts they were 16 of the 43 us of a cfg2 / K2 iteration.  Element e = 4 tid + r of the
// bitonic network: the distance-1 / -2 stages are register exchanges, distances 4-128 cross lanes (DPP, two shuffles), only the three
// stages at distance 256 / 512 cross waves through LDS.  Selection as above, a lane searching for its four positions at once (four
// independent chains of LDS reads in flight).  Reads the ROW-MAJOR particles (a.theta): no transposed copy is needed in front of it.
// lds: K2_BW256_LDS floats.  Writes a.h[c] (a.h_prev may be the same buffer: one workgroup per dimension).
enum { K2_BW256_LDS = 3072 + 128 };
#ifdef K2_STAMPS  // (tools/k2_bw_probe.hip: where the role's time goes)
__device__ unsigned long long k2_stamps[64 * 16];
#define K2_STAMP(k)                                                                                                   \
  do {                                                                                                                \
    if (threadIdx.x == 0) k2_stamps[blockIdx.x * 16 + (k)] = (unsigned long long)__builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define K2_STAMP(k) \
  do {              \
  } while (0)
#endif
__device__ __forceinline__ void k2_bandwidth256(const K2Args &a, const int c, float *const lds) {
  // (no static LDS: 388 bytes of it in the prior + rollout kernel took that launch from 6 to 5 resident workgroups per CU - every
  //  workgroup of the cfg2 grid resident at once before, two rounds after: 17 -> 42 us)
  unsigned *const redq = (unsigned *)(lds + 3072);  // [3][8] rotating count slots (two counts x four waves), [24] the gathered answer
  float *const candq = lds + 3072 + 32;             // [64]
  unsigned &ncandq = *(unsigned *)(lds + 3072 + 96);
  const int tid = threadIdx.x, N = a.N, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  float v[4];
  K2_STAMP(0);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int e = 4 * tid + r;
    v[r] = e < N ? a.theta[(size_t)e * a.D + c] : INFINITY;
  }
  if (tid == 0) ncandq = 0u;
  if (v[0] == 12345.f) K2_STAMP(15);  // (waits for the loads)
  K2_STAMP(1);
  {
    auto pair = [](float &x, float &y, const bool up) {  // (x, y) ascending when up
      const float lo_ = fminf(x, y), hi_ = fmaxf(x, y);
      x = up ? lo_ : hi_;
      y = up ? hi_ : lo_;
    };
    int pp = 0;
#pragma unroll
    for (int k = 2; k <= 1024; k <<= 1) {
      const bool up = (tid & (k >> 2)) == 0;  // (k >= 4; k = 2 is the register stage below alone)
      auto cross = [&](const int m, const float (&o)[4]) {
        const bool keep_min = ((tid & m) == 0) == up;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = keep_min ? fminf(v[r], o[r]) : fmaxf(v[r], o[r]);
      };
      auto lds_stage = [&](const int m) {
        float *const buf = lds + (pp ? 1024 : 0);
        pp ^= 1;
#pragma unroll
        for (int r = 0; r < 4; ++r) buf[r * 256 + tid] = v[r];
        wg_sync();
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = buf[r * 256 + (tid ^ m)];
        cross(m, o);
      };
      if (k >= 1024) lds_stage(128);
      if (k >= 512) lds_stage(64);
      if (k >= 256) {
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = __shfl_xor(v[r], 32, 64);
        cross(32, o);
      }
      if (k >= 128) {
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = __shfl_xor(v[r], 16, 64);
        cross(16, o);
      }
      if (k >= 64) {
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[r]), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
        cross(8, o);
      }
      if (k >= 32) {
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[r]), 0x104 /* row_shl:4 */, 0xf, 0x5, false);  // banks 0 / 2: the lane 4 up
          t = __builtin_amdgcn_update_dpp(t, __builtin_bit_cast(int, v[r]), 0x114 /* row_shr:4 */, 0xf, 0xa, false);      // banks 1 / 3: the lane 4 down
          o[r] = __builtin_bit_cast(float, t);
        }
        cross(4, o);
      }
      if (k >= 16) {
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[r]), 0x4E /* quad_perm [2,3,0,1] */, 0xf, 0xf, true));
        cross(2, o);
      }
      if (k >= 8) {
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[r]), 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, true));
        cross(1, o);
      }
      if (k >= 4) {  // distance 2 inside the lane
        pair(v[0], v[2], up);
        pair(v[1], v[3], up);
        pair(v[0], v[1], up);
        pair(v[2], v[3], up);
      } else {  // k = 2: (0, 1) ascending, (2, 3) descending
        pair(v[0], v[1], true);
        pair(v[2], v[3], false);
      }
    }
  }
  K2_STAMP(2);
  float *const xs = lds + 2048;
  *(v4f *)(xs + 4 * tid) = v4f{v[0], v[1], v[2], v[3]};
  wg_sync();
  const unsigned long long want = ((unsigned long long)N * N - 1ull) / 2ull + 1ull;  // rank (lower middle) + 1
  const float span = xs[N - 1] - xs[0];
  unsigned lo = 0u, hi = __float_as_uint(span * span);
  bool has[4];
  int bl[4], br[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    has[r] = 4 * tid + r < N;
    bl[r] = 4 * tid + r;
    br[r] = has[r] ? N - 1 : 4 * tid + r;
  }
  // b_v(i) = largest j >= i with (x_j - x_i)^2 <= thr, searched inside [l, rr] for the lane's four positions at once
  auto search = [&](const float thr, int (&l)[4], const int (&r0)[4]) {
    // Branch-free steps: the four LDS reads of a step are issued together (with a branch per position each read waited for its own
    // compare).  l is always a position where the predicate holds (it starts at the position itself: distance 0), so a finished
    // search (l == rr: m == l) re-reads its own answer and changes nothing; padded positions (e >= N: INFINITY - INFINITY is NaN, the
    // predicate fails) start finished and only ever lower their private rr.
    int rr[4] = {r0[0], r0[1], r0[2], r0[3]};
    bool any;
    do {
      int m[4];
      float x[4];
      any = false;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        any |= l[r] < rr[r];
        m[r] = (l[r] + rr[r] + 1) >> 1;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) x[r] = xs[m[r]];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float dlt = x[r] - v[r];
        const bool ok = dlt * dlt <= thr;
        l[r] = ok ? m[r] : l[r];
        rr[r] = ok ? rr[r] : m[r] - 1;
      }
    } while (any);
  };
  auto count_of = [&](const int (&l)[4]) {
    unsigned n = 0u;
#pragma unroll
    for (int r = 0; r < 4; ++r) n += (unsigned)(l[r] - (4 * tid + r));  // (padded positions: l stays at the position)
    return n;
  };
  int round = 0;
  // block sums of up to two per-lane counts (< 2^12 each: the wave sums are exact in fp32); one barrier per call - the slot sets rotate
  auto block_counts = [&](const unsigned n0, const unsigned n1, unsigned &s0, unsigned &s1) {
    unsigned *const slot = redq + (round % 3) * 8;
    const unsigned w0 = (unsigned)wave_sum((float)n0), w1 = (unsigned)wave_sum((float)n1);
    if (lane == 0) {
      slot[w] = w0;
      slot[4 + w] = w1;
    }
    wg_sync();
    ++round;
    s0 = slot[0] + slot[1] + slot[2] + slot[3];
    s1 = slot[4] + slot[5] + slot[6] + slot[7];
  };
  auto block_count = [&](const unsigned n0) {
    unsigned *const slot = redq + (round % 3) * 8;
    const unsigned w0 = (unsigned)wave_sum((float)n0);
    if (lane == 0) slot[w] = w0;
    wg_sync();
    ++round;
    const uint4 q = *(const uint4 *)slot;
    return q.x + q.y + q.z + q.w;
  };
  const unsigned Wc = (unsigned)((want - (unsigned long long)N + 1ull) / 2ull);
  unsigned Clo = 0u, Chi = (unsigned)(((unsigned long long)N * (N - 1)) / 2ull);
  {  // warm start from the previous bandwidth (see k2_sorted_bandwidth)
    const float hp = a.h_prev[c];
    const float vp = hp > a.min_bw && a.bw_scale > 0.f ? (hp / a.bw_scale) * a.log_n1 : 0.f;
    if (vp > 0.f && vp < span * span) {
      const unsigned plo = __float_as_uint(vp * (1.0f - 0.0078125f)), phi_ = __float_as_uint(vp * (1.0f + 0.0078125f));
      int b0[4] = {bl[0], bl[1], bl[2], bl[3]};
      search(__uint_as_float(plo), b0, br);
      // the second threshold is 1.6 % above the first: its answers lie a few positions above the first one's - a 4-step search in
      // [b0, b0 + 15], and only a lane that ends on that range's last position goes on to the rest
      int b1[4] = {b0[0], b0[1], b0[2], b0[3]};
      {
        int cap[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) cap[r] = min(b0[r] + 15, br[r]);
        search(__uint_as_float(phi_), b1, cap);
        bool more = false;
#pragma unroll
        for (int r = 0; r < 4; ++r) more |= b1[r] == cap[r] && cap[r] < br[r];
        if (__builtin_amdgcn_ballot_w64(more) != 0ull) search(__uint_as_float(phi_), b1, br);
      }
      unsigned c_lo, c_hi;
      block_counts(count_of(b0), count_of(b1), c_lo, c_hi);
      const bool lo_below = 2ull * c_lo + (unsigned long long)N < want, hi_reaches = !(2ull * c_hi + (unsigned long long)N < want);
      if (plo < phi_ && phi_ <= hi) {
        if (lo_below) {
          lo = plo + 1u;
#pragma unroll
          for (int r = 0; r < 4; ++r) bl[r] = b0[r];
          Clo = c_lo;
        }
        if (hi_reaches) {
          hi = phi_;
#pragma unroll
          for (int r = 0; r < 4; ++r) br[r] = b1[r];
          Chi = c_hi;
        } else {  // the answer lies above both probes
          lo = phi_ + 1u;
#pragma unroll
          for (int r = 0; r < 4; ++r) bl[r] = b1[r];
          Clo = c_hi;
        }
        if (!lo_below) {  // ... or below both
          hi = plo;
#pragma unroll
          for (int r = 0; r < 4; ++r) br[r] = b0[r];
          Chi = c_lo;
        }
        const bool open_hi = !hi_reaches, open_lo = !lo_below;
        if (open_hi || open_lo) {
#pragma unroll 1
          for (int e = 4; e >= 2; e -= 2) {
            const float wv = open_hi ? vp * (1.0f + 1.0f / (float)(1 << e)) : vp * (1.0f - 1.0f / (float)(1 << e));
            const unsigned pv = __float_as_uint(wv);
            if (!(pv >= lo && pv < hi)) break;  // (workgroup-uniform)
            int l[4] = {bl[0], bl[1], bl[2], bl[3]};
            search(wv, l, br);
            const unsigned cw = block_count(count_of(l));
            if (cw >= Wc) {
              hi = pv;
#pragma unroll
              for (int r = 0; r < 4; ++r) br[r] = l[r];
              Chi = cw;
              if (open_hi) break;  // closed
            } else {
              lo = pv + 1u;
#pragma unroll
              for (int r = 0; r < 4; ++r) bl[r] = l[r];
              Clo = cw;
              if (open_lo) break;  // closed
            }
          }
        }
      }
    }
  }
  K2_STAMP(3);
  bool interp = true;
  int nrounds = 0;
  while (lo < hi) {  // narrowing: interpolated thresholds while they pay, bit-pattern midpoints otherwise (see k2_sorted_bandwidth)
    const unsigned K = Chi - Clo;
    if (K <= 64u) break;
    unsigned mid = lo + ((hi - lo) >> 1);
    if (interp) {
      const float vlo = lo ? __uint_as_float(lo - 1u) : 0.f, vhi = __uint_as_float(hi);
      const float fr = ((float)(Wc - Clo) - 0.5f) * __builtin_amdgcn_rcpf((float)K);  // (any threshold inside the bracket is a legal probe)
      const unsigned mi = __float_as_uint(fmaf(vhi - vlo, fr, vlo));
      mid = min(max(mi, lo), hi - 1u);
    }
    int l[4] = {bl[0], bl[1], bl[2], bl[3]};
    ++nrounds;
    search(__uint_as_float(mid), l, br);
    const unsigned cnt = block_count(count_of(l));
    if (cnt >= Wc) {  // the answer is <= mid
      interp = (cnt - Clo) * 2u <= K;
      hi = mid;
      Chi = cnt;
#pragma unroll
      for (int r = 0; r < 4; ++r) br[r] = l[r];
    } else {
      interp = (Chi - cnt) * 2u <= K;
      lo = mid + 1u;
      Clo = cnt;
#pragma unroll
      for (int r = 0; r < 4; ++r) bl[r] = l[r];
    }
  }
  K2_STAMP(4);
#ifdef K2_STAMPS
  if (tid == 0) k2_stamps[blockIdx.x * 16 + 8] = nrounds;
#endif
  if (lo < hi) {  // <= 64 candidates: position i's are the pairs (i, j), bl < j <= br
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (has[r])
        for (int j = bl[r] + 1; j <= br[r]; ++j) {
          const float dlt = xs[j] - v[r];
          candq[atomicAdd(&ncandq, 1u)] = dlt * dlt;
        }
    wg_sync();
    if (tid < 64) {
      float x = tid < (int)ncandq ? candq[tid] : INFINITY;
      for (int k = 2; k <= 64; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
          const float o = __shfl_xor(x, j, 64);
          const bool up = (tid & k) == 0, lower = (tid & j) == 0;
          x = (lower == up) ? fminf(x, o) : fmaxf(x, o);
        }
      const float ans = __shfl(x, (int)(Wc - Clo) - 1, 64);
      if (tid == 0) redq[24] = __float_as_uint(ans);
    }
    wg_sync();
    lo = redq[24];
  }
  if (tid == 0) {
    float h = __uint_as_float(lo);
    h = h / a.log_n1;  // base_kernels.py:77 ((float)log(N + 1) formed on the host)
    h = a.bw_scale * h;
    a.h[c] = fmaxf(h, a.min_bw);
  }
  K2_STAMP(5);
  (void)nrounds;
}

// (the role as a launch of its own - DUST_K2_FORM=3, a measuring aid: its duration in a kernel trace is the role's time without the
//  rollout waves around it)
__global__ __launch_bounds__(256) void k2_bandwidth256_kernel(const K2Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds256[];
  k2_bandwidth256(a, (int)blockIdx.x, lds256);
}

// N > 1024 (several particles per lane): plain bisection, a full binary search per particle and round
__global__ __launch_bounds__(1024) void k2_bandwidth_sorted_big_kernel(const K2Args a, int npow2) {
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [npow2]
  __shared__ unsigned long long red64[16];
  const int tid = threadIdx.x, nt = blockDim.x, N = a.N, c = blockIdx.x;
  for (int i = tid; i < npow2; i += nt) xs[i] = i < N ? a.thetaT[(size_t)c * N + i] : INFINITY;
  wg_sync();
  for (int k = 2; k <= npow2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < npow2; i += nt) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const float u = xs[i], v = xs[ixj];
          const bool up = (i & k) == 0;
          if ((u > v) == up) {
            xs[i] = v;
            xs[ixj] = u;
          }
        }
      }
      wg_sync();
    }
  const unsigned long long want = ((unsigned long long)N * N - 1ull) / 2ull + 1ull;  // rank (lower middle) + 1
  const float span = xs[N - 1] - xs[0];
  unsigned lo = 0u, hi = __float_as_uint(span * span);
  while (lo < hi) {
    const unsigned mid = lo + ((hi - lo) >> 1);
    const float v = __uint_as_float(mid);
    unsigned long long cnt = 0;
    for (int i = tid; i < N; i += nt) {
      const float xi = xs[i];
      int l = i, r = N - 1;  // largest j in [i, N-1] with (x_j - x_i)^2 <= v  (j = i always qualifies)
      while (l < r) {
        const int m = (l + r + 1) >> 1;
        const float dlt = xs[m] - xi;
        if (dlt * dlt <= v) l = m;
        else r = m - 1;
      }
      cnt += (unsigned long long)(l - i);
    }
    cnt = 2ull * block_sum_u64(cnt, red64) + (unsigned long long)N;
    if (cnt >= want) hi = mid;
    else lo = mid + 1;
  }
  if (tid == 0) {
    float h = __uint_as_float(lo);
    h = h / (float)log((double)N + 1.0);  // base_kernels.py:77
    h = a.bw_scale * h;
    a.h[c] = fmaxf(h, a.min_bw);
  }
}

// K2_SHARED: one workgroup per timestep, O(N^2) count per bisection step
__global__ __launch_bounds__(1024) void k2_bandwidth_pairs_kernel(const K2Args a) {
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [da][N]
  __shared__ unsigned long long red64[16];
  __shared__ float redf[32];
  const int tid = threadIdx.x, nt = blockDim.x, N = a.N, da = a.da, g = blockIdx.x;
  for (int idx = tid; idx < da * N; idx += nt) xs[idx] = a.thetaT[(size_t)(g * da) * N + idx];
  wg_sync();
  float mx = 0.f;
  for (int i = tid; i < N; i += nt)
    for (int j = 0; j < N; ++j) {
      float d2 = 0.f;
      for (int q = 0; q < da; ++q) {
        const float dl = xs[q * N + i] - xs[q * N + j];
        d2 = fmaf(dl, dl, d2);
      }
      mx = fmaxf(mx, d2);
    }
  mx = block_reduce<RED_MAX>(mx, redf);
  const unsigned long long want = ((unsigned long long)N * N - 1ull) / 2ull + 1ull;
  unsigned lo = 0u, hi = __float_as_uint(mx);
  while (lo < hi) {
    const unsigned mid = lo + ((hi - lo) >> 1);
    const float v = __uint_as_float(mid);
    unsigned long long cnt = 0;
    for (int i = tid; i < N; i += nt)
      for (int j = 0; j < N; ++j) {
        float d2 = 0.f;
        for (int q = 0; q < da; ++q) {
          const float dl = xs[q * N + i] - xs[q * N + j];
          d2 = fmaf(dl, dl, d2);
        }
        cnt += d2 <= v ? 1ull : 0ull;
      }
    cnt = block_sum_u64(cnt, red64);
    if (cnt >= want) hi = mid;
    else lo = mid + 1;
  }
  if (tid == 0) {
    float h = __uint_as_float(lo);
    h = h / (float)log((double)N + 1.0);
    h = a.bw_scale * h;
    a.h[g] = fmaxf(h, a.min_bw);
  }
}

// phi_ic = mean_j K^c_ij score_jc + mean_j K^c_ij (x_ic - x_jc) 2/h_c   (svmpc.py:69-73, base_kernels.py:100-101)
// grid = (ceil(n_local/64), G) x 256 lanes: lane = (particle i of a 64-tile, one of 4 wave-uniform slices of the other
// particles j).  The j-columns (coordinates and scores of the group's dimension(s)) are staged through LDS in chunks of
// K2_JT rows and read back as broadcasts; the 4 slice partials are combined in slice order (reproducible).
// exp(-d2/h): exact fp32 division as the reference, then the bare v_exp_f32 (error ~|x| 2^-24, as in the K1 Gram value);
// the 2/h factor of the repulsive term is applied once to the sum.
enum { K2_JT = 2048 };
template <int GD /* dimensions per kernel group: 1, or d_a (<= 2) when shared */>
__global__ __launch_bounds__(256) void k2_phi_kernel(const K2Args a) {
  __shared__ float xcol[GD][K2_JT], scol[GD][K2_JT];
  __shared__ float part[4][64][4];
  const int N = a.N, D = a.D, da = a.da;
  const int g = blockIdx.y;
  constexpr int gd = GD;
  const int c0 = g * gd;
  const int ii = threadIdx.x & 63, js = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int il = blockIdx.x * 64 + ii;
  const bool on = il < a.n_local;
  const int i = a.i0 + (on ? il : 0);
  const float h = a.h[g];
  // k = exp(-d2 / h) as exp2(d2 * ce) with ce = -log2(e) / h formed once: a division per PAIR (v_div_scale / v_rcp / 4 FMAs /
  // v_div_fmas / v_div_fixup) was half of this kernel's instructions (26 -> 14 us at cfg2 / K2); the argument differs from the
  // divided one by <= 1.2e-7 relative, i.e. the kernel value by <= |arg| 1.2e-7 - inside what the bare v_exp_f32 already leaves
  const float ce = -1.44269504088896340736f / h;
  float xi[2] = {0.f, 0.f}, g1[2] = {0.f, 0.f}, g2[2] = {0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 2; ++q)
    if (q < gd) xi[q] = a.thetaT[(size_t)(c0 + q) * N + i];
  for (int jb = 0; jb < N; jb += K2_JT) {
    const int jn = min(K2_JT, N - jb);
    wg_sync();
    for (int idx = threadIdx.x; idx < gd * jn; idx += 256) {
      const int q = idx / jn, j = idx - q * jn;
      xcol[q][j] = a.thetaT[(size_t)(c0 + q) * N + jb + j];
      scol[q][j] = a.score[(size_t)(jb + j) * D + c0 + q];
    }
    wg_sync();
    const int per = (jn + 3) >> 2, j0 = js * per, j1 = min(jn, j0 + per);
#pragma unroll 8
    for (int j = j0; j < j1; ++j) {
      float d2 = 0.f, df[2];
#pragma unroll
      for (int q = 0; q < 2; ++q)
        if (q < gd) {
          df[q] = xi[q] - xcol[q][j];
          d2 = fmaf(df[q], df[q], d2);
        }
      const float k = __builtin_amdgcn_exp2f(d2 * ce);
#pragma unroll
      for (int q = 0; q < 2; ++q)
        if (q < gd) {
          g1[q] = fmaf(k, scol[q][j], g1[q]);
          g2[q] = fmaf(k, df[q], g2[q]);
        }
    }
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    part[js][ii][q] = g1[q];
    part[js][ii][2 + q] = g2[q];
  }
  wg_sync();
  if (js == 0 && on) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
      if (q < gd) {
        float s1 = part[0][ii][q], s2 = part[0][ii][2 + q];
        for (int r = 1; r < 4; ++r) {
          s1 += part[r][ii][q];
          s2 += part[r][ii][2 + q];
        }
        const size_t o = (size_t)i * D + c0 + q;
        const float phi = s1 / (float)N + ((s2 * 2.0f) / h) / (float)N;
        a.phi[o] = phi;
        if (a.apply) {  // (update_from_phi_kernel's element)
          float th = a.theta_rw[o];
          const float gr = -phi;
          if (a.optimizer == DUST_OPT_SGD) {
            th = fmaf(-a.lr, gr, th);
          } else {
            float m = a.adam_m[o], v = a.adam_v[o];
            th = adam_step(th, gr, m, v, a.lr, a.beta1, a.beta2, a.eps, (float)a.ctr[2]);
            a.adam_m[o] = m;
            a.adam_v[o] = v;
          }
          a.theta_rw[o] = th;
          if (a.thetaT_out) a.thetaT_out[(size_t)(c0 + q) * N + i] = th;
        }
      }
  }
  if (a.apply && blockIdx.x == 0 && blockIdx.y == 0) {  // (... and its bookkeeping: the fused launches' counters re-armed, the iteration counted)
    for (int t = threadIdx.x; t < a.fused_tiles; t += 256) a.fused_cnt[t * 32] = 0u;
    if (threadIdx.x == 0) a.ctr[1] += 1u;
  }
}

// The same for one dimension per kernel group (indep_controls = True: every demo) with TWO queries per lane (i and i + 64 of a 128-query
// tile) in packed fp32: per key one packed subtract, two packed multiplies, two exponentials and two packed FMAs serve both queries
// (19 instead of 25 issue cycles per pair, tools/valu_rate_probe.hip) and the broadcast reads of the key and its score are shared.
// Element for element the operations of k2_phi_kernel<1>, in its order: the same bits.
__global__ __launch_bounds__(256) void k2_phi2_kernel(const K2Args a) {
  __shared__ float xcol[K2_JT], scol[K2_JT];
  __shared__ float part[4][128][2];
  const int N = a.N, D = a.D;
  const int g = blockIdx.y;
  const int ii = threadIdx.x & 63, js = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int ila = blockIdx.x * 128 + ii, ilb = ila + 64;
  const bool ona = ila < a.n_local, onb = ilb < a.n_local;
  const int ia = a.i0 + (ona ? ila : 0), ib = a.i0 + (onb ? ilb : 0);
  const float h = a.h[g];
  const float ce = -1.44269504088896340736f / h;
  const v2f ce2 = {ce, ce};
  const float *const xr = a.x_rows;  // (uniform)
  const v2f xi = xr ? v2f{xr[(size_t)ia * D + g], xr[(size_t)ib * D + g]} : v2f{a.thetaT[(size_t)g * N + ia], a.thetaT[(size_t)g * N + ib]};
  v2f g1 = {0.f, 0.f}, g2 = {0.f, 0.f};
  for (int jb = 0; jb < N; jb += K2_JT) {
    const int jn = min(K2_JT, N - jb);
    wg_sync();
    for (int j = threadIdx.x; j < jn; j += 256) {
      xcol[j] = xr ? xr[(size_t)(jb + j) * D + g] : a.thetaT[(size_t)g * N + jb + j];
      scol[j] = a.score[(size_t)(jb + j) * D + g];
    }
    wg_sync();
    const int per = (jn + 3) >> 2, j0 = js * per, j1 = min(jn, j0 + per);
#pragma unroll 8
    for (int j = j0; j < j1; ++j) {
      const float xc = xcol[j], sc = scol[j];
      const v2f df = xi - v2f{xc, xc};
      const v2f arg = (df * df) * ce2;
      const v2f k = {__builtin_amdgcn_exp2f(arg.x), __builtin_amdgcn_exp2f(arg.y)};
      g1 = __builtin_elementwise_fma(k, v2f{sc, sc}, g1);
      g2 = __builtin_elementwise_fma(k, df, g2);
    }
  }
  part[js][ii][0] = g1.x;
  part[js][ii][1] = g2.x;
  part[js][ii + 64][0] = g1.y;
  part[js][ii + 64][1] = g2.y;
  wg_sync();
  if (threadIdx.x < 128) {
    const int q = threadIdx.x, il = blockIdx.x * 128 + q;
    if (il < a.n_local) {
      const int i = a.i0 + il;
      float s1 = part[0][q][0], s2 = part[0][q][1];
      for (int r = 1; r < 4; ++r) {
        s1 += part[r][q][0];
        s2 += part[r][q][1];
      }
      const size_t o = (size_t)i * D + g;
      const float phi = s1 / (float)N + ((s2 * 2.0f) / h) / (float)N;
      a.phi[o] = phi;
      if (a.apply) {  // (update_from_phi_kernel's element)
        float th = a.theta_rw[o];
        const float gr = -phi;
        if (a.optimizer == DUST_OPT_SGD) {
          th = fmaf(-a.lr, gr, th);
        } else {
          float m = a.adam_m[o], v = a.adam_v[o];
          th = adam_step(th, gr, m, v, a.lr, a.beta1, a.beta2, a.eps, (float)a.ctr[2]);
          a.adam_m[o] = m;
          a.adam_v[o] = v;
        }
        (a.theta_out ? a.theta_out : a.theta_rw)[o] = th;
        if (a.thetaT_out) a.thetaT_out[(size_t)g * N + i] = th;
      }
    }
  }
  if (a.apply && blockIdx.x == 0 && blockIdx.y == 0) {
    for (int t = threadIdx.x; t < a.fused_tiles; t += 256) a.fused_cnt[t * 32] = 0u;
    if (threadIdx.x == 0) a.ctr[1] += 1u;
  }
}

// k2_phi2_kernel's arithmetic on 1024 lanes (round 6; N <= K2_P3_N): 128 queries per workgroup (two per lane, packed fp32), the 16 waves
// take 16 key slices, the slice partials are summed in slice order.  The 256-lane launch was 10.9 us per cfg2 iteration of pure
// latency (a lane walked 256 keys); here a lane walks 64.  Sharded and unsharded contexts take the same kernel: the slices cut the key
// range [0, N), whoever owns the queries - the ranks of a sharded run stay bit-identical to the unsharded one.
enum { K2_P3_Q = 128, K2_P3_N = 2048 };
__global__ __launch_bounds__(1024) void k2_phi3_kernel(const K2Args a) {
  __shared__ float xcol[K2_P3_N], scol[K2_P3_N];
  __shared__ float part[16][K2_P3_Q][2];
  const int N = a.N, D = a.D, g = blockIdx.y, tid = threadIdx.x;
  const int ii = tid & 63, js = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ila = blockIdx.x * K2_P3_Q + ii, ilb = ila + 64;
  const bool ona = ila < a.n_local, onb = ilb < a.n_local;
  const int ia = a.i0 + (ona ? ila : 0), ib = a.i0 + (onb ? ilb : 0);
  const float *const xr = a.x_rows;  // (uniform)
  for (int j = tid; j < N; j += 1024) {
    xcol[j] = xr ? xr[(size_t)j * D + g] : a.thetaT[(size_t)g * N + j];
    scol[j] = a.score[(size_t)j * D + g];
  }
  const float h = a.h[g];
  const float ce = -1.44269504088896340736f / h;
  const v2f ce2 = {ce, ce};
  wg_sync();
  const v2f xi = {xcol[ia], xcol[ib]};
  v2f g1 = {0.f, 0.f}, g2 = {0.f, 0.f};
  {
    const int per = (N + 15) >> 4, j0 = js * per, j1 = min(N, j0 + per);
#pragma unroll 8
    for (int j = j0; j < j1; ++j) {
      const float xc = xcol[j], sc = scol[j];
      const v2f df = xi - v2f{xc, xc};
      const v2f arg = (df * df) * ce2;
      const v2f k = {__builtin_amdgcn_exp2f(arg.x), __builtin_amdgcn_exp2f(arg.y)};
      g1 = __builtin_elementwise_fma(k, v2f{sc, sc}, g1);
      g2 = __builtin_elementwise_fma(k, df, g2);
    }
  }
  part[js][ii][0] = g1.x;
  part[js][ii][1] = g2.x;
  part[js][ii + 64][0] = g1.y;
  part[js][ii + 64][1] = g2.y;
  wg_sync();
  if (tid < K2_P3_Q) {
    const int q = tid, il = blockIdx.x * K2_P3_Q + q;
    if (il < a.n_local) {
      const int i = a.i0 + il;
      float s1 = part[0][q][0], s2 = part[0][q][1];
      for (int r = 1; r < 16; ++r) {
        s1 += part[r][q][0];
        s2 += part[r][q][1];
      }
      const size_t o = (size_t)i * D + g;
      const float phi = s1 / (float)N + ((s2 * 2.0f) / h) / (float)N;
      a.phi[o] = phi;
      if (a.apply) {  // (update_from_phi_kernel's element)
        float th = a.theta_rw[o];
        const float gr = -phi;
        if (a.optimizer == DUST_OPT_SGD) {
          th = fmaf(-a.lr, gr, th);
        } else {
          float m = a.adam_m[o], v = a.adam_v[o];
          th = adam_step(th, gr, m, v, a.lr, a.beta1, a.beta2, a.eps, (float)a.ctr[2]);
          a.adam_m[o] = m;
          a.adam_v[o] = v;
        }
        (a.theta_out ? a.theta_out : a.theta_rw)[o] = th;
        if (a.thetaT_out) a.thetaT_out[(size_t)g * N + i] = th;
      }
    }
  }
  if (a.apply && blockIdx.x == 0 && blockIdx.y == 0) {
    for (int t = tid; t < a.fused_tiles; t += 1024) a.fused_cnt[t * 32] = 0u;
    if (tid == 0) a.ctr[1] += 1u;
  }
}

__global__ void k2_bandwidth_fixed_kernel(float *h, int G, float v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < G) h[i] = v;
}

// the two halves of the K2 step: bandwidths (read theta only) and phi (reads the score as well)
static inline int launch_k2_bandwidth(hipStream_t stream, const K2Args &a) {
  const int G = a.shared ? a.H : a.D;
  if (a.da > 2) return DUST_ERR_UNSUPPORTED;  // models on the path have d_a <= 2
  if (a.fixed_h > 0.f) {
    k2_bandwidth_fixed_kernel<<<(G + 255) / 256, 256, 0, stream>>>(a.h, G, a.fixed_h);
    return hipGetLastError() != hipSuccess ? DUST_ERR_HIP : DUST_OK;
  }
  if (a.shared) {
    k2_bandwidth_pairs_kernel<<<G, 1024, (size_t)a.da * a.N * sizeof(float), stream>>>(a);
  } else {
    int np = 1;
    while (np < a.N) np <<= 1;
    static const bool alone256 = getenv("DUST_K2_FORM") && atoi(getenv("DUST_K2_FORM")) == 3;
    if (a.N <= 1024 && alone256) k2_bandwidth256_kernel<<<G, 256, (size_t)K2_BW256_LDS * sizeof(float), stream>>>(a);
    else if (a.N <= 1024) k2_bandwidth_sorted_kernel<<<G, 1024, (size_t)2 * np * sizeof(float), stream>>>(a, np);  // (two buffers: the sort's LDS stages alternate)
    else k2_bandwidth_sorted_big_kernel<<<G, 1024, (size_t)np * sizeof(float), stream>>>(a, np);
  }
  return hipGetLastError() != hipSuccess ? DUST_ERR_HIP : DUST_OK;
}
static inline int launch_k2_phi(hipStream_t stream, const K2Args &a) {
  const int G = a.shared ? a.H : a.D;
  if (a.da > 2) return DUST_ERR_UNSUPPORTED;
  dim3 grid((a.n_local + 63) / 64, G);
  if (a.shared && a.da == 2) {
    k2_phi_kernel<2><<<grid, 256, 0, stream>>>(a);
  } else {
    static const bool one_q = getenv("DUST_K2_PHI1") != nullptr;  // development switch: one query per lane (k2_phi_kernel<1>)
    static const bool two_q = getenv("DUST_K2_PHI2") != nullptr;  // development switch: the 256-lane launch at every size
    if (one_q && !a.x_rows) k2_phi_kernel<1><<<grid, 256, 0, stream>>>(a);
    else if (a.N <= K2_P3_N && !two_q) k2_phi3_kernel<<<dim3((a.n_local + K2_P3_Q - 1) / K2_P3_Q, G), 1024, 0, stream>>>(a);
    else k2_phi2_kernel<<<dim3((a.n_local + 127) / 128, G), 256, 0, stream>>>(a);
  }
  return hipGetLastError() != hipSuccess ? DUST_ERR_HIP : DUST_OK;
}
static inline int launch_k2(hipStream_t stream, const K2Args &a) {
  const int s = launch_k2_bandwidth(stream, a);
  return s != DUST_OK ? s : launch_k2_phi(stream, a);
}

}  // namespace dust
