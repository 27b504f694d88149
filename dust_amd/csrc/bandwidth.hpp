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

namespace dust {

struct K2Args {
  int N, H, da, D;
  int shared;  // 1: one kernel per timestep (indep_controls=False)
  int i0, n_local;
  float bw_scale;
  const float *theta;   // [N][D]
  const float *thetaT;  // [D][N]
  const float *score;   // [N][D]
  float *h;             // [G] bandwidths out (G = D, or H when shared)
  float *phi;           // [N][D]
};

__device__ __forceinline__ unsigned long long block_sum_u64(unsigned long long v, unsigned long long *scratch) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if (lane == 0) scratch[wid] = v;
  __syncthreads();
  unsigned long long r = 0;
  for (int w = 0; w < nw; ++w) r += scratch[w];
  return r;
}

// one workgroup per independent scalar dimension c
__global__ __launch_bounds__(1024) void k2_bandwidth_sorted_kernel(const K2Args a, int npow2) {
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [npow2]
  __shared__ unsigned long long red64[16];
  const int tid = threadIdx.x, nt = blockDim.x, N = a.N, c = blockIdx.x;
  for (int i = tid; i < npow2; i += nt) xs[i] = i < N ? a.thetaT[(size_t)c * N + i] : INFINITY;
  __syncthreads();
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
      __syncthreads();
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
    a.h[c] = fmaxf(h, 1e-5f);
  }
}

// K2_SHARED: one workgroup per timestep, O(N^2) count per bisection step
__global__ __launch_bounds__(1024) void k2_bandwidth_pairs_kernel(const K2Args a) {
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [da][N]
  __shared__ unsigned long long red64[16];
  __shared__ float redf[32];
  const int tid = threadIdx.x, nt = blockDim.x, N = a.N, da = a.da, g = blockIdx.x;
  for (int idx = tid; idx < da * N; idx += nt) xs[idx] = a.thetaT[(size_t)(g * da) * N + idx];
  __syncthreads();
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
    a.h[g] = fmaxf(h, 1e-5f);
  }
}

// phi_ic = mean_j K^c_ij score_jc + mean_j K^c_ij (x_ic - x_jc) 2/h_c   (svmpc.py:69-73, base_kernels.py:100-101)
// grid = (ceil(n_local/256), G); lane = particle i; the j loop reads wave-uniform addresses (scalar loads).
__global__ __launch_bounds__(256) void k2_phi_kernel(const K2Args a) {
  const int N = a.N, D = a.D, da = a.da;
  const int g = blockIdx.y;
  const int gd = a.shared ? da : 1;
  const int c0 = g * gd;
  const int il = blockIdx.x * blockDim.x + threadIdx.x;
  if (il >= a.n_local) return;
  const int i = a.i0 + il;
  const float h = a.h[g];
  float xi[4], g1[4], g2[4];
  for (int q = 0; q < gd; ++q) {
    xi[q] = a.thetaT[(size_t)(c0 + q) * N + i];
    g1[q] = g2[q] = 0.f;
  }
  for (int j = 0; j < N; ++j) {
    float d2 = 0.f, df[4];
    for (int q = 0; q < gd; ++q) {
      df[q] = xi[q] - a.thetaT[(size_t)(c0 + q) * N + j];
      d2 = fmaf(df[q], df[q], d2);
    }
    const float k = expf(-d2 / h);
    for (int q = 0; q < gd; ++q) {
      g1[q] = fmaf(k, a.score[(size_t)j * D + c0 + q], g1[q]);
      g2[q] += ((k * df[q]) * 2.0f) / h;
    }
  }
  for (int q = 0; q < gd; ++q) a.phi[(size_t)i * D + c0 + q] = g1[q] / (float)N + g2[q] / (float)N;
}

static inline int launch_k2(hipStream_t stream, const K2Args &a) {
  const int G = a.shared ? a.H : a.D;
  if (a.shared) {
    k2_bandwidth_pairs_kernel<<<G, 1024, (size_t)a.da * a.N * sizeof(float), stream>>>(a);
  } else {
    int np = 1;
    while (np < a.N) np <<= 1;
    k2_bandwidth_sorted_kernel<<<G, 1024, (size_t)np * sizeof(float), stream>>>(a, np);
  }
  if (hipGetLastError() != hipSuccess) return DUST_ERR_HIP;
  dim3 grid((a.n_local + 255) / 256, G);
  k2_phi_kernel<<<grid, 256, 0, stream>>>(a);
  if (hipGetLastError() != hipSuccess) return DUST_ERR_HIP;
  return DUST_OK;
}

}  // namespace dust
