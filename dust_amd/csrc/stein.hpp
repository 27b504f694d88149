// stein.hpp - pairwise (N x N x D) passes: GMM prior score / log-density, Stein kernel Gram + phi, optimiser update.
//
// Replaces (reference file:line): the prior half of SVMPC.phi svmpc.py:38-41 (autograd through
// MixtureSameFamily.log_prob -> closed form sum_k r_ik (mu_k - x_i)/sigma^2), prior.log_prob in SVMPC.get_weights
// svmpc.py:128-140, the K1 kernel branch svmpc.py:76-83 (gpytorch RBFKernel semantics, lengthscale ln 2), the new IMQ
// kernel, and SVMPC.step's optimiser update svmpc.py:87-95.
//
// Mapping (MI355X): a workgroup owns TI query particles i and streams all N "key" particles j in chunks of JC:
//   pass A  lane = j : d2[t][j] = sum_d ((x_i[t][d] - YT[d][j]) / s_d)^2 from the TRANSPOSED copy YT[D][N] (coalesced
//           256-B wave loads; x_i broadcast from LDS), kernel value / softmax logit into LDS kv[TI][JC];
//   pass B  lane = (d, q): acc[t] += kv[t][j] * V[j][d] from the ROW-MAJOR copy V[N][D] (contiguous D-float rows),
//           j interleaved over the Q = blockDim/D lane groups so LDS reads of kv broadcast / stay conflict free.
// Nothing N x N is ever written to HBM (the reference materialises [N,N,H,da]).  Differences (x_i - x_j) are formed
// before multiplying by the kernel value, so collapsed particle sets do not cancel catastrophically.
#pragma once
#include "common.hpp"

namespace dust {

enum { PAIR_PRIOR = 0, PAIR_K1 = 1, PAIR_IMQ = 2 };

struct PairArgs {
  int N, D, da, H;
  int i0, n_local;     // query rows [i0, i0 + n_local)
  int JC;              // j-chunk held in LDS
  const float *X;      // [N][D] queries (theta)
  const float *YT;     // [D][N] keys, transposed (mu^T for the prior, theta^T for Stein)
  const float *Y;      // [N][D] keys, row-major
  const float *V;      // [N][D] second value array (score) for Stein; unused for the prior
  const float *logmix; // [N] prior mixture log-weights
  float inv_s[4];      // 1/sigma_p[d % da]  (prior)  or 1/ell (Stein)
  float inv_s2[4];
  float log_norm;      // -H*sum(log sigma_p) - D/2 log(2 pi)
  float inv_n;         // 1/N
  float *out;          // prior: grad_pri [N][D] ; Stein: phi [N][D]
  const float *add;    // prior: grad_lik to add -> score written to out2
  float *out2;         // prior: score [N][D]
  float *logp;         // prior: log p(x_i) [N] (nullptr to skip)
};

template <int MODE, int TI>
__global__ __launch_bounds__(256) void pairwise_kernel(const PairArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, nt = blockDim.x;
  const int N = a.N, D = a.D, da = a.da, JC = a.JC;
  float *xi = lds;                 // [TI][D]
  float *kv = xi + TI * D;         // [TI][JC]
  float *red = kv + TI * JC;       // [32]
  float *mrun = red + 32;          // [TI] running max (prior)
  float *part = mrun + TI;         // [TI][nt] + [TI][nt] partial sums
  const int ib = a.i0 + blockIdx.x * TI;

  for (int idx = tid; idx < TI * D; idx += nt) {
    const int t = idx / D, d = idx - t * D;
    const int i = ib + t;
    xi[idx] = (i < a.i0 + a.n_local) ? a.X[(size_t)i * D + d] : 0.f;
  }
  if (tid < TI) mrun[tid] = -INFINITY;
  __syncthreads();

  const int Q = nt / D > 0 ? nt / D : 1;
  const int d = tid % D, q = tid / D;
  const bool active = q < Q;
  float accA[TI], accB[TI], accC[TI];  // prior: A = sum p (mu - x), C = sum p ; Stein: A = sum k s, B = sum k' (x_i - x_j), C unused
#pragma unroll
  for (int t = 0; t < TI; ++t) accA[t] = accB[t] = accC[t] = 0.f;
  float xid[TI];
#pragma unroll
  for (int t = 0; t < TI; ++t) xid[t] = active ? xi[t * D + d] : 0.f;

  for (int j0 = 0; j0 < N; j0 += JC) {
    const int jc = min(JC, N - j0);
    // ---- pass A: kernel values / logits for this chunk ----
    float lmax[TI];
#pragma unroll
    for (int t = 0; t < TI; ++t) lmax[t] = -INFINITY;
    for (int jj = tid; jj < jc; jj += nt) {
      const int j = j0 + jj;
      float d2[TI];
#pragma unroll
      for (int t = 0; t < TI; ++t) d2[t] = 0.f;
      for (int dd = 0; dd < D; ++dd) {
        const float y = a.YT[(size_t)dd * N + j];
        const float is = a.inv_s[dd % da];
#pragma unroll
        for (int t = 0; t < TI; ++t) {
          const float z = (xi[t * D + dd] - y) * is;
          d2[t] = fmaf(z, z, d2[t]);
        }
      }
#pragma unroll
      for (int t = 0; t < TI; ++t) {
        float v;
        if (MODE == PAIR_PRIOR) {
          v = a.logmix[j] - 0.5f * d2[t];
          lmax[t] = fmaxf(lmax[t], v);
        } else if (MODE == PAIR_K1) {
          v = expf(-0.5f * d2[t]);
        } else {
          v = d2[t];  // IMQ: keep the scaled squared distance; both k and k' are formed in pass B
        }
        kv[t * JC + jj] = v;
      }
    }
    if (MODE == PAIR_PRIOR) {
      // online softmax across chunks: rescale the running sums when the max moves
#pragma unroll
      for (int t = 0; t < TI; ++t) {
        const float cm = block_reduce<RED_MAX>(lmax[t], red);
        const float mo = mrun[t];
        const float mn = fmaxf(mo, cm);
        const float sc = (mo == -INFINITY) ? 0.f : expf(mo - mn);
        accA[t] *= sc;
        accC[t] *= sc;
        __syncthreads();
        if (tid == 0) mrun[t] = mn;
      }
      __syncthreads();
      for (int idx = tid; idx < TI * jc; idx += nt) {
        const int t = idx / jc, jj = idx - t * jc;
        kv[t * JC + jj] = expf(kv[t * JC + jj] - mrun[t]);
      }
    }
    __syncthreads();
    // ---- pass B: accumulate over the chunk ----
    if (active) {
      for (int jj = q; jj < jc; jj += Q) {
        const int j = j0 + jj;
        const float y = a.Y[(size_t)j * D + d];
        if (MODE == PAIR_PRIOR) {
#pragma unroll
          for (int t = 0; t < TI; ++t) {
            const float p = kv[t * JC + jj];
            accA[t] = fmaf(p, y - xid[t], accA[t]);
            accC[t] += p;
          }
        } else {
          const float sv = a.V[(size_t)j * D + d];
#pragma unroll
          for (int t = 0; t < TI; ++t) {
            float k, kp;
            if (MODE == PAIR_K1) {
              k = kv[t * JC + jj];
              kp = -k;  // d k / d x_i = -k (x_i - x_j)/ell^2
            } else {
              const float base = 1.0f + kv[t * JC + jj];
              k = rsqrtf(base);
              kp = -k / base;
            }
            accA[t] = fmaf(k, sv, accA[t]);
            accB[t] = fmaf(kp, xid[t] - y, accB[t]);
          }
        }
      }
    }
    __syncthreads();
  }

  // ---- cross-group reduction and epilogue ----
#pragma unroll
  for (int t = 0; t < TI; ++t) {
    part[t * nt + tid] = accA[t];
    part[(TI + t) * nt + tid] = (MODE == PAIR_PRIOR) ? accC[t] : accB[t];
  }
  __syncthreads();
  for (int idx = tid; idx < TI * D; idx += nt) {
    const int t = idx / D, dd = idx - t * D;
    const int i = ib + t;
    if (i >= a.i0 + a.n_local) continue;
    float sa = 0.f, sb = 0.f;
    for (int qq = 0; qq < Q; ++qq) {
      sa += part[t * nt + qq * D + dd];
      sb += part[(TI + t) * nt + qq * D + dd];
    }
    if (MODE == PAIR_PRIOR) {
      const float gp = (sa / sb) * a.inv_s2[dd % da];
      if (a.out) a.out[(size_t)i * D + dd] = gp;
      if (a.out2) a.out2[(size_t)i * D + dd] = a.add[(size_t)i * D + dd] + gp;
      if (a.logp && dd == 0) a.logp[i] = (mrun[t] + logf(sb)) + a.log_norm;
    } else {
      a.out[(size_t)i * D + dd] = sb * a.inv_s2[0] + sa * a.inv_n;  // grad_k (not /N) + K score / N  (svmpc.py:83)
    }
  }
}

static inline size_t pairwise_lds_bytes(int TI, int D, int JC, int nt) {
  return sizeof(float) * ((size_t)TI * D + (size_t)TI * JC + 32 + TI + 2 * (size_t)TI * nt);
}

// ---------------------------------------------------------------------------------------------------------------
// Optimiser update (svmpc.py:87-95: theta.grad = -phi; optimizer.step()).  Writes the row-major and the transposed copy.
struct UpdateArgs {
  int N, D, i0, n_local;
  int optimizer;
  float lr, beta1, beta2, eps;
  int step;  // Adam step count (1-based)
  const float *phi;
  float *theta;   // [N][D] in place (each element touched by exactly one thread)
  float *thetaT;  // [D][N]
  float *adam_m, *adam_v;
};

__global__ void update_kernel(const UpdateArgs a) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= a.n_local * a.D) return;
  const int il = idx / a.D, d = idx - il * a.D;
  const int i = a.i0 + il;
  const size_t o = (size_t)i * a.D + d;
  float th = a.theta[o];
  const float g = -a.phi[o];
  if (a.optimizer == DUST_OPT_SGD) {
    th = fmaf(-a.lr, g, th);  // torch SGD: p.add_(grad, alpha=-lr), a vectorised fmadd
  } else {  // torch.optim.Adam (no weight decay, no amsgrad)
    float m = a.adam_m[o], v = a.adam_v[o];
    m = fmaf(a.beta1, m, (1.f - a.beta1) * g);
    v = fmaf(a.beta2, v, (1.f - a.beta2) * g * g);
    a.adam_m[o] = m;
    a.adam_v[o] = v;
    const float bc1 = 1.f - powf(a.beta1, (float)a.step), bc2 = 1.f - powf(a.beta2, (float)a.step);
    const float denom = sqrtf(v) / sqrtf(bc2) + a.eps;
    th = th - (a.lr / bc1) * (m / denom);
  }
  a.theta[o] = th;
  a.thetaT[(size_t)d * a.N + i] = th;
}

// row-major [N][D] -> transposed [D][N]
__global__ void transpose_kernel(const float *src, float *dst, int N, int D) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * D) return;
  const int i = idx / D, d = idx - i * D;
  dst[(size_t)d * N + i] = src[idx];
}

// [N][S] -> [S][N] (and back) for the host-facing cost / weight layouts
__global__ void transpose2_kernel(const float *src, float *dst, int R, int C) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= R * C) return;
  const int r = idx / C, c = idx - r * C;
  dst[(size_t)c * R + r] = src[idx];
}

}  // namespace dust
