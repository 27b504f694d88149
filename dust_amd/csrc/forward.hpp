// forward.hpp - tick epilogue: particle weights, argmax, roll, prior refresh; a_mix; MultiDISCO.step.
//
// Replaces (reference file:line): SVMPC.get_weights svmpc.py:128-140, SVMPC.forward svmpc.py:172-200, SVMPC.roll
// svmpc.py:142-158, SVMPC.update_prior svmpc.py:160-170 (+ get_gmm svgd.py:84-89 and the Categorical/
// MixtureSameFamily log-weight construction in torch.distributions), a_mix disco.py:393, MultiDISCO.step disco.py:396-417.
#pragma once
#include "common.hpp"
#include "stein.hpp"

namespace dust {

struct FinalizeArgs {
  int N, D;
  const float *logl, *logp;  // [N]
  float *lw;                 // [N] log_l + log_p (gathered across shards by the caller when sharded)
  float *pw;                 // [N] out: particle weights
  int *istar;                // out
  const float *theta;        // [N][D]
  float *a_seq_out;          // [D] out: theta[i*]
  float *logmix;             // [N] out: new prior log mixture weights
  float *mixw;               // [N] out: new prior mixture weights as given to get_gmm (ones or p)
  int weighted_prior;
  int keep_prior;            // SVMPC.get_weights alone (svmpc.py:128-140): weights / argmax only, the prior mixture stays
  int merge_logp;            // unsharded path: combine the prior-pass partials here (log p, then lw = logl + logp)
  float *logp_out;           // [N]
  PriorMerge pm;
};

// block-wide reduction of TWO values at once (one barrier pair instead of two)
__device__ __forceinline__ void block_reduce2(float &mx, float &sm, float *scratch /* >= 32 */) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  mx = wave_max(mx);
  sm = wave_sum(sm);
  wg_sync();
  if (lane == 0) {
    scratch[wid] = mx;
    scratch[16 + wid] = sm;
  }
  wg_sync();
  mx = scratch[0];
  sm = scratch[16];
  for (int w = 1; w < nw; ++w) {
    mx = fmaxf(mx, scratch[w]);
    sm += scratch[16 + w];
  }
}

// single workgroup: softmax over N, first-index argmax, a_seq, new mixture log-weights.  Each lane keeps its particles'
// values in registers between the passes (N <= 8 * 1024 on that path), so global memory is read once and written once.
__device__ __forceinline__ void finalize_body(const FinalizeArgs &a) {
  __shared__ float red[32];
  __shared__ int redi[32];
  const int tid = threadIdx.x, nt = blockDim.x;
  constexpr int R = 16;  // N <= 16384 particles (validated at create)
  float lwr[R];  // requires N <= R * blockDim (the host checks)
  // pass 1: log-weights (merging the prior-pass partials when unsharded) and their max
  float m = -INFINITY;
  _Pragma("unroll") for (int r = 0; r < R; ++r) { const int i = tid + r * nt; if (i >= a.N) continue;
    float lw;
    if (a.merge_logp) {  // prior.log_prob(theta) from the slice partials (svmpc.py:137), then log_w = log_l + log_p (:138)
      const float ll = a.logl[i];  // in flight together with the partials
      float pmx, pl;
      prior_merge_row(a.pm, i, &pmx, &pl);
      const float lp = (pmx + logf(pl)) + a.pm.log_norm;
      a.logp_out[i] = lp;
      lw = ll + lp;
      a.lw[i] = lw;
    } else {
      lw = a.lw[i];
    }
    lwr[r] = lw;
    m = fmaxf(m, lw);
  }
  m = block_reduce<RED_MAX>(m, red);
  float z = 0.f;
  _Pragma("unroll") for (int r = 0; r < R; ++r) { const int i = tid + r * nt; if (i < a.N) z += expf(lwr[r] - m); }
  z = block_reduce<RED_SUM>(z, red);
  const float lz = m + logf(z);
  // p = exp(log_w - logsumexp(log_w)) ; argmax = first index of the maximum (torch.argmax on CPU)
  float best = -INFINITY;
  int bi = 0x7fffffff;
  float psum = 0.f;
  _Pragma("unroll") for (int r = 0; r < R; ++r) { const int i = tid + r * nt; if (i >= a.N) continue;
    const float p = expf(lwr[r] - lz);
    a.pw[i] = p;
    lwr[r] = p;
    psum += p;
    if (p > best) {
      best = p;
      bi = i;
    }
  }
  // block argmax with lowest-index tie break
  const int lane = tid & 63, wid = tid >> 6, nw = (nt + 63) >> 6;
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ob > best || (ob == best && oi < bi)) {
      best = ob;
      bi = oi;
    }
  }
  wg_sync();
  if (lane == 0) {
    red[wid] = best;
    redi[wid] = bi;
  }
  wg_sync();
  best = lane < nw ? red[lane] : -INFINITY;  // second level in registers (lane w holds wave w's candidate)
  bi = lane < nw ? redi[lane] : 0x7fffffff;
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ob > best || (ob == best && oi < bi)) {
      best = ob;
      bi = oi;
    }
  }
  if (tid == 0) *a.istar = bi;
  for (int d = tid; d < a.D; d += nt) a.a_seq_out[d] = a.theta[(size_t)bi * a.D + d];
  if (a.keep_prior) return;
  // new prior mixture: Categorical(probs = w / sum w) -> logits = log(clamp(probs, eps, 1 - eps)) -> log_softmax
  if (!a.weighted_prior) {  // uniform: every logit is log(1/N) and the log_softmax of a constant vector is -log N exactly
    const float l = logf(fminf(fmaxf(1.0f / (float)a.N, 1.1920929e-07f), 1.0f - 1.1920929e-07f));
    const float lzz = l + logf((float)a.N);  // sum_i exp(l - l) = N exactly (N <= 16384): no reduction needed
    for (int i = tid; i < a.N; i += nt) {
      a.mixw[i] = 1.0f;
      a.logmix[i] = l - lzz;
    }
    return;
  }
  psum = block_reduce<RED_SUM>(psum, red);
  float lm = -INFINITY;
  _Pragma("unroll") for (int r = 0; r < R; ++r) { const int i = tid + r * nt; if (i >= a.N) continue;
    const float w = lwr[r];
    a.mixw[i] = w;
    float p = w / psum;
    p = fminf(fmaxf(p, 1.1920929e-07f), 1.0f - 1.1920929e-07f);
    const float l = logf(p);
    lwr[r] = l;
    lm = fmaxf(lm, l);
  }
  lm = block_reduce<RED_MAX>(lm, red);
  float zs = 0.f;
  _Pragma("unroll") for (int r = 0; r < R; ++r) { const int i = tid + r * nt; if (i < a.N) zs += expf(lwr[r] - lm); }
  zs = block_reduce<RED_SUM>(zs, red);
  const float lzz = lm + logf(zs);
  _Pragma("unroll") for (int r = 0; r < R; ++r) { const int i = tid + r * nt; if (i < a.N) a.logmix[i] = lwr[r] - lzz; }
}

__global__ __launch_bounds__(1024) void finalize_kernel(const FinalizeArgs a) { finalize_body(a); }

// mixture log-weights from user-supplied weights (set_prior): same construction as above
__global__ __launch_bounds__(1024) void logmix_kernel(const float *w, float *logmix, int N) {
  __shared__ float red[32];
  const int tid = threadIdx.x, nt = blockDim.x;
  float s = 0.f;
  for (int i = tid; i < N; i += nt) s += w[i];
  s = block_reduce<RED_SUM>(s, red);
  float lm = -INFINITY;
  for (int i = tid; i < N; i += nt) {
    float p = w[i] / s;
    p = fminf(fmaxf(p, 1.1920929e-07f), 1.0f - 1.1920929e-07f);
    const float l = logf(p);
    logmix[i] = l;
    lm = fmaxf(lm, l);
  }
  lm = block_reduce<RED_MAX>(lm, red);
  float zs = 0.f;
  for (int i = tid; i < N; i += nt) zs += expf(logmix[i] - lm);
  zs = block_reduce<RED_SUM>(zs, red);
  const float lzz = lm + logf(zs);
  for (int i = tid; i < N; i += nt) logmix[i] = logmix[i] - lzz;
}

// SVMPC.roll svmpc.py:142-158 (steps = -1): shift left along H; last row per strategy.  One 128-lane group per particle,
// lane = element: every element is read into a register, the group syncs, then the shifted value is written.
struct RollArgs {
  const float *theta;  // current particles
  float *theta_dst;    // where the rolled particles go (the home buffer of the ping-pong; may equal theta)
  int N, H, da, strategy, i0, n_local;
  uint32_t *ctr;
  unsigned int *rearm;  // hand-off counters of the one-launch SVGD iteration (both sets; one 128-byte line each) or nullptr:
  int rearm_lines;      // zeroed here, so that a tick - and a replayed graph - starts with both sets clean
  unsigned int *hs;     // both score hand-off buffers of the one-launch iteration (stein.hpp SCORE_SENTINEL) or nullptr:
  int hs_n;             // filled with the sentinel again
  // SVMPC.roll builds a NEW parameter tensor (theta.roll(...), svmpc.py:142-158) and stores it into the optimiser's param group:
  // torch keys optimiser state by tensor object, so Adam's exp_avg / exp_avg_sq / step restart at zero after every forward()
  float *adam_m, *adam_v;  // [N][D] or nullptr (SGD)
  int steps;               // theta.roll(steps, dims=-2): circular shift along the horizon (svmpc.py:144); -1 in SVMPC.forward's default
  const float *last_row;   // [N][da] strategy "resample": the last action of a fresh prior sample per particle (svmpc.py:148-150)
};

__global__ __launch_bounds__(128) void roll_kernel(const RollArgs a) {
  __shared__ float red[32];
  if (blockIdx.x == 0 && threadIdx.x == 0) {  // next tick: new Philox sub-stream, fresh optimiser step count
    a.ctr[0] += 1u;
    a.ctr[1] = 0u;
    a.ctr[2] = 0u;
  }
  if (blockIdx.x == 0)
    for (int t = threadIdx.x; t < a.rearm_lines; t += blockDim.x) a.rearm[t * 32] = 0u;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < a.hs_n; e += gridDim.x * blockDim.x) a.hs[e] = SCORE_SENTINEL;
  const int i = a.i0 + blockIdx.x;
  const int D = a.H * a.da, j = threadIdx.x, da = a.da;
  const float *th = a.theta + (size_t)i * D;
  const float own = j < D ? th[j] : 0.f;
  // torch.roll is circular: row t of the result is row (t - steps) mod H; the strategy then overwrites the LAST row only
  const int H = a.H, t = j / da, cdim = j - t * da;
  const int sh = ((-a.steps) % H + H) % H;  // rows move up by sh (steps = -1: sh = 1)
  float out = own;
  if (j < D) {
    int src = (t + sh) % H;
    if (t == H - 1 && a.strategy == DUST_ROLL_REPEAT && H >= 2) src = (H - 2 + sh) % H;  // theta[-1] = theta[-2] of the ROLLED tensor
    out = th[src * da + cdim];
    if (t == H - 1 && a.strategy == DUST_ROLL_RESAMPLE) out = a.last_row[(size_t)i * da + cdim];
  }
  if (a.strategy == DUST_ROLL_MEAN) {  // mean over the horizon of each control dimension (svmpc.py:151-153)
    for (int c = 0; c < da; ++c) {
      const float s = block_reduce<RED_SUM>((j < D && j % da == c) ? own : 0.f, red);
      if (j + da >= D && j < D && j % da == c) out = s / (float)a.H;
    }
  }
  wg_sync();
  if (j < D) {
    a.theta_dst[(size_t)i * D + j] = out;
    if (a.adam_m) {
      a.adam_m[(size_t)i * D + j] = 0.f;
      a.adam_v[(size_t)i * D + j] = 0.f;
    }
  }
}

// finalize + roll in ONE launch (1024-lane workgroups: workgroup 0 finalizes, every other one rolls 8 particles).  Only
// when the roll is out of place ("repeat" strategy, theta != theta_dst): finalize gathers a_seq = theta[i*] from the
// buffer the roll only reads, so the two are independent and the roll hides under the single-workgroup finalize.
__global__ __launch_bounds__(1024) void finalize_roll_kernel(const FinalizeArgs f, const RollArgs a) {
  if (blockIdx.x == 0) {
    finalize_body(f);
    if (threadIdx.x == 0) {
      a.ctr[0] += 1u;
      a.ctr[1] = 0u;
      a.ctr[2] = 0u;
    }
    return;
  }
  if (blockIdx.x == 1)
    for (int t = threadIdx.x; t < a.rearm_lines; t += 1024) a.rearm[t * 32] = 0u;
  for (int e = ((int)blockIdx.x - 1) * 1024 + (int)threadIdx.x; e < a.hs_n; e += ((int)gridDim.x - 1) * 1024) a.hs[e] = SCORE_SENTINEL;
  const int il = ((int)blockIdx.x - 1) * 8 + ((int)threadIdx.x >> 7), j = threadIdx.x & 127;
  if (il >= a.n_local) return;
  const int i = a.i0 + il, D = a.H * a.da;
  const float *th = a.theta + (size_t)i * D;
  if (j < D) {
    a.theta_dst[(size_t)i * D + j] = (j + a.da < D) ? th[j + a.da] : th[j];
    if (a.adam_m) {
      a.adam_m[(size_t)i * D + j] = 0.f;
      a.adam_v[(size_t)i * D + j] = 0.f;
    }
  }
}

// a_mix = softmax_n(eta) disco.py:393 (single workgroup)
__global__ __launch_bounds__(1024) void amix_kernel(const float *eta, float *a_mix, int N) {
  __shared__ float red[32];
  const int tid = threadIdx.x, nt = blockDim.x;
  float m = -INFINITY;
  for (int i = tid; i < N; i += nt) m = fmaxf(m, eta[i]);
  m = block_reduce<RED_MAX>(m, red);
  float z = 0.f;
  for (int i = tid; i < N; i += nt) z += expf(eta[i] - m);
  z = block_reduce<RED_SUM>(z, red);
  const float lz = m + logf(z);
  for (int i = tid; i < N; i += nt) a_mix[i] = expf(eta[i] - lz);
}

// MultiDISCO.step disco.py:396-417 (single workgroup; N*D is small)
struct StepArgs {
  int N, H, da, strategy, steps;
  float min_a[4], max_a[4];
  const float *a_mix;
  const float *ext;  // [D] for strategy external
  float *a_mat;      // [N][D]
  float *a_seq;      // [D]
  float *next;       // [steps][da]
};
__global__ __launch_bounds__(1024) void disco_step_kernel(const StepArgs a) {
  __shared__ float red[32];
  __shared__ int redi[32];
  __shared__ int s_best;
  extern __shared__ float seq[];  // [D]
  const int tid = threadIdx.x, nt = blockDim.x;
  const int D = a.H * a.da;
  if (a.strategy == DUST_STEP_ARGMAX) {
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = tid; i < a.N; i += nt)
      if (a.a_mix[i] > best) {
        best = a.a_mix[i];
        bi = i;
      }
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ob > best || (ob == best && oi < bi)) {
        best = ob;
        bi = oi;
      }
    }
    if ((tid & 63) == 0) {
      red[tid >> 6] = best;
      redi[tid >> 6] = bi;
    }
    wg_sync();
    if (tid == 0) {
      for (int w = 1; w < (nt + 63) / 64; ++w)
        if (red[w] > best || (red[w] == best && redi[w] < bi)) {
          best = red[w];
          bi = redi[w];
        }
      s_best = bi;
    }
    wg_sync();
    for (int j = tid; j < D; j += nt) {
      const float v = clampf(a.a_mat[(size_t)s_best * D + j], a.min_a[j % a.da], a.max_a[j % a.da]);
      seq[j] = v;
      a.a_mat[(size_t)s_best * D + j] = v;  // quirk: a_mat[argmax] is a view, the in-place clamp_ reaches a_mat too
    }
  } else if (a.strategy == DUST_STEP_AVERAGE) {
    for (int j = tid; j < D; j += nt) {
      double acc = 0.0;
      for (int n = 0; n < a.N; ++n) acc += (double)a.a_mat[(size_t)n * D + j] * (double)a.a_mix[n];
      seq[j] = clampf((float)acc, a.min_a[j % a.da], a.max_a[j % a.da]);
    }
  } else {
    for (int j = tid; j < D; j += nt) seq[j] = clampf(a.ext[j], a.min_a[j % a.da], a.max_a[j % a.da]);
  }
  wg_sync();
  for (int j = tid; j < a.steps * a.da; j += nt) a.next[j] = seq[j];
  const int sh = a.steps * a.da;
  for (int j = tid; j < D; j += nt) a.a_seq[j] = (j + sh < D) ? seq[j + sh] : 0.f;
}

// a_mat.roll(-steps, dims=1) with zero fill (disco.py:415-416): one thread per (n, control dim), sequential in t
__global__ void amat_roll_kernel(float *a_mat, int N, int H, int da, int steps) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * da) return;
  const int n = idx / da, c = idx % da;
  float *row = a_mat + (size_t)n * H * da;
  for (int t = 0; t < H; ++t) row[t * da + c] = (t + steps < H) ? row[(t + steps) * da + c] : 0.f;
}

// advance the Philox position after a stand-alone sample (no optimiser step / roll follows)
__global__ void bump_iter_kernel(uint32_t *ctr) { ctr[1] += 1u; }

}  // namespace dust
