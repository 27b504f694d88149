// mpf.hpp - dynamics-parameter SVGD filter ("MPF"), included at the end of dust_amd.hip.
//
// Replaces (reference file:line): MPF.phi / step / optimize / update_prior mpf.py:26-86, GaussianLikelihood.sample /
// log_prob / condition likelihoods.py:30-64, default_kernel + squared_distance svgd.py:28-39, 92-99, and the autograd
// calls at mpf.py:45 and mpf.py:50 (closed forms: GMM responsibilities; J^T (y - f(x)) / sigma_o^2 with the analytic
// Jacobian of one model step w.r.t. the uncertain parameters).
//
// The problem is tiny (M_p <= 1024 particles x P <= 4 parameters, n_steps ~ 20 dependent SVGD steps), i.e. pure launch
// latency on a GPU: ALL n_steps run inside ONE single-workgroup kernel with the particles, scores and squared norms in
// LDS; up to 1024 lanes = (particle, slice of the other particles); the only HBM traffic is the final particle
// write-back and the gradient norms.
//
// Reference quirk kept: MPF.update_prior hands `self.x` itself to MultivariateNormal(loc=...), which aliases the
// particle storage, and SGD updates x in place - so the prior means are always the CURRENT particles (mpf.py:26-38).
#pragma once

namespace dust {

struct MpfArgs {
  DevModel dm;
  int Mp, P, ds, da, n_steps, log_space, have_past;
  float prior_bwv[4];  // prior bandwidth per parameter dimension (equal after the first update_prior; MPF(bw=None) starts per-dimension)
  float bw, lr, obs_std;
  float past_obs[4], past_action[2], obs[4];
  // control-channel noise of the one-step prediction (Particle(deterministic=False), particle.py:145-148 reached through
  // likelihoods.py:30-46): `acts` there is the bare action vector, so ONE d_a-vector is drawn per phi() call - i.e. per SVGD step -
  // and shared by all filter particles.  act_seq[step][2] = fl(past_action + fl(dyn_std * z_step)), prepared by the host, or nullptr
  const float *act_seq;
  float *x;           // [Mp][P] in/out
  float *grad_norms;  // [n_steps] or nullptr
  float *phi_out;     // [Mp][P] or nullptr (phi of the first step, when n_steps == 0 semantics are wanted use n_steps=1, lr=0)
  // optimiser (SVGD.__init__ svgd.py:115: the class default is torch.optim.Adam; built once in MPF.__init__ mpf.py:24, so its
  // state persists across optimize() calls): DUST_OPT_SGD, or DUST_OPT_ADAM with moments [Mp][P] in / out and t0 steps taken so far
  int optimizer, t0;
  float beta1, beta2, eps;
  float *adam_m, *adam_v;
};

// d(next state)/d(params) of one model step, as autograd returns it through model.step (incl. clamp masks).
// J is [4][4] (state row, parameter column) and every index below is a compile-time constant or a select: a
// dynamically indexed local array would live in scratch memory.
__device__ __forceinline__ double sel4(const double v[4], int c) { return c == 0 ? v[0] : (c == 1 ? v[1] : (c == 2 ? v[2] : v[3])); }
__device__ __forceinline__ void add_col(double J[4][4], int row, int col, double v) {
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (k == row && c == col) J[k][c] += v;
}
template <int P>
__device__ __forceinline__ void step_jacobian(const DevModel &dm, const float *x, const float *a, const float prow[4], double J[4][4]) {
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int c = 0; c < 4; ++c) J[k][c] = 0.0;
  double pv[4] = {0, 0, 0, 0};
#pragma unroll
  for (int p = 0; p < P; ++p) pv[p] = dm.log_space ? exp((double)prow[p]) : (double)prow[p];
  if (dm.model == DUST_MODEL_PENDULUM) {
    const double g = dm.g.kind == DUST_PARAM_SAMPLED ? sel4(pv, dm.g.col) : dm.g.value;
    const double m = dm.mass.kind == DUST_PARAM_SAMPLED ? sel4(pv, dm.mass.col) : dm.mass.value;
    const double l = dm.length.kind == DUST_PARAM_SAMPLED ? sel4(pv, dm.length.col) : dm.length.value;
    const double dt = dm.dt;
    const double u = clampf(a[0], -dm.max_torque, dm.max_torque);
    const double s = sin((double)x[0] + M_PI);
    const double thd = (double)x[1] + dt * (-3.0 * g / (2.0 * l) * s + 3.0 / (m * l * l) * u);
    if (!(thd >= -dm.max_speed_pend && thd <= dm.max_speed_pend)) return;
    const double dg = dt * (-3.0 / (2.0 * l) * s), dmass = dt * (-3.0 / (m * m * l * l) * u);
    const double dl = dt * (3.0 * g / (2.0 * l * l) * s - 6.0 / (m * l * l * l) * u);
    if (dm.g.kind == DUST_PARAM_SAMPLED) {
      const double ch = dm.log_space ? sel4(pv, dm.g.col) : 1.0;
      add_col(J, 1, dm.g.col, dg * ch);
      add_col(J, 0, dm.g.col, dg * dt * ch);
    }
    if (dm.mass.kind == DUST_PARAM_SAMPLED) {
      const double ch = dm.log_space ? sel4(pv, dm.mass.col) : 1.0;
      add_col(J, 1, dm.mass.col, dmass * ch);
      add_col(J, 0, dm.mass.col, dmass * dt * ch);
    }
    if (dm.length.kind == DUST_PARAM_SAMPLED) {
      const double ch = dm.log_space ? sel4(pv, dm.length.col) : 1.0;
      add_col(J, 1, dm.length.col, dl * ch);
      add_col(J, 0, dm.length.col, dl * dt * ch);
    }
  } else {
    if (dm.mass.kind != DUST_PARAM_SAMPLED) return;
    const int col = dm.mass.col;
    const double m = sel4(pv, col), dt = dm.dt;
    double om = 1.0;
    if (dm.can_crash && dm.with_obstacle) om = 1.0 - (double)collision(dm, x[0], x[1]);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const double acc = (double)a[k] / m;
      const bool live_a = acc >= -dm.max_acc && acc <= dm.max_acc;
      const double accc = acc < -dm.max_acc ? -dm.max_acc : (acc > dm.max_acc ? dm.max_acc : acc);
      const double v = (double)x[2 + k] + accc * dt * om;
      const bool live_v = v >= -dm.max_speed && v <= dm.max_speed;
      if (live_a && live_v) add_col(J, 2 + k, col, dt * om * (-(double)a[k] / (m * m)) * (dm.log_space ? m : 1.0));
    }
  }
}

// Lane = (particle i, slice r of the other particles): R = blockDim / Mpad slices share the O(M_p) loops of a particle and
// their partial sums are combined in slice order through LDS (fixed order: reproducible).  No divisions or fp64
// transcendentals inside the O(M_p^2) loops: reciprocals are hoisted (fp64, error 1e-16), weights use expf.
// P is a template parameter: with a run-time P the per-lane arrays are indexed dynamically and live in scratch memory
// (measured: 3 us per inner-loop iteration instead of ~50 ns).
template <int P>
__global__ __launch_bounds__(1024) void mpf_optimize_kernel(const MpfArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int Mp = a.Mp;
  const int Mpad = (Mp + 63) & ~63, R = blockDim.x / Mpad;
  const int r = __builtin_amdgcn_readfirstlane((int)threadIdx.x / Mpad), i = (int)threadIdx.x - r * Mpad;
  double *dbuf = reinterpret_cast<double *>(sm);  // [R][Mpad][2 P] partial sums (prior: zs + acc[P]; Stein: gk[P] + ks[P])
  float *xs = reinterpret_cast<float *>(dbuf + (size_t)R * Mpad * 2 * P);  // [Mp][P]
  float *sc = xs + Mp * P;    // [Mp][P] scores
  float *nrm = sc + Mp * P;   // [Mp] squared norms
  float *red = nrm + Mp;      // [32]
  const bool on = i < Mp;
  const int per = (Mp + R - 1) / R, k0 = r * per, k1 = min(Mp, k0 + per);
  if (on && r == 0)
    _Pragma("unroll") for (int p = 0; p < P; ++p) xs[i * P + p] = a.x[i * P + p];
  wg_sync();
  const float bw2 = (float)((double)a.bw * (double)a.bw);
  double inv_pbw[4], inv_pbw2[4];
  _Pragma("unroll") for (int p = 0; p < 4; ++p) {
    inv_pbw[p] = 1.0 / (double)a.prior_bwv[p < P ? p : 0];
    inv_pbw2[p] = inv_pbw[p] * inv_pbw[p];
  }
  const double inv_bw2 = 1.0 / ((double)a.bw * (double)a.bw), inv_obs2 = 1.0 / ((double)a.obs_std * (double)a.obs_std);
  float am[4] = {0.f, 0.f, 0.f, 0.f}, av[4] = {0.f, 0.f, 0.f, 0.f};  // Adam moments of this lane's particle (registers for the whole launch)
  const bool adam = a.optimizer == DUST_OPT_ADAM;
  if (adam && on && r == 0)
    _Pragma("unroll") for (int p = 0; p < P; ++p) {
      am[p] = a.adam_m[i * P + p];
      av[p] = a.adam_v[i * P + p];
    }
  for (int it = 0; it < a.n_steps; ++it) {
    float xi[4] = {0.f, 0.f, 0.f, 0.f};
    if (on) {
      _Pragma("unroll") for (int p = 0; p < P; ++p) xi[p] = xs[i * P + p];
      // prior score (mpf.py:45): means alias the CURRENT particles (covariance prior_bw^2 I, uniform mixture), so the
      // i-th logit is exactly 0 and no other is larger: the softmax needs no max pass
      double zs = 0.0, acc[4] = {0, 0, 0, 0};
      for (int k = k0; k < k1; ++k) {
        double q = 0.0;
        _Pragma("unroll") for (int p = 0; p < P; ++p) {
          const double z = ((double)xi[p] - (double)xs[k * P + p]) * inv_pbw[p];
          q += z * z;
        }
        const double w = (double)expf((float)(-0.5 * q));
        zs += w;
        _Pragma("unroll") for (int p = 0; p < P; ++p) acc[p] += w * ((double)xs[k * P + p] - (double)xi[p]);
      }
      double *d = dbuf + ((size_t)r * Mpad + i) * 2 * P;
      d[0] = zs;
      _Pragma("unroll") for (int p = 0; p < P; ++p) d[1 + p] = acc[p];  // 1 + P <= 2 P slots
    }
    wg_sync();
    if (on && r == 0) {
      double zs = 0.0, acc[4] = {0, 0, 0, 0};
      for (int rr = 0; rr < R; ++rr) {
        const double *d = dbuf + ((size_t)rr * Mpad + i) * 2 * P;
        zs += d[0];
        _Pragma("unroll") for (int p = 0; p < P; ++p) acc[p] += d[1 + p];
      }
      double s[4];
      _Pragma("unroll") for (int p = 0; p < P; ++p) s[p] = acc[p] / zs * inv_pbw2[p];
      // likelihood score (mpf.py:46-50, likelihoods.py:30-49)
      float pred[4];
      for (int k = 0; k < 4; ++k) pred[k] = k < a.ds ? a.past_obs[k] : 0.f;
      const Coef cf = make_coef(a.dm, xi);
      const float pa[2] = {a.act_seq ? a.act_seq[2 * it] : a.past_action[0], a.act_seq ? a.act_seq[2 * it + 1] : a.past_action[1]};
      if (a.dm.model == DUST_MODEL_PENDULUM) model_step<DUST_MODEL_PENDULUM>(a.dm, cf, pred, pa);
      else model_step<DUST_MODEL_PARTICLE>(a.dm, cf, pred, pa);
      double J[4][4];
      step_jacobian<P>(a.dm, a.past_obs, pa, xi, J);
      _Pragma("unroll") for (int p = 0; p < P; ++p) {
        double g = 0.0;
        _Pragma("unroll") for (int k = 0; k < 4; ++k)
          if (k < a.ds) g += J[k][p] * ((double)a.obs[k] - (double)pred[k]);
        s[p] += g * inv_obs2;
        sc[i * P + p] = (float)s[p];
      }
      float nn = 0.f;
      _Pragma("unroll") for (int p = 0; p < P; ++p) nn = nn + xi[p] * xi[p];
      nrm[i] = nn;
    }
    wg_sync();
    // kernel + phi (svgd.py:92-99, mpf.py:52-56).  squared_distance's fp32 addmm rounding is followed: it is part of the
    // reference's result (d^2 / bw^2 amplifies it) - dot as an fma chain, then |b|^2 - 2 a.b, then + |a|^2, clamp 0.
    if (on) {
      double gk[4] = {0, 0, 0, 0}, ks[4] = {0, 0, 0, 0};
      const float ni = nrm[i];
      for (int j = k0; j < k1; ++j) {
        float dot = xi[0] * xs[j * P];
        _Pragma("unroll") for (int q = 1; q < P; ++q) dot = fmaf(xi[q], xs[j * P + q], dot);
        float q = (nrm[j] + (-2.0f * dot)) + ni;
        q = fmaxf(q, 0.f);
        const double k = (double)expf(((-q) / bw2) / 2.0f);
        _Pragma("unroll") for (int p = 0; p < P; ++p) {
          gk[p] -= k * ((double)xi[p] - (double)xs[j * P + p]);
          ks[p] += k * (double)sc[j * P + p];
        }
      }
      double *d = dbuf + ((size_t)r * Mpad + i) * 2 * P;
      _Pragma("unroll") for (int p = 0; p < P; ++p) {
        d[p] = gk[p];
        d[P + p] = ks[p];
      }
    }
    wg_sync();
    float ph[4] = {0.f, 0.f, 0.f, 0.f};
    if (on && r == 0) {
      double gk[4] = {0, 0, 0, 0}, ks[4] = {0, 0, 0, 0};
      for (int rr = 0; rr < R; ++rr) {
        const double *d = dbuf + ((size_t)rr * Mpad + i) * 2 * P;
        _Pragma("unroll") for (int p = 0; p < P; ++p) {
          gk[p] += d[p];
          ks[p] += d[P + p];
        }
      }
      _Pragma("unroll") for (int p = 0; p < P; ++p) ph[p] = (float)(gk[p] * inv_bw2 + ks[p] / Mp);
    }
    float n2 = 0.f;
    _Pragma("unroll") for (int p = 0; p < P; ++p) n2 += ph[p] * ph[p];
    n2 = block_reduce<RED_SUM>(n2, red);
    if (threadIdx.x == 0 && a.grad_norms) a.grad_norms[it] = sqrtf(n2);
    if (on && r == 0 && it == 0 && a.phi_out)
      _Pragma("unroll") for (int p = 0; p < P; ++p) a.phi_out[i * P + p] = ph[p];
    if (on && r == 0) {
      if (adam) {  // x.grad = -phi; optimizer.step() (mpf.py:59-62), the single-tensor Adam of stein.hpp adam_step
        _Pragma("unroll") for (int p = 0; p < P; ++p)
            xs[i * P + p] = adam_step(xi[p], -ph[p], am[p], av[p], a.lr, a.beta1, a.beta2, a.eps, (float)(a.t0 + it + 1));
      } else {
        _Pragma("unroll") for (int p = 0; p < P; ++p) xs[i * P + p] = fmaf(a.lr, ph[p], xi[p]);
      }
    }
    wg_sync();
  }
  if (on && r == 0) {
    _Pragma("unroll") for (int p = 0; p < P; ++p) a.x[i * P + p] = xs[i * P + p];
    if (adam)
      _Pragma("unroll") for (int p = 0; p < P; ++p) {
        a.adam_m[i * P + p] = am[p];
        a.adam_v[i * P + p] = av[p];
      }
  }
}

// ---- the same optimisation spread over the chip -------------------------------------------------------------------------------------
// The single-workgroup kernel above is bound by the VALU issue rate of ONE compute unit: 2 M_p^2 pair terms of ~40 (mostly fp64)
// instructions per step - 40 us per step at M_p = 256, 815 us for the 20 steps of a filter update (profiles/round3_cfg5_kernel_stats.txt),
// longer than the control tick it runs beside.  Here ONE WAVE owns one particle i (4 waves per workgroup, M_p / 4 workgroups): its lanes
// stride over the other particles, the partial sums are combined by a fixed xor-shuffle tree in fp64 (reproducible), lane 0 adds the
// likelihood score and takes the optimiser step.  Per step the waves exchange the scores and the new particles through global memory
// (write-through stores, one arrival per wave on a counter sharded over 16 lines, one polling wave per workgroup) - two grid-wide
// hand-offs of ~3 us; the arithmetic per particle is the kernel's above, summed in another (fixed) order.
// The grid must be co-resident.  Protocol of tick2.hpp: a START BARRIER (workgroup 0 waits <= 200 us for every workgroup, then
// publishes go / abort; nothing is written before "go"), bounded waits, and a COMMIT - particles and optimiser moments are written
// behind the last hand-off by waves that find the time-out flag clear.  The host runs an aborted or uncommitted call on the kernel
// above, which needs no co-residency.
struct MpfGridArgs {
  MpfArgs a;
  float *xg;             // [2][Mp][P] particle generations (step parity)
  float *scg;            // [2][Mp][P] scores
  float *n2g;            // [n_steps][Mp] |phi_i|^2
  unsigned int *cnt;     // 16 lines of score arrivals | 16 lines of particle arrivals | start arrivals | go   (zeroed by the host per launch)
  unsigned int *status;  // [0] a wait timed out [1] calls that did not start [2] waves that did not commit
  int test;              // test hook: 1 abort at the start barrier, 2 "a wait gave up" before the last hand-off
};
enum { MPF_G_WAVES = 4, MPF_G_NT = 64 * MPF_G_WAVES, MPF_G_NSH = 16, MPF_G_LINE = 32 /* words: one 128-byte line per counter */ };

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <int P>
__global__ __launch_bounds__(MPF_G_NT) void mpf_optimize_grid_kernel(const MpfGridArgs g) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const MpfArgs &a = g.a;
  const int Mp = a.Mp, tid = (int)threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = (int)gridDim.x, b = (int)blockIdx.x, i = b * MPF_G_WAVES + wave;
  float *xs = sm;             // [Mp][P] current particles
  float *sc = xs + Mp * P;    // [Mp][P] scores
  float *nrm = sc + Mp * P;   // [Mp] squared norms
  unsigned int *sig = reinterpret_cast<unsigned int *>(nrm + Mp);
  unsigned int *cnt_sc = g.cnt, *cnt_x = g.cnt + MPF_G_NSH * MPF_G_LINE, *cnt_start = g.cnt + 2 * MPF_G_NSH * MPF_G_LINE, *go = cnt_start + MPF_G_LINE;
  unsigned int *tflag = g.status;
  const bool on = i < Mp, lead = on && lane == 0;
  // ---- start barrier
  if (tid == 0) {
    __hip_atomic_fetch_add(cnt_start, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned int v = 0u;
    if (b == 0) {
      const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
      bool ok = true;
      while ((int)(__hip_atomic_load(cnt_start, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (unsigned int)G) < 0) {
        __builtin_amdgcn_s_sleep(2);
        if (__builtin_amdgcn_s_memrealtime() - t_start > 20000ull) {
          ok = false;
          break;
        }
      }
      ok = ok && __hip_atomic_load(tflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && g.test != 1;
      if (!ok) __hip_atomic_fetch_add(g.status + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      v = ok ? 1u : 2u;
      __hip_atomic_store(go, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      unsigned int spins = 0u;
      unsigned long long t_start = 0;
      for (;;) {
        v = __hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v) break;
        __builtin_amdgcn_s_sleep(2);
        if ((++spins & 255u) == 0u) {
          const unsigned long long now = __builtin_amdgcn_s_memrealtime();
          if (!t_start) t_start = now;
          else if (now - t_start > DUST_SPIN_TIMEOUT_TICKS) {  // workgroup 0 never came
            __hip_atomic_store(tflag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v = 2u;
            break;
          }
        }
      }
    }
    sig[0] = v;
  }
  for (int t = tid; t < Mp * P; t += MPF_G_NT) xs[t] = a.x[t];
  wg_sync();
  if (sig[0] != 1u) return;
  for (int j = tid; j < Mp; j += MPF_G_NT) {
    float nn = 0.f;
    _Pragma("unroll") for (int p = 0; p < P; ++p) nn = nn + xs[j * P + p] * xs[j * P + p];
    nrm[j] = nn;
  }
  wg_sync();

  const float bw2 = (float)((double)a.bw * (double)a.bw);
  double inv_pbw[4], inv_pbw2[4];
  _Pragma("unroll") for (int p = 0; p < 4; ++p) {
    inv_pbw[p] = 1.0 / (double)a.prior_bwv[p < P ? p : 0];
    inv_pbw2[p] = inv_pbw[p] * inv_pbw[p];
  }
  const double inv_bw2 = 1.0 / ((double)a.bw * (double)a.bw), inv_obs2 = 1.0 / ((double)a.obs_std * (double)a.obs_std);
  float am[4] = {0.f, 0.f, 0.f, 0.f}, av[4] = {0.f, 0.f, 0.f, 0.f}, xn[4] = {0.f, 0.f, 0.f, 0.f};
  const bool adam = a.optimizer == DUST_OPT_ADAM;
  if (adam && lead)
    _Pragma("unroll") for (int p = 0; p < P; ++p) {
      am[p] = a.adam_m[i * P + p];
      av[p] = a.adam_v[i * P + p];
    }
  // lanes [0, 16) of wave 0 wait for the arrivals of every particle's wave on their shard line
  auto poll = [&](unsigned int *lines, const unsigned int phase) {
    if (wave == 0 && lane < MPF_G_NSH) {
      const unsigned int target = (unsigned int)((Mp >> 4) + ((Mp & 15) > lane ? 1 : 0)) * phase;
      if (target) spin_until(lines + lane * MPF_G_LINE, target, tflag);
    }
    wg_sync();
  };
  auto arrive = [&](unsigned int *lines) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lead) __hip_atomic_fetch_add(lines + (i & 15) * MPF_G_LINE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };

  // likelihood score of this wave's particle (mpf.py:46-50, likelihoods.py:30-49): one model step and its Jacobian, on lane 0 - 2.5 us.
  // It depends on the particle alone, so the term of step it + 1 is computed behind the particle store of step it, under the hop.
  double glik[4] = {0, 0, 0, 0};
  auto lik = [&](const float *xp, const int step) {
    float pred[4];
    for (int k = 0; k < 4; ++k) pred[k] = k < a.ds ? a.past_obs[k] : 0.f;
    const Coef cf = make_coef(a.dm, xp);
    const float pa[2] = {a.act_seq ? a.act_seq[2 * step] : a.past_action[0], a.act_seq ? a.act_seq[2 * step + 1] : a.past_action[1]};
    if (a.dm.model == DUST_MODEL_PENDULUM) model_step<DUST_MODEL_PENDULUM>(a.dm, cf, pred, pa);
    else model_step<DUST_MODEL_PARTICLE>(a.dm, cf, pred, pa);
    double J[4][4];
    step_jacobian<P>(a.dm, a.past_obs, pa, xp, J);
    _Pragma("unroll") for (int p = 0; p < P; ++p) {
      double gl = 0.0;
      _Pragma("unroll") for (int k = 0; k < 4; ++k)
        if (k < a.ds) gl += J[k][p] * ((double)a.obs[k] - (double)pred[k]);
      glik[p] = gl * inv_obs2;
    }
  };
  if (lead) {
    float x0v[4] = {0.f, 0.f, 0.f, 0.f};
    _Pragma("unroll") for (int p = 0; p < P; ++p) x0v[p] = xs[i * P + p];
    lik(x0v, 0);
  }

  for (int it = 0; it < a.n_steps; ++it) {
    float xi[4] = {0.f, 0.f, 0.f, 0.f};
    if (on) {
      _Pragma("unroll") for (int p = 0; p < P; ++p) xi[p] = xs[i * P + p];
      // prior score (mpf.py:45), as in the kernel above
      double zs = 0.0, acc[4] = {0, 0, 0, 0};
      for (int k = lane; k < Mp; k += 64) {
        double q = 0.0;
        _Pragma("unroll") for (int p = 0; p < P; ++p) {
          const double z = ((double)xi[p] - (double)xs[k * P + p]) * inv_pbw[p];
          q += z * z;
        }
        const double w = (double)expf((float)(-0.5 * q));
        zs += w;
        _Pragma("unroll") for (int p = 0; p < P; ++p) acc[p] += w * ((double)xs[k * P + p] - (double)xi[p]);
      }
      zs = wave_sum_f64(zs);
      _Pragma("unroll") for (int p = 0; p < P; ++p) acc[p] = wave_sum_f64(acc[p]);
      if (lane == 0)
        _Pragma("unroll") for (int p = 0; p < P; ++p)
            st_sc1(g.scg + ((size_t)(it & 1) * Mp + i) * P + p, (float)(acc[p] / zs * inv_pbw2[p] + glik[p]));
    }
    arrive(cnt_sc);
    poll(cnt_sc, (unsigned int)(it + 1));
    for (int t = tid; t < Mp * P; t += MPF_G_NT) sc[t] = ld_sc1(g.scg + (size_t)(it & 1) * Mp * P + t);
    wg_sync();
    // kernel + phi (svgd.py:92-99, mpf.py:52-56), as in the kernel above
    if (on) {
      double gk[4] = {0, 0, 0, 0}, ks[4] = {0, 0, 0, 0};
      const float ni = nrm[i];
      for (int j = lane; j < Mp; j += 64) {
        float dot = xi[0] * xs[j * P];
        _Pragma("unroll") for (int q = 1; q < P; ++q) dot = fmaf(xi[q], xs[j * P + q], dot);
        float q = (nrm[j] + (-2.0f * dot)) + ni;
        q = fmaxf(q, 0.f);
        const double k = (double)expf(((-q) / bw2) / 2.0f);
        _Pragma("unroll") for (int p = 0; p < P; ++p) {
          gk[p] -= k * ((double)xi[p] - (double)xs[j * P + p]);
          ks[p] += k * (double)sc[j * P + p];
        }
      }
      _Pragma("unroll") for (int p = 0; p < P; ++p) {
        gk[p] = wave_sum_f64(gk[p]);
        ks[p] = wave_sum_f64(ks[p]);
      }
      if (lane == 0) {
        float ph[4] = {0.f, 0.f, 0.f, 0.f}, n2 = 0.f;
        _Pragma("unroll") for (int p = 0; p < P; ++p) {
          ph[p] = (float)(gk[p] * inv_bw2 + ks[p] / Mp);
          n2 += ph[p] * ph[p];
        }
        if (g.n2g) st_sc1(g.n2g + (size_t)it * Mp + i, n2);
        if (it == 0 && a.phi_out)
          _Pragma("unroll") for (int p = 0; p < P; ++p) a.phi_out[i * P + p] = ph[p];
        _Pragma("unroll") for (int p = 0; p < P; ++p) {
          xn[p] = adam ? adam_step(xi[p], -ph[p], am[p], av[p], a.lr, a.beta1, a.beta2, a.eps, (float)(a.t0 + it + 1)) : fmaf(a.lr, ph[p], xi[p]);
          st_sc1(g.xg + ((size_t)((it + 1) & 1) * Mp + i) * P + p, xn[p]);
        }
      }
    }
    if (g.test == 2 && it == a.n_steps - 1 && b == G - 1 && tid == 0) __hip_atomic_store(tflag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    arrive(cnt_x);
    if (lead && it + 1 < a.n_steps) lik(xn, it + 1);
    poll(cnt_x, (unsigned int)(it + 1));
    if (it + 1 < a.n_steps) {
      for (int j = tid; j < Mp; j += MPF_G_NT) {
        float nn = 0.f;
        _Pragma("unroll") for (int p = 0; p < P; ++p) {
          const float v = ld_sc1(g.xg + ((size_t)((it + 1) & 1) * Mp + j) * P + p);
          xs[j * P + p] = v;
          nn = nn + v * v;
        }
        nrm[j] = nn;
      }
      wg_sync();
    }
  }
  // COMMIT: one look at the flag per workgroup, behind the last hand-off
  if (tid == 0) sig[1] = __hip_atomic_load(tflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  wg_sync();
  if (sig[1] != 0u) {
    if (lead) __hip_atomic_fetch_add(g.status + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  if (lead) {
    _Pragma("unroll") for (int p = 0; p < P; ++p) a.x[i * P + p] = xn[p];
    if (adam)
      _Pragma("unroll") for (int p = 0; p < P; ++p) {
        a.adam_m[i * P + p] = am[p];
        a.adam_v[i * P + p] = av[p];
      }
  }
  if (b == 0 && a.grad_norms)  // ||phi|| of every step: the per-particle squares in particle order
    for (int it = tid; it < a.n_steps; it += MPF_G_NT) {
      double n2 = 0.0;
      for (int j = 0; j < Mp; ++j) n2 += (double)ld_sc1(g.n2g + (size_t)it * Mp + j);
      a.grad_norms[it] = sqrtf((float)n2);
    }
}

// ---- the same grid with a DATA-POLLED exchange (the default; DUST_MPF_POLL=0 switches it off) ------------------------------------------
// Rows cross as 16-byte pieces {tag, v0, v1, tag}: the owner's lane 0 writes each with ONE write-through store; every lane re-loads
// the pieces of ITS keys (sc1 loads, all in flight at once) until both tag words are this step's.  A piece is written by one aligned
// 16-byte store and the tag sits in its first AND last word, so a reader that finds both has the words between them.  No counters, no
// polling wave, no LDS copy, no workgroup barrier in the loop - a hop costs one store-to-load round trip (2.4 us against 5.0 for
// counter + data in tools/allgather_probe.hip); keys live in registers (KC = 4 / 8 / 16 per lane).  Tags are unique per launch and step
// (launch number x 8192 + 2 step + 1 | 2, kept to bit patterns of normal floats), pieces alternate between two buffers by step
// parity: an owner overwrites the piece of two steps ago only after it has seen every other owner's piece of the step in between,
// i.e. after everybody finished reading the old one.  The likelihood term of the NEXT step is computed behind the particle store,
// under the hop.  Start barrier (monotonic counter, no memset per call), bounded waits, commit and fallback as above.
// A trap met on the way: `.y` / `.z` of the loaded vector were folded to `.x` by the compiler inside the `x == tag && w == tag` branch
// (ISA: v_mov v3, v2) - the words are taken out through memcpy.  (Everything below the one-lane start code is derived from opaque
// copies of the arguments; that was part of the same rewrite and is kept.)
struct MpfPollArgs {
  MpfArgs a;
  float *xpc;            // [2][Mp][NP + 1] pieces: NP pieces of particle values, then {tag, |phi_i|^2, 0, tag}
  float *scp;            // [2][Mp][NP] pieces of scores
  unsigned int *cnt;     // [0] start arrivals (monotonic over launches) [32] go word = launch << 2 | 1 (go) / 2 (abort)
  unsigned int *status;
  unsigned int tag0, seq;
  int test;
};

template <int P, int KC>
__global__ __launch_bounds__(MPF_G_NT) void mpf_optimize_poll_kernel(const MpfPollArgs g) {
  constexpr int NP = (P + 1) / 2, NX = NP + 1;
  __shared__ unsigned int sig[2];
  const int tid = (int)threadIdx.x, lane = tid & 63;
  // ---- start barrier (one lane per workgroup)
  if (tid == 0) {
    const int G = (int)gridDim.x;
    unsigned int *cnt_start = g.cnt, *go = g.cnt + MPF_G_LINE;
    __hip_atomic_fetch_add(cnt_start, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned int v = 0u;
    if (blockIdx.x == 0) {
      const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
      bool ok = true;
      while ((int)(__hip_atomic_load(cnt_start, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - g.seq * (unsigned int)G) < 0) {
        __builtin_amdgcn_s_sleep(2);
        if (__builtin_amdgcn_s_memrealtime() - t_start > 20000ull) {
          ok = false;
          break;
        }
      }
      ok = ok && __hip_atomic_load(g.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && g.test != 1;
      if (!ok) __hip_atomic_fetch_add(g.status + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      v = ok ? 1u : 2u;
      __hip_atomic_store(go, (g.seq << 2) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      unsigned int spins = 0u;
      unsigned long long t_start = 0;
      for (;;) {
        const unsigned int w = __hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((w >> 2) == g.seq) {
          v = w & 3u;
          break;
        }
        __builtin_amdgcn_s_sleep(2);
        if ((++spins & 255u) == 0u) {
          const unsigned long long now = __builtin_amdgcn_s_memrealtime();
          if (!t_start) t_start = now;
          else if (now - t_start > DUST_SPIN_TIMEOUT_TICKS) {
            __hip_atomic_store(g.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v = 2u;
            break;
          }
        }
      }
    }
    sig[0] = v;
  }
  wg_sync();
  if (sig[0] != 1u) return;
  // everything below is derived AFTER the barrier from opaque copies of the arguments (nothing scalar lives across the start code)
  const MpfArgs &a = g.a;
  const int Mp = opaque_s(a.Mp), wave = opaque_s(__builtin_amdgcn_readfirstlane(tid >> 6));
  const int G = (int)gridDim.x, b = (int)blockIdx.x, i = b * MPF_G_WAVES + wave;
  unsigned int *tflag = g.status;
  const unsigned int tag0 = (unsigned int)opaque_s((int)g.tag0);
  const bool on = i < Mp;
  const int io = on ? i : 0;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(g.xpc, 0, 2 * Mp * NX * 16, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(g.scp, 0, 2 * Mp * NP * 16, 0x00020000);
  typedef unsigned int v4u_t __attribute__((ext_vector_type(4)));
  float xk[KC][P], sk[KC][P], nk[KC], n2k[KC];
  // poll the pieces of this lane's keys until both tag words match; the values go straight to their registers (SCORES: sk; else xk
  // and, from the extra piece of a particle row, n2k); `done` is a bit mask over (key, piece)
  auto poll = [&](auto scores, const __amdgpu_buffer_rsrc_t r, const int parity, const unsigned int tag) {
    constexpr bool SC = decltype(scores)::value;
    constexpr int RS = SC ? NP : NX;  // pieces per row
    unsigned long long done = 0ull;
#pragma unroll
    for (int c = 0; c < KC; ++c)
      if (!(lane + 64 * c < Mp)) done |= ((1ull << RS) - 1ull) << (c * RS);  // (nothing to see)
    constexpr unsigned long long all = KC * RS >= 64 ? ~0ull : ((1ull << (KC * RS)) - 1ull);
    unsigned int spins = 0u;
    unsigned long long t0 = 0;
    for (;;) {
#pragma unroll
      for (int c = 0; c < KC; ++c)
#pragma unroll
        for (int q = 0; q < RS; ++q)
          if (!((done >> (c * RS + q)) & 1ull)) {
            const v4u_t t = __builtin_amdgcn_raw_buffer_load_b128(r, ((parity * Mp + lane + 64 * c) * RS + q) * 16, 0, 16);
            unsigned int w[4];
            __builtin_memcpy(w, &t, sizeof w);
            if (w[0] == tag && w[3] == tag) {
              const float v0 = __builtin_bit_cast(float, w[1]), v1 = __builtin_bit_cast(float, w[2]);
              if (SC) {
                sk[c][2 * q < P ? 2 * q : 0] = v0;
                if (2 * q + 1 < P) sk[c][2 * q + 1] = v1;
              } else if (q < NP) {
                xk[c][2 * q < P ? 2 * q : 0] = v0;
                if (2 * q + 1 < P) xk[c][2 * q + 1] = v1;
              } else {
                n2k[c] = v0;
              }
              done |= 1ull << (c * RS + q);
            }
          }
      if (!__any(done != all ? 1 : 0)) break;
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 63u) == 0u) {
        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
        if (!t0) t0 = now;
        else if (now - t0 > DUST_SPIN_TIMEOUT_TICKS || __hip_atomic_load(tflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
          __hip_atomic_store(tflag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
    }
  };
  auto put = [&](const __amdgpu_buffer_rsrc_t r, const int piece, const unsigned int tag, const float v0, const float v1) {
    const v4u_t t = {tag, __builtin_bit_cast(unsigned int, v0), __builtin_bit_cast(unsigned int, v1), tag};
    __builtin_amdgcn_raw_buffer_store_b128(t, r, piece * 16, 0, 16);
  };
  auto norms = [&]() {
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      float nn = 0.f;
      _Pragma("unroll") for (int p = 0; p < P; ++p) nn = nn + xk[c][p] * xk[c][p];
      nk[c] = nn;
    }
  };
  const double inv_obs2 = 1.0 / ((double)a.obs_std * (double)a.obs_std);
  double glik[4] = {0, 0, 0, 0};
  auto lik = [&](const float *xp, const int step) {
    float pred[4];
    for (int k = 0; k < 4; ++k) pred[k] = k < a.ds ? a.past_obs[k] : 0.f;
    const Coef cf = make_coef(a.dm, xp);
    const float pa[2] = {a.act_seq ? a.act_seq[2 * step] : a.past_action[0], a.act_seq ? a.act_seq[2 * step + 1] : a.past_action[1]};
    if (a.dm.model == DUST_MODEL_PENDULUM) model_step<DUST_MODEL_PENDULUM>(a.dm, cf, pred, pa);
    else model_step<DUST_MODEL_PARTICLE>(a.dm, cf, pred, pa);
    double J[4][4];
    step_jacobian<P>(a.dm, a.past_obs, pa, xp, J);
    _Pragma("unroll") for (int p = 0; p < P; ++p) {
      double gl = 0.0;
      _Pragma("unroll") for (int k = 0; k < 4; ++k)
        if (k < a.ds) gl += J[k][p] * ((double)a.obs[k] - (double)pred[k]);
      glik[p] = gl * inv_obs2;
    }
  };
  float xi[4] = {0.f, 0.f, 0.f, 0.f};
  _Pragma("unroll") for (int p = 0; p < P; ++p) xi[p] = a.x[io * P + p];
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    const int k = lane + 64 * c;
    _Pragma("unroll") for (int p = 0; p < P; ++p) xk[c][p] = k < Mp ? a.x[k * P + p] : 0.f;
    n2k[c] = 0.f;
  }
  norms();
  const float bw2 = (float)((double)a.bw * (double)a.bw);
  double inv_pbw[4], inv_pbw2[4];
  _Pragma("unroll") for (int p = 0; p < 4; ++p) {
    inv_pbw[p] = 1.0 / (double)a.prior_bwv[p < P ? p : 0];
    inv_pbw2[p] = inv_pbw[p] * inv_pbw[p];
  }
  const double inv_bw2 = 1.0 / ((double)a.bw * (double)a.bw);
  float am[4] = {0.f, 0.f, 0.f, 0.f}, av[4] = {0.f, 0.f, 0.f, 0.f};
  const bool adam = a.optimizer == DUST_OPT_ADAM;
  if (adam && on)
    _Pragma("unroll") for (int p = 0; p < P; ++p) {
      am[p] = a.adam_m[io * P + p];
      av[p] = a.adam_v[io * P + p];
    }
  lik(xi, 0);

  for (int it = 0; it < a.n_steps; ++it) {
    const unsigned int tag_s = 0x40000000u | ((tag0 + 2u * (unsigned int)it + 1u) & 0x3fffffffu), tag_x = 0x40000000u | ((tag0 + 2u * (unsigned int)it + 2u) & 0x3fffffffu);
    {
      double zs = 0.0, acc[4] = {0, 0, 0, 0};
#pragma unroll
      for (int c = 0; c < KC; ++c)
        if (lane + 64 * c < Mp) {
          double q = 0.0;
          _Pragma("unroll") for (int p = 0; p < P; ++p) {
            const double z = ((double)xi[p] - (double)xk[c][p]) * inv_pbw[p];
            q += z * z;
          }
          const double w = (double)expf((float)(-0.5 * q));
          zs += w;
          _Pragma("unroll") for (int p = 0; p < P; ++p) acc[p] += w * ((double)xk[c][p] - (double)xi[p]);
        }
      zs = wave_sum_f64(zs);
      _Pragma("unroll") for (int p = 0; p < P; ++p) acc[p] = wave_sum_f64(acc[p]);
      float sv[4] = {0.f, 0.f, 0.f, 0.f};
      _Pragma("unroll") for (int p = 0; p < P; ++p) sv[p] = (float)(acc[p] / zs * inv_pbw2[p] + glik[p]);
      if (on && lane == 0)
#pragma unroll
        for (int q = 0; q < NP; ++q) put(rs_s, ((it & 1) * Mp + i) * NP + q, tag_s, sv[2 * q], sv[2 * q + 1]);
    }
    poll(std::true_type{}, rs_s, it & 1, tag_s);
    float xn[4] = {0.f, 0.f, 0.f, 0.f};
    {
      double gk[4] = {0, 0, 0, 0}, ks[4] = {0, 0, 0, 0};
      float ni = 0.f;
      _Pragma("unroll") for (int p = 0; p < P; ++p) ni = ni + xi[p] * xi[p];
#pragma unroll
      for (int c = 0; c < KC; ++c)
        if (lane + 64 * c < Mp) {
          float dot = xi[0] * xk[c][0];
          _Pragma("unroll") for (int q = 1; q < P; ++q) dot = fmaf(xi[q], xk[c][q], dot);
          float q = (nk[c] + (-2.0f * dot)) + ni;
          q = fmaxf(q, 0.f);
          const double k = (double)expf(((-q) / bw2) / 2.0f);
          _Pragma("unroll") for (int p = 0; p < P; ++p) {
            gk[p] -= k * ((double)xi[p] - (double)xk[c][p]);
            ks[p] += k * (double)sk[c][p];
          }
        }
      _Pragma("unroll") for (int p = 0; p < P; ++p) {
        gk[p] = wave_sum_f64(gk[p]);
        ks[p] = wave_sum_f64(ks[p]);
      }
      float ph[4] = {0.f, 0.f, 0.f, 0.f}, n2 = 0.f;
      _Pragma("unroll") for (int p = 0; p < P; ++p) {
        ph[p] = (float)(gk[p] * inv_bw2 + ks[p] / Mp);
        n2 += ph[p] * ph[p];
      }
      _Pragma("unroll") for (int p = 0; p < P; ++p)
        xn[p] = adam ? adam_step(xi[p], -ph[p], am[p], av[p], a.lr, a.beta1, a.beta2, a.eps, (float)(a.t0 + it + 1)) : fmaf(a.lr, ph[p], xi[p]);
      if (on && lane == 0) {
        const int row = (((it + 1) & 1) * Mp + i) * NX;
#pragma unroll
        for (int q = 0; q < NP; ++q) put(rs_x, row + q, tag_x, xn[2 * q], xn[2 * q + 1]);
        put(rs_x, row + NP, tag_x, n2, 0.f);
        if (it == 0 && a.phi_out)
          _Pragma("unroll") for (int p = 0; p < P; ++p) a.phi_out[i * P + p] = ph[p];
      }
    }
    _Pragma("unroll") for (int p = 0; p < P; ++p) xi[p] = xn[p];
    if (it + 1 < a.n_steps) lik(xi, it + 1);  // (under the hop)
    if (g.test == 2 && it == a.n_steps - 1 && b == G - 1 && tid == 0) __hip_atomic_store(tflag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    {
      poll(std::false_type{}, rs_x, (it + 1) & 1, tag_x);
      double n2s = 0.0;
#pragma unroll
      for (int c = 0; c < KC; ++c)
        if (lane + 64 * c < Mp) n2s += (double)n2k[c];
      norms();
      if (i == 0 && a.grad_norms) {
        n2s = wave_sum_f64(n2s);
        if (lane == 0) a.grad_norms[it] = sqrtf((float)n2s);
      }
    }
  }
  const unsigned int fl = __builtin_amdgcn_readfirstlane(__hip_atomic_load(tflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  if (!on || lane != 0) return;
  if (fl != 0u) {
    __hip_atomic_fetch_add(g.status + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  _Pragma("unroll") for (int p = 0; p < P; ++p) a.x[i * P + p] = xi[p];
  if (adam)
    _Pragma("unroll") for (int p = 0; p < P; ++p) {
      a.adam_m[i * P + p] = am[p];
      a.adam_v[i * P + p] = av[p];
    }
}

struct MpfBw {
  float v[4];
};
__global__ void mpf_log_prob_kernel(const float *x, const float *means, int n, int K, int P, const MpfBw bw, float *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double mx = -INFINITY;
  for (int k = 0; k < K; ++k) {
    double q = 0.0;
    for (int p = 0; p < P; ++p) {
      const double z = ((double)x[i * P + p] - (double)means[k * P + p]) / (double)bw.v[p];
      q += z * z;
    }
    mx = fmax(mx, -0.5 * q);
  }
  double zs = 0.0;
  for (int k = 0; k < K; ++k) {
    double q = 0.0;
    for (int p = 0; p < P; ++p) {
      const double z = ((double)x[i * P + p] - (double)means[k * P + p]) / (double)bw.v[p];
      q += z * z;
    }
    zs += exp(-0.5 * q - mx);
  }
  double ld = 0.0;
  for (int p = 0; p < P; ++p) ld += log((double)bw.v[p]);
  out[i] = (float)(mx + log(zs) - log((double)K) - ld - 0.5 * P * log(2.0 * M_PI));
}

// mpf.prior.sample([n]): categorical over the M_p components (uniform), then N(mean, bw^2 I)
__global__ void mpf_sample_kernel(const float *means, int K, int P, const MpfBw bw, uint64_t seed, int n, float *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t r[4];
  philox4x32_10((uint32_t)i, 0x6d7066u, 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
  const int k = (int)(((unsigned long long)r[0] * (unsigned long long)K) >> 32);
  float z[4];
  philox_normal4(seed, (uint32_t)i, 0x6d7067u, 1u, 0u, z);
  for (int p = 0; p < P; ++p) out[i * P + p] = means[k * P + p] + bw.v[p] * z[p];
}

// KDEpy 1.1.0 `silvermans_rule` of the pooled particle values (mpf.py:68-73: `silvermans_rule(self.x.view(-1, 1))`; restated in
// oracle/ref_shim.py - third party, parity unpinned) on the device: sigma = min(std(ddof = 1), IQR / 1.349) (the positive one when one
// of them is 0), bw = sigma (3 n / 4)^(-1/5), times bw_scale; 1 when n = 1 or both spreads are 0.  float64 throughout, as the host rule
// (numpy on the float64 copy): mean and squared deviations in two passes, the quartiles by numpy's linear interpolation
// (`a + (b - a) t`, from the upper end when t >= 1/2) between the order statistics around q (n - 1), which are found by RANK (every lane
// counts the values below its own: n <= 4096 values, n^2 comparisons - 0.26 M at 256 particles x 2 parameters).  One workgroup.
__global__ __launch_bounds__(1024) void mpf_silverman_kernel(const float *x, const int n, const float bw_scale, float *out) {
  extern __shared__ float sv_x[];  // [n]
  __shared__ double red[16];
  __shared__ double quart[4];      // order statistics floor / ceil of the two quartile positions
  const int tid = threadIdx.x;
  for (int i = tid; i < n; i += 1024) sv_x[i] = x[i];
  __syncthreads();
  auto block_sum = [&](double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double r = 0.0;
    for (int w = 0; w < 16; ++w) r += red[w];
    return r;
  };
  double acc = 0.0;
  for (int i = tid; i < n; i += 1024) acc += (double)sv_x[i];
  const double mean = block_sum(acc) / (double)n;
  acc = 0.0;
  for (int i = tid; i < n; i += 1024) {
    const double d = (double)sv_x[i] - mean;
    acc += d * d;
  }
  const double ss = block_sum(acc);
  const double p25 = 0.25 * (double)(n - 1), p75 = 0.75 * (double)(n - 1);
  const int k[4] = {(int)floor(p25), min(n - 1, (int)floor(p25) + 1), (int)floor(p75), min(n - 1, (int)floor(p75) + 1)};
  for (int i = tid; i < n; i += 1024) {
    const float v = sv_x[i];
    int rank = 0;
    for (int j = 0; j < n; ++j) {
      const float u = sv_x[j];
      rank += (u < v || (u == v && j < i)) ? 1 : 0;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (rank == k[q]) quart[q] = (double)v;
  }
  __syncthreads();
  if (tid == 0) {
    float bw = 1.0f;
    if (n > 1) {
      const double sd = sqrt(ss / (double)(n - 1));
      auto lerp = [](const double a, const double b, const double t) { return t >= 0.5 ? b - (b - a) * (1.0 - t) : a + (b - a) * t; };
      const double q25 = lerp(quart[0], quart[1], p25 - floor(p25)), q75 = lerp(quart[2], quart[3], p75 - floor(p75));
      const double iqr = (q75 - q25) / 1.3489795003921634;
      double sigma = fmin(sd, iqr);
      if (!(sigma > 0.0)) sigma = fmax(sd, iqr);
      if (sigma > 0.0) bw = (float)(sigma * pow((double)n * 3.0 / 4.0, -0.2) * (double)bw_scale);
      else bw = (float)(1.0 * (double)bw_scale);
    } else {
      bw = (float)(1.0 * (double)bw_scale);
    }
    out[0] = bw;
  }
}

}  // namespace dust

struct dust_mpf {
  dust_mpf_config cfg;
  int Mp, P;
  hipStream_t stream;
  float *x, *gn, *phi, *tmp;
  size_t tmp_cap;
  int optimizer, adam_t;       // dust_mpf_set_optimizer; steps taken so far
  float beta1, beta2, adam_eps;
  float *adam_m, *adam_v;      // [Mp][P] or nullptr (SGD)
  uint32_t *grid_bits;
  int nx, ny;
  float off_x, off_y;
  float prior_bwv[4];          // per parameter dimension; equal once update_prior(bw) has run (mpf.py:85)
  float loc[4], past_obs[4], past_action[2];
  bool have_past;
  // control-channel noise of the one-step prediction (model_cfg.ctrl_noise; particle.py:145-148 through likelihoods.py:30-46)
  std::vector<float> *cz;   // recorded draws [n][da] (dust_mpf_set_ctrl_noise), consumed one per SVGD step
  size_t cz_next;
  std::mt19937_64 *rng;     // draws once the recorded ones are used up
  float *act_seq;           // device: effective action per step of the current launch [steps][2]
  int act_seq_cap;
  // the multi-workgroup form of the optimisation (mpf_optimize_grid_kernel): exchange buffers, counters, status; lazily allocated
  float *gbuf;            // xg [2][Mp][P] | scg [2][Mp][P] | n2g [gsteps][Mp]
  unsigned int *gcnt;     // counters (zeroed per launch) followed by the 4 status words
  int gsteps;             // rows of n2g allocated
  float *pbuf;            // data-polled form (experimental): particle pieces | score pieces
  unsigned int *pcnt;     // ... its start counter and go word (monotonic over launches)
  unsigned int pseq;
  float *hpin;            // pinned host staging: gradient norms [4096] + status words
  bool grid_banned;       // a wait of the grid form timed out once (device shared with another process): single-workgroup kernel from then on
  long long n_grid, n_grid_fallback;
};

static DevModel mpf_dev_model(const dust_mpf *m) {
  dust_ctx fake;
  memset((void *)&fake, 0, sizeof fake);
  fake.cfg = m->cfg.model_cfg;
  fake.cfg.params_log_space = m->cfg.log_space;
  fake.P = m->P;
  fake.grid_bits = m->grid_bits;
  fake.nx = m->nx;
  fake.ny = m->ny;
  fake.off_x = m->off_x;
  fake.off_y = m->off_y;
  return make_dev_model(&fake);
}

extern "C" void dust_mpf_destroy(dust_mpf *m) {
  if (!m) return;
  (void)hipSetDevice(m->cfg.device);
  if (m->stream) (void)hipStreamSynchronize(m->stream);
  float *fp[] = {m->x, m->gn, m->phi, m->tmp, m->adam_m, m->adam_v, m->gbuf, reinterpret_cast<float *>(m->gcnt), m->pbuf, reinterpret_cast<float *>(m->pcnt)};
  for (float *p : fp)
    if (p) (void)hipFree(p);
  if (m->grid_bits) (void)hipFree(m->grid_bits);
  if (m->hpin) (void)hipHostFree(m->hpin);
  if (m->act_seq) (void)hipFree(m->act_seq);
  if (m->stream) (void)hipStreamDestroy(m->stream);
  delete m->cz;
  delete m->rng;
  delete m;
}

extern "C" int dust_mpf_create(const dust_mpf_config *cfg, const float *init_particles, const float *initial_obs, dust_mpf **out) {
  if (!cfg || !init_particles || !initial_obs || !out) return fail(DUST_ERR_INVALID, "null argument");
  *out = nullptr;
  if (cfg->abi_version != DUST_ABI_VERSION) return fail(DUST_ERR_INVALID, "ABI version mismatch");
  if (cfg->n_particles < 1 || cfg->n_particles > 1024) return fail(DUST_ERR_UNSUPPORTED, "MPF supports 1..1024 particles (one workgroup)");
  if (cfg->dim_p < 1 || cfg->dim_p > 4) return fail(DUST_ERR_INVALID, "dim_p must be 1..4");
  if (!(cfg->init_bw > 0.f)) return fail(DUST_ERR_INVALID, "init_bw must be > 0 (the host layer evaluates bw_silverman)");
  if (cfg->model_cfg.model != DUST_MODEL_PENDULUM && cfg->model_cfg.model != DUST_MODEL_PARTICLE)
    return fail(DUST_ERR_UNSUPPORTED, "MPF's one-step prediction and its Jacobian exist for the Pendulum and Particle models only");
  if (!(cfg->obs_std > 0.f)) return fail(DUST_ERR_INVALID, "obs_std must be > 0");
  if (cfg->model_cfg.model == DUST_MODEL_PARTICLE && cfg->model_cfg.control_type != DUST_CONTROL_ACCELERATION)
    return fail(DUST_ERR_UNSUPPORTED, "MPF over Particle(control_type='velocity'): the mass does not enter that model's step (particle.py:152-153), there is nothing to filter");
  if (cfg->model_cfg.model != DUST_MODEL_PARTICLE && cfg->model_cfg.ctrl_noise)
    return fail(DUST_ERR_UNSUPPORTED, "ctrl_noise is a Particle field (particle.py:145-148)");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(DUST_ERR_NO_DEVICE, "no HIP device: libdust_amd has no CPU fallback");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(DUST_ERR_NO_DEVICE, "device %d of %d", cfg->device, ndev);
  HIP_TRY(hipSetDevice(cfg->device));
  dust_mpf *m = new (std::nothrow) dust_mpf();
  if (!m) return fail(DUST_ERR_HIP, "out of host memory");
  memset((void *)m, 0, sizeof *m);
  m->cfg = *cfg;
  m->Mp = cfg->n_particles;
  m->P = cfg->dim_p;
  *out = m;
  HIP_TRY(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
  TRY(dalloc(&m->x, (size_t)m->Mp * m->P));
  TRY(dalloc(&m->phi, (size_t)m->Mp * m->P));
  TRY(dalloc(&m->gn, (size_t)4096));
  HIP_TRY(hipMemcpyAsync(m->x, init_particles, (size_t)m->Mp * m->P * sizeof(float), hipMemcpyHostToDevice, m->stream));
  HIP_TRY(hipStreamSynchronize(m->stream));
  for (int p = 0; p < 4; ++p) m->prior_bwv[p] = cfg->init_bw;
  for (int k = 0; k < cfg->dim_s && k < 4; ++k) m->loc[k] = initial_obs[k];
  m->have_past = false;
  return DUST_OK;
}

extern "C" int dust_mpf_set_grid(dust_mpf *m, const float *grid, int nx, int ny, float off_x, float off_y) {
  if (!m || !grid || nx < 1 || ny < 1) return fail(DUST_ERR_INVALID, "bad grid");
  const size_t cells = (size_t)nx * ny, words = (cells + 31) / 32;
  std::vector<uint32_t> bits(words, 0u);
  for (size_t i = 0; i < cells; ++i) {
    if (grid[i] == 1.0f) bits[i >> 5] |= 1u << (i & 31);
    else if (grid[i] != 0.0f) return fail(DUST_ERR_UNSUPPORTED, "occupancy grid must be binary");
  }
  HIP_TRY(hipSetDevice(m->cfg.device));
  if (m->grid_bits) HIP_TRY(hipFree(m->grid_bits));
  m->grid_bits = nullptr;
  TRY(dalloc(&m->grid_bits, words));
  HIP_TRY(hipMemcpyAsync(m->grid_bits, bits.data(), words * 4, hipMemcpyHostToDevice, m->stream));
  HIP_TRY(hipStreamSynchronize(m->stream));
  m->nx = nx;
  m->ny = ny;
  m->off_x = off_x;
  m->off_y = off_y;
  return DUST_OK;
}

// MPF(optimizer_class=..., **opt_args) (svgd.py:108-122): SGD (the demos' choice) or Adam (the class default).  (Re)starts the state.
extern "C" int dust_mpf_set_optimizer(dust_mpf *m, int optimizer, float beta1, float beta2, float eps) {
  if (!m) return fail(DUST_ERR_INVALID, "null mpf");
  if (optimizer != DUST_OPT_SGD && optimizer != DUST_OPT_ADAM) return fail(DUST_ERR_UNSUPPORTED, "MPF optimiser: SGD or Adam");
  HIP_TRY(hipSetDevice(m->cfg.device));
  m->optimizer = optimizer;
  m->adam_t = 0;
  m->beta1 = beta1;
  m->beta2 = beta2;
  m->adam_eps = eps;
  if (optimizer == DUST_OPT_ADAM) {
    const size_t n = (size_t)m->Mp * m->P;
    if (!m->adam_m) TRY(dalloc(&m->adam_m, n));
    if (!m->adam_v) TRY(dalloc(&m->adam_v, n));
    HIP_TRY(hipMemsetAsync(m->adam_m, 0, n * sizeof(float), m->stream));
    HIP_TRY(hipMemsetAsync(m->adam_v, 0, n * sizeof(float), m->stream));
  }
  return DUST_OK;
}

extern "C" int dust_mpf_clone(const dust_mpf *src, dust_mpf **out) {
  if (!src || !out) return fail(DUST_ERR_INVALID, "null argument");
  std::vector<float> x((size_t)src->Mp * src->P);
  HIP_TRY(hipSetDevice(src->cfg.device));
  HIP_TRY(hipMemcpy(x.data(), src->x, x.size() * sizeof(float), hipMemcpyDeviceToHost));
  TRY(dust_mpf_create(&src->cfg, x.data(), src->loc, out));
  dust_mpf *m = *out;
  memcpy(m->prior_bwv, src->prior_bwv, sizeof m->prior_bwv);
  memcpy(m->loc, src->loc, sizeof m->loc);
  memcpy(m->past_obs, src->past_obs, sizeof m->past_obs);
  memcpy(m->past_action, src->past_action, sizeof m->past_action);
  m->have_past = src->have_past;
  if (src->optimizer == DUST_OPT_ADAM) {
    TRY(dust_mpf_set_optimizer(m, DUST_OPT_ADAM, src->beta1, src->beta2, src->adam_eps));
    const size_t nb = (size_t)src->Mp * src->P * sizeof(float);
    HIP_TRY(hipMemcpy(m->adam_m, src->adam_m, nb, hipMemcpyDeviceToDevice));
    HIP_TRY(hipMemcpy(m->adam_v, src->adam_v, nb, hipMemcpyDeviceToDevice));
    m->adam_t = src->adam_t;
  }
  if (src->grid_bits) {
    const size_t words = ((size_t)src->nx * src->ny + 31) / 32;
    TRY(dalloc(&m->grid_bits, words));
    HIP_TRY(hipMemcpy(m->grid_bits, src->grid_bits, words * 4, hipMemcpyDeviceToDevice));
    m->nx = src->nx;
    m->ny = src->ny;
    m->off_x = src->off_x;
    m->off_y = src->off_y;
  }
  return DUST_OK;
}

enum { MPF_POLL_MAX = 1024 };  // particles up to which the data-polled form is taken (A/B: 512 = counter form above)
enum { MPF_GCNT_WORDS = (2 * dust::MPF_G_NSH + 2) * dust::MPF_G_LINE };
// Whether this call takes a multi-workgroup kernel: an optimisation of >= 2 steps over >= 96 particles - measured, us per 20-step call
// (profiles/round3_mpf_time.txt): single workgroup 64: 151, 96: 231, 128: 270, 256: 797, 512: 2 927, 1024: 11 433; data-polled grid
// 64: 163, 96: 185, 128: 176, 256: 195, 512: 259, 1024: 400; counter grid 256: 363, 512: 498, 1024: 730.
// DUST_MPF_GRID=0 / 1: never / from 8 particles on (tests); DUST_MPF_POLL=0: the counter form.
static bool mpf_grid_ok(const dust_mpf *m, int n_steps, bool optimise) {
  if (!optimise || n_steps < 2 || m->grid_banned) return false;
  const char *env = getenv("DUST_MPF_GRID");
  if (env && atoi(env) == 0) return false;
  return m->Mp >= ((env && atoi(env) == 1) ? 8 : 96);
}

// Effective action of every SVGD step of the coming launch when the model carries control noise: acts = past_action + dyn_std * z, one
// z [da] per phi() call (particle.py:145-148; `acts` is the bare action vector at likelihoods.py:44), in fp32 as the reference adds it.
static int mpf_noisy_actions(dust_mpf *m, int n_steps, const float **dev) {
  *dev = nullptr;
  const dust_config &g = m->cfg.model_cfg;
  if (g.model != DUST_MODEL_PARTICLE || !g.ctrl_noise || (g.dyn_std[0] == 0.f && g.dyn_std[1] == 0.f) || n_steps < 1) return DUST_OK;
  if (!m->act_seq || m->act_seq_cap < n_steps) {
    if (m->act_seq) HIP_TRY(hipFree(m->act_seq));
    m->act_seq = nullptr;
    m->act_seq_cap = std::max(n_steps, 64);
    TRY(dalloc(&m->act_seq, (size_t)2 * m->act_seq_cap));
  }
  std::vector<float> h((size_t)2 * n_steps);
  for (int it = 0; it < n_steps; ++it)
    for (int d = 0; d < 2; ++d) {
      float z;
      if (m->cz && m->cz_next < m->cz->size()) {
        z = (*m->cz)[m->cz_next++];
      } else {
        if (!m->rng) m->rng = new std::mt19937_64(g.seed ^ 0x6d70666e6f697365ull);
        z = std::normal_distribution<float>(0.f, 1.f)(*m->rng);
      }
      const float nz = g.dyn_std[d] * z;
      h[(size_t)2 * it + d] = m->past_action[d] + nz;
    }
  HIP_TRY(hipMemcpyAsync(m->act_seq, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice, m->stream));
  HIP_TRY(hipStreamSynchronize(m->stream));  // (h is a local; the filter's calls are synchronous anyway)
  *dev = m->act_seq;
  return DUST_OK;
}

extern "C" int dust_mpf_set_ctrl_noise(dust_mpf *m, const float *z, int n) {
  if (!m) return fail(DUST_ERR_INVALID, "null mpf");
  if (m->cfg.model_cfg.model != DUST_MODEL_PARTICLE || !m->cfg.model_cfg.ctrl_noise)
    return fail(DUST_ERR_STATE, "control noise belongs to a Particle(deterministic=False) model (model_cfg.ctrl_noise)");
  if (!m->cz) m->cz = new std::vector<float>();
  m->cz->clear();
  m->cz_next = 0;
  if (z && n > 0) m->cz->assign(z, z + (size_t)n * m->cfg.dim_a);
  return DUST_OK;
}

static int mpf_launch(dust_mpf *m, float bw, float lr, int n_steps, float *gn_dev, float *phi_dev, bool optimise = true, bool grid = false,
                      const float *act_seq_dev = nullptr) {
  serve_cancel_device(m->cfg.device);  // (an armed control tick - closed-loop serving - would hold every CU until its plant state arrives)
  if (m->cfg.model_cfg.model == DUST_MODEL_PARTICLE && m->cfg.model_cfg.with_obstacle && m->cfg.model_cfg.can_crash && !m->grid_bits)
    return fail(DUST_ERR_STATE, "Particle model with obstacles: call dust_mpf_set_grid first");
  MpfArgs a;
  memset(&a, 0, sizeof a);
  a.dm = mpf_dev_model(m);
  a.Mp = m->Mp;
  a.P = m->P;
  a.ds = m->cfg.dim_s;
  a.da = m->cfg.dim_a;
  a.n_steps = n_steps;
  a.log_space = m->cfg.log_space;
  for (int p = 0; p < 4; ++p) a.prior_bwv[p] = m->prior_bwv[p];
  a.bw = bw;
  a.lr = lr;
  a.obs_std = m->cfg.obs_std;
  for (int k = 0; k < 4; ++k) {
    a.past_obs[k] = m->past_obs[k];
    a.obs[k] = m->loc[k];
  }
  a.past_action[0] = m->past_action[0];
  a.past_action[1] = m->past_action[1];
  a.act_seq = act_seq_dev;
  a.x = m->x;
  a.grad_norms = gn_dev;
  a.phi_out = phi_dev;
  a.optimizer = (optimise && m->optimizer == DUST_OPT_ADAM) ? DUST_OPT_ADAM : DUST_OPT_SGD;  // (the bare phi evaluation takes no step)
  a.t0 = m->adam_t;
  a.beta1 = m->beta1;
  a.beta2 = m->beta2;
  a.eps = m->adam_eps;
  a.adam_m = m->adam_m;
  a.adam_v = m->adam_v;
  // the data-polled form (keys in registers) unless DUST_MPF_POLL=0, which selects the counter form
  if (grid && m->Mp <= MPF_POLL_MAX && !(getenv("DUST_MPF_POLL") && atoi(getenv("DUST_MPF_POLL")) == 0)) {
    const int NP = (m->P + 1) / 2, NX = NP + 1;
    const size_t fx = (size_t)2 * m->Mp * NX * 4, fs = (size_t)2 * m->Mp * NP * 4;  // floats
    if (!m->pbuf) {
      TRY(dalloc(&m->pbuf, fx + fs));
      HIP_TRY(hipMemsetAsync(m->pbuf, 0, (fx + fs) * sizeof(float), m->stream));  // (tag 0 is never waited for)
    }
    if (!m->pcnt) {
      TRY(dalloc(&m->pcnt, (size_t)2 * MPF_G_LINE));
      HIP_TRY(hipMemsetAsync(m->pcnt, 0, (size_t)2 * MPF_G_LINE * sizeof(unsigned int), m->stream));
    }
    if (!m->gcnt) {
      TRY(dalloc(&m->gcnt, (size_t)MPF_GCNT_WORDS + 4));
      HIP_TRY(hipMemsetAsync(m->gcnt, 0, ((size_t)MPF_GCNT_WORDS + 4) * sizeof(unsigned int), m->stream));
    }
    MpfPollArgs g;
    memset(&g, 0, sizeof g);
    g.a = a;
    g.xpc = m->pbuf;
    g.scp = m->pbuf + fx;
    g.cnt = m->pcnt;
    g.status = m->gcnt + MPF_GCNT_WORDS;
    g.seq = ++m->pseq;
    g.tag0 = g.seq * 8192u;  // (n_steps <= 4096: 2 tags per step)
    if (const char *t = getenv("DUST_MPF_GRID_TEST")) g.test = atoi(t);
    const int G = (m->Mp + MPF_G_WAVES - 1) / MPF_G_WAVES;
#define DUST_LAUNCH_MPFP(PP)                                                                       \
  do {                                                                                              \
    if (m->Mp <= 256) mpf_optimize_poll_kernel<PP, 4><<<G, MPF_G_NT, 0, m->stream>>>(g);            \
    else if (m->Mp <= 512) mpf_optimize_poll_kernel<PP, 8><<<G, MPF_G_NT, 0, m->stream>>>(g);       \
    else mpf_optimize_poll_kernel<PP, 16><<<G, MPF_G_NT, 0, m->stream>>>(g);                        \
  } while (0)
    if (m->P == 1) DUST_LAUNCH_MPFP(1);
    else if (m->P == 2) DUST_LAUNCH_MPFP(2);
    else if (m->P == 3) DUST_LAUNCH_MPFP(3);
    else DUST_LAUNCH_MPFP(4);
#undef DUST_LAUNCH_MPFP
    HIP_TRY(hipGetLastError());
    return DUST_OK;
  }
  if (grid) {
    const size_t np = (size_t)m->Mp * m->P;
    if (!m->gbuf || m->gsteps < n_steps) {
      if (m->gbuf) HIP_TRY(hipFree(m->gbuf));
      m->gbuf = nullptr;
      m->gsteps = std::max(n_steps, 32);
      TRY(dalloc(&m->gbuf, 4 * np + (size_t)m->gsteps * m->Mp));
    }
    if (!m->gcnt) {
      TRY(dalloc(&m->gcnt, (size_t)MPF_GCNT_WORDS + 4));
      HIP_TRY(hipMemsetAsync(m->gcnt, 0, ((size_t)MPF_GCNT_WORDS + 4) * sizeof(unsigned int), m->stream));
    }
    HIP_TRY(hipMemsetAsync(m->gcnt, 0, (size_t)MPF_GCNT_WORDS * sizeof(unsigned int), m->stream));
    MpfGridArgs g;
    memset(&g, 0, sizeof g);
    g.a = a;
    g.xg = m->gbuf;
    g.scg = m->gbuf + 2 * np;
    g.n2g = m->gbuf + 4 * np;
    g.cnt = m->gcnt;
    g.status = m->gcnt + MPF_GCNT_WORDS;
    if (const char *t = getenv("DUST_MPF_GRID_TEST")) g.test = atoi(t);
    const int G = (m->Mp + MPF_G_WAVES - 1) / MPF_G_WAVES;
    const size_t lds = sizeof(float) * ((size_t)2 * np + m->Mp + 8);
    if (m->P == 1) mpf_optimize_grid_kernel<1><<<G, MPF_G_NT, lds, m->stream>>>(g);
    else if (m->P == 2) mpf_optimize_grid_kernel<2><<<G, MPF_G_NT, lds, m->stream>>>(g);
    else if (m->P == 3) mpf_optimize_grid_kernel<3><<<G, MPF_G_NT, lds, m->stream>>>(g);
    else mpf_optimize_grid_kernel<4><<<G, MPF_G_NT, lds, m->stream>>>(g);
    HIP_TRY(hipGetLastError());
    return DUST_OK;
  }
  const int mpad = ((m->Mp + 63) / 64) * 64;
  int R = 1;
  while (mpad * R * 2 <= 1024) R *= 2;
  const size_t lds = sizeof(double) * (size_t)R * mpad * 2 * m->P + sizeof(float) * ((size_t)2 * m->Mp * m->P + m->Mp + 32);
#define DUST_LAUNCH_MPF(PP)                                                                                                        \
  do {                                                                                                                              \
    if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void *)mpf_optimize_kernel<PP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    mpf_optimize_kernel<PP><<<1, mpad * R, lds, m->stream>>>(a);                                                                    \
  } while (0)
  if (m->P == 1) DUST_LAUNCH_MPF(1);
  else if (m->P == 2) DUST_LAUNCH_MPF(2);
  else if (m->P == 3) DUST_LAUNCH_MPF(3);
  else DUST_LAUNCH_MPF(4);
#undef DUST_LAUNCH_MPF
  HIP_TRY(hipGetLastError());
  return DUST_OK;
}

extern "C" int dust_mpf_optimize(dust_mpf *m, const float *action, const float *new_obs, float bw, int n_steps, float *grad_norms) {
  if (!m) return fail(DUST_ERR_INVALID, "null mpf");
  if (n_steps < 0 || n_steps > 4096) return fail(DUST_ERR_INVALID, "n_steps out of range");
  if (!(bw > 0.f)) return fail(DUST_ERR_INVALID, "bw must be > 0 (the host layer evaluates silvermans_rule when bw is None)");
  HIP_TRY(hipSetDevice(m->cfg.device));
  if (new_obs) {  // GaussianLikelihood.condition likelihoods.py:51-64
    if (!action) return fail(DUST_ERR_INVALID, "condition() needs the action that produced new_obs");
    memcpy(m->past_obs, m->loc, sizeof m->past_obs);
    for (int k = 0; k < m->cfg.dim_s && k < 4; ++k) m->loc[k] = new_obs[k];
    for (int k = 0; k < 2; ++k) m->past_action[k] = k < m->cfg.dim_a ? action[k] : 0.f;
    m->have_past = true;
  }
  if (!m->have_past) return fail(DUST_ERR_STATE, "Previous action is None. Need at least one observation to start sampling.");
  const bool grid = mpf_grid_ok(m, n_steps, true);
  const float *acts = nullptr;
  TRY(mpf_noisy_actions(m, n_steps, &acts));
  TRY(mpf_launch(m, bw, m->cfg.lr, n_steps, m->gn, nullptr, true, grid, acts));
  const bool want_gn = grad_norms && n_steps > 0;
  if (!m->hpin) HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&m->hpin), (4096 + 8) * sizeof(float), hipHostMallocDefault));
  bool gn_read = false;
  if (grid) {  // did the grid form start, and did it commit?  (tick2.hpp's protocol; this call is synchronous anyway)
    // status and gradient norms come back through pinned memory behind ONE synchronisation
    unsigned int *st = reinterpret_cast<unsigned int *>(m->hpin + 4096);
    HIP_TRY(hipMemcpyAsync(st, m->gcnt + MPF_GCNT_WORDS, 3 * sizeof(unsigned int), hipMemcpyDeviceToHost, m->stream));
    if (want_gn) HIP_TRY(hipMemcpyAsync(m->hpin, m->gn, n_steps * sizeof(float), hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    gn_read = want_gn;
    m->n_grid++;
    if (st[0] | st[1] | st[2]) {
      HIP_TRY(hipMemsetAsync(m->gcnt + MPF_GCNT_WORDS, 0, 4 * sizeof(unsigned int), m->stream));
      if (st[0]) m->grid_banned = true;  // (a wait gave up: another process computes on the device - handoff.hpp)
      if (st[2] != 0u && st[2] != (unsigned int)m->Mp)
        return fail(DUST_ERR_HIP, "MPF: a hand-off wait timed out while some waves were committing (particles invalid: re-seed them); the device seems to be shared with another process");
      if (st[1] || st[2]) {  // nothing was written: the single-workgroup kernel runs the call
        m->n_grid_fallback++;
        TRY(mpf_launch(m, bw, m->cfg.lr, n_steps, m->gn, nullptr, true, false, acts));  // (the same draws: nothing was committed)
        gn_read = false;
      }
    }
  }
  if (m->optimizer == DUST_OPT_ADAM) m->adam_t += n_steps;
  for (int p = 0; p < 4; ++p) m->prior_bwv[p] = bw;  // update_prior(bw) mpf.py:85
  if (!gn_read) {
    if (want_gn) HIP_TRY(hipMemcpyAsync(m->hpin, m->gn, n_steps * sizeof(float), hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
  }
  if (want_gn) memcpy(grad_norms, m->hpin, n_steps * sizeof(float));
  return DUST_OK;
}

extern "C" int dust_mpf_stats(dust_mpf *m, long long out[2]) {
  if (!m || !out) return fail(DUST_ERR_INVALID, "null argument");
  out[0] = m->n_grid;
  out[1] = m->n_grid_fallback;
  return DUST_OK;
}

extern "C" int dust_mpf_phi(dust_mpf *m, float bw, float *phi) {
  if (!m || !phi) return fail(DUST_ERR_INVALID, "null argument");
  if (!m->have_past) return fail(DUST_ERR_STATE, "Previous action is None. Need at least one observation to start sampling.");
  HIP_TRY(hipSetDevice(m->cfg.device));
  const float *acts = nullptr;
  TRY(mpf_noisy_actions(m, 1, &acts));
  TRY(mpf_launch(m, bw, 0.0f, 1, nullptr, m->phi, false, false, acts));  // lr = 0, SGD form: particles and optimiser state unchanged
  HIP_TRY(hipMemcpyAsync(phi, m->phi, (size_t)m->Mp * m->P * sizeof(float), hipMemcpyDeviceToHost, m->stream));
  HIP_TRY(hipStreamSynchronize(m->stream));
  return DUST_OK;
}

// GaussianLikelihood.condition without an optimisation (used by the host mirror's condition())
extern "C" int dust_mpf_condition(dust_mpf *m, const float *action, const float *new_obs) {
  if (!m || !new_obs) return fail(DUST_ERR_INVALID, "null argument");
  memcpy(m->past_obs, m->loc, sizeof m->past_obs);
  for (int k = 0; k < m->cfg.dim_s && k < 4; ++k) m->loc[k] = new_obs[k];
  if (action) {
    for (int k = 0; k < 2; ++k) m->past_action[k] = k < m->cfg.dim_a ? action[k] : 0.f;
    m->have_past = true;
  }
  return DUST_OK;
}

extern "C" int dust_mpf_get_particles(dust_mpf *m, float *x) {
  if (!m || !x) return fail(DUST_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(m->cfg.device));
  HIP_TRY(hipMemcpyAsync(x, m->x, (size_t)m->Mp * m->P * sizeof(float), hipMemcpyDeviceToHost, m->stream));
  HIP_TRY(hipStreamSynchronize(m->stream));
  return DUST_OK;
}
extern "C" int dust_mpf_set_particles(dust_mpf *m, const float *x) {
  if (!m || !x) return fail(DUST_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(m->cfg.device));
  HIP_TRY(hipMemcpyAsync(m->x, x, (size_t)m->Mp * m->P * sizeof(float), hipMemcpyHostToDevice, m->stream));
  HIP_TRY(hipStreamSynchronize(m->stream));
  return DUST_OK;
}
static dust::MpfBw mpf_bw(const dust_mpf *m) {
  dust::MpfBw b;
  for (int p = 0; p < 4; ++p) b.v[p] = m->prior_bwv[p];
  return b;
}

// MPF(bw=None) with P > 1: bw_silverman of the particle columns is a [P] vector and `bw ** 2 * torch.eye(P)` (mpf.py:31-36) turns it
// into the covariance diag(bw_p^2) of the FIRST prior; every later update_prior(bw) (mpf.py:85) is scalar again.
extern "C" int dust_mpf_set_prior_bw(dust_mpf *m, const float *bw, int n) {
  if (!m || !bw) return fail(DUST_ERR_INVALID, "null argument");
  if (n != 1 && n != m->P) return fail(DUST_ERR_INVALID, "prior bandwidths: 1 or P = %d values, got %d", m->P, n);
  for (int p = 0; p < n; ++p)
    if (!(bw[p] > 0.f)) return fail(DUST_ERR_INVALID, "prior bandwidth %d must be > 0", p);
  for (int p = 0; p < 4; ++p) m->prior_bwv[p] = bw[n == 1 ? 0 : (p < n ? p : 0)];
  return DUST_OK;
}

extern "C" int dust_mpf_get_prior_bw(dust_mpf *m, float *bw) {
  if (!m || !bw) return fail(DUST_ERR_INVALID, "null argument");
  for (int p = 0; p < m->P; ++p) bw[p] = m->prior_bwv[p];
  return DUST_OK;
}

extern "C" int dust_mpf_get_prior(dust_mpf *m, float *means, float *bw) {
  if (!m) return fail(DUST_ERR_INVALID, "null mpf");
  if (means) TRY(dust_mpf_get_particles(m, means));
  if (bw) *bw = m->prior_bwv[0];  // (per-dimension bandwidths: dust_mpf_get_prior_bw)
  return DUST_OK;
}
extern "C" int dust_mpf_prior_sample(dust_mpf *m, int n, uint64_t seed, float *samples) {
  if (!m || !samples || n < 1) return fail(DUST_ERR_INVALID, "bad argument");
  HIP_TRY(hipSetDevice(m->cfg.device));
  TRY(ensure(&m->tmp, &m->tmp_cap, (size_t)n * m->P));
  mpf_sample_kernel<<<(n + 255) / 256, 256, 0, m->stream>>>(m->x, m->Mp, m->P, mpf_bw(m), seed, n, m->tmp);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(samples, m->tmp, (size_t)n * m->P * sizeof(float), hipMemcpyDeviceToHost, m->stream));
  HIP_TRY(hipStreamSynchronize(m->stream));
  return DUST_OK;
}
extern "C" int dust_mpf_prior_log_prob(dust_mpf *m, int n, const float *x, float *log_prob) {
  if (!m || !x || !log_prob || n < 1) return fail(DUST_ERR_INVALID, "bad argument");
  HIP_TRY(hipSetDevice(m->cfg.device));
  TRY(ensure(&m->tmp, &m->tmp_cap, (size_t)n * (m->P + 1)));
  HIP_TRY(hipMemcpyAsync(m->tmp, x, (size_t)n * m->P * sizeof(float), hipMemcpyHostToDevice, m->stream));
  mpf_log_prob_kernel<<<(n + 255) / 256, 256, 0, m->stream>>>(m->tmp, m->x, n, m->Mp, m->P, mpf_bw(m), m->tmp + (size_t)n * m->P);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(log_prob, m->tmp + (size_t)n * m->P, n * sizeof(float), hipMemcpyDeviceToHost, m->stream));
  HIP_TRY(hipStreamSynchronize(m->stream));
  return DUST_OK;
}

// MPF.optimize's `bw = silvermans_rule(self.x.view(-1, 1)) * self.bw_scale` (mpf.py:68-73) on the device: one launch and a 4-byte read-back
// instead of a copy of the particles and a host percentile.
extern "C" int dust_mpf_silverman(dust_mpf *m, float *bw) {
  if (!m || !bw) return fail(DUST_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(m->cfg.device));
  const int n = m->Mp * m->P;
  if (n > 4096) return fail(DUST_ERR_UNSUPPORTED, "Silverman's rule on the device takes up to 4096 pooled values (%d particles x %d parameters)", m->Mp, m->P);
  if (!m->hpin) HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&m->hpin), (4096 + 8) * sizeof(float), hipHostMallocDefault));
  TRY(ensure(&m->tmp, &m->tmp_cap, 4));
  mpf_silverman_kernel<<<1, 1024, (size_t)n * sizeof(float), m->stream>>>(m->x, n, m->cfg.bw_scale, m->tmp);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(m->hpin + 4096 + 4, m->tmp, sizeof(float), hipMemcpyDeviceToHost, m->stream));
  HIP_TRY(hipStreamSynchronize(m->stream));
  *bw = m->hpin[4096 + 4];
  if (!(*bw > 0.f)) return fail(DUST_ERR_HIP, "Silverman's rule gave a bandwidth of %g", (double)*bw);
  return DUST_OK;
}

// One control period of the DUAL loop (simulations.py:104-138) in one call: the filter update for the action just applied and the state
// it led to (mpf.optimize(action, state, bw, n_steps): skipped when action_prev is NULL - the first period), then the controller's
// dynamics samples drawn from the filter's refreshed prior ON THE DEVICE, straight into the controller's parameter buffer (n_steps
// draws of [M][P]: disco.py:171, one per SVGD iteration), then the control tick (optimize + forward).  mpf_bw <= 0: Silverman's rule of
// the filter's particles (mpf.py:68-73), evaluated on the device.  Host round trips: the 4-byte bandwidth and the filter's status words
// (the stream is idle there: the caller has just used the previous tick's outputs), then the tick's outputs.  seed: the Philox key of
// this period's draws (dust_mpf_prior_sample's stream).  *bw_used: the bandwidth the filter update ran with (0: no update).
extern "C" int dust_dual_tick(dust_ctx *c, dust_mpf *m, const float *state, const float *action_prev, int n_steps, int mpf_steps, float bw_in,
                              uint64_t seed, float *a_seq, float *p_weights, float *bw_used) {
  if (!c || !m || !state) return fail(DUST_ERR_INVALID, "null argument");
  if (n_steps < 1 || mpf_steps < 0) return fail(DUST_ERR_INVALID, "bad step counts");
  if (c->cfg.dim_p != m->P) return fail(DUST_ERR_INVALID, "the controller samples dim_p = %d dynamics parameters, the filter carries P = %d", c->cfg.dim_p, m->P);
  if (c->cfg.device != m->cfg.device) return fail(DUST_ERR_INVALID, "controller and filter live on different devices");
  if (comm_active(c)) return fail(DUST_ERR_UNSUPPORTED, "the dual tick runs on an unsharded controller (the filter is replicated: tick it per rank)");
  HIP_TRY(hipSetDevice(c->cfg.device));
  float bw = bw_in;
  if (action_prev) {
    if (!(bw > 0.f)) TRY(dust_mpf_silverman(m, &bw));
    TRY(dust_mpf_optimize(m, action_prev, state, bw, mpf_steps, nullptr));  // (synchronises the filter's stream: its particles are final)
  }
  if (bw_used) *bw_used = action_prev ? bw : 0.f;
  TRY(settle_pending(c));
  const int n = n_steps * c->M;
  TRY(ensure(&c->params_dev, &c->params_cap, (size_t)n * c->P));
  dust::mpf_sample_kernel<<<(n + 255) / 256, 256, 0, c->stream>>>(m->x, m->Mp, m->P, mpf_bw(m), seed, n, c->params_dev);
  HIP_TRY(hipGetLastError());
  c->params_staged = true;  // (dust_svmpc_tick finds its dynamics samples in place: no host copy, no one-launch tick - its replay record keeps host samples)
  const int st = dust_svmpc_tick(c, state, n_steps, nullptr, c->params_dev, 0, a_seq, p_weights);
  c->params_staged = false;
  return st;
}
